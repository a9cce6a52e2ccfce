// lane_api_emu.h -- TEST INFRASTRUCTURE ONLY.  The lane API of particlemdi.jl_amd/csrc/pmdi_sweep2_body.h (PM2_* macros and the
// wave primitives of namespace pmdi_s2) for the HOST: every primitive resolves through the lock-step workgroup emulator of
// wavesim.h.  A translation unit includes this header and THEN the kernel body; the body sees PM2_LANE_API_PROVIDED and leaves
// its own gfx950 definitions out.  Nothing under particlemdi.jl_amd/ includes or knows this file.
#pragma once
#include <math.h>

#include "wavesim.h"

#define PM2_LANE_API_PROVIDED 1
#define PM2_DEV inline
#define PM2_COLD __attribute__((noinline))
#define PM2_HD inline
#define PM2_SMEM (wavesim::lds_base())
#define PM2_TID() (wavesim::tid())
#define PM2_BID() (wavesim::bid())
#define PM2_BALLOT(p) wavesim::ballot((p), __LINE__)
#define PM2_SHFL64(v, src) wavesim::shfl64((v), (src), __LINE__)
#define PM2_WAVE_BARRIER() wavesim::wave_barrier(__LINE__)
#define PM2_BARRIER() wavesim::block_barrier(__LINE__)
#define PM2_LDS_BARRIER() wavesim::block_barrier(__LINE__)
#define PM2_UNI(x) (x)
#define PM2_CLOCK() (0ll)
#define PM2_WALLCLOCK() (0ll)
#define PM2_G(T, p) ((T *)(p))
#define PM2_CONST
#define PM2_LAUNDER(ptr_, T) do { } while (0)
#define PM2_FRESH_VGPR(x_) do { } while (0)
template <class T> inline T pm2_atomic_add(T *p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> inline T pm2_atomic_min(T *p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T pm2_atomic_or(T *p, T v) { const T o = *p; *p = o | v; return o; }
template <class T> inline T pm2_atomic_max(T *p, T v) { const T o = *p; if (v > o) *p = v; return o; }
inline int pm2_popc64(unsigned long long x) { return __builtin_popcountll(x); }
inline int pm2_ffs64(unsigned long long x) { return __builtin_ffsll((long long)x); }

namespace pmdi_s2 {

typedef unsigned long long u64;

inline double shfl_d(double v, int src)
{
    union { double d; u64 u; } a, b;
    a.d = v;
    b.u = PM2_SHFL64(a.u, src);
    return b.d;
}
inline int shfl_i(int v, int src) { return (int)(unsigned)PM2_SHFL64((u64)(unsigned)v, src); }
// the value of lane `src` (wave-uniform)
inline int readlane_i(int v, int src) { return shfl_i(v, src); }
inline u64 readlane_u64(u64 v, int src) { return PM2_SHFL64(v, src); }
inline double wave_max_d(double v)
{
    const int lane = PM2_TID() & 63;
    for (int o = 1; o < 64; o <<= 1) { const double t = shfl_d(v, lane ^ o); v = (t > v) ? t : v; }
    return v;
}
inline double wave_min_d(double v)
{
    const int lane = PM2_TID() & 63;
    for (int o = 1; o < 64; o <<= 1) { const double t = shfl_d(v, lane ^ o); v = (t < v) ? t : v; }
    return v;
}
inline double prev_lane_d(double v) { const int lane = PM2_TID() & 63; return shfl_d(v, lane ? lane - 1 : 0); }
// (the device's order: quad, quad pair, half row, row, then (r0 + r1) + (r2 + r3) over the four rows of sixteen lanes)
inline double wave_sum_d(double v)
{
    const int lane = PM2_TID() & 63;
    v = v + shfl_d(v, lane ^ 1);
    v = v + shfl_d(v, lane ^ 2);
    v = v + shfl_d(v, (lane & ~7) | (7 - (lane & 7)));         // row_half_mirror
    v = v + shfl_d(v, (lane & ~15) | (15 - (lane & 15)));      // row_mirror
    return (shfl_d(v, 0) + shfl_d(v, 16)) + (shfl_d(v, 32) + shfl_d(v, 48));
}
// exclusive prefix sum of an int over the wave, and the total
inline int wave_excl_scan_i(int v, int &total)
{
    const int lane = PM2_TID() & 63;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = shfl_i(inc, lane - o); if (lane >= o) inc += t; }
    total = shfl_i(inc, 63);
    return inc - v;
}

}  // namespace pmdi_s2
