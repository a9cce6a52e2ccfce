// pmdi_sweep2_body.h -- the conditional-SMC sweep for SETTLED chains (rounds 3-4): the same sweep as pmdi_sweep.hip
// (src/pmdi.jl:165-171, 188-350, 373; src/misc.jl:15-59), re-designed around what a settled chain looks like (scripts/step_stats.py:
// 1-2 particle classes, <= 8 clusters a class leader can reach, 6-40 distinct columns of particle[:, :, k], ids below ~200):
//
//   * one workgroup of NW waves per chain (256 threads for P <= 1024, 512 for P = 2048), PPL = P / (64 NW) consecutive particles per
//     lane, their state (log-weight, column index and class slot per dataset) in REGISTERS for the whole sweep;
//   * the K datasets of an observation run CONCURRENTLY: wave k owns dataset k for everything per cluster (lanes = features), whatever
//     its cluster type -- Gaussian, Categorical, NegBinom (src/datatypes/*.jl): st_load / st_add_store / st_terms / ordered_sum.
//     Gaussian: the clusters a class leader can reach are cached in that wave's registers (mu, lambda per feature) under a
//     stable slot per cluster id, so a step evaluates one log per (cluster, feature) and reads no pool memory; integer types read
//     their counts / sums from the pool and host-built log / lgamma tables; ordered sums (calc_logprob's feature loop, bit-identical)
//     by one lane per cluster from an LDS transposition; the class CDFs by shuffles;
//   * the draw, the census of the chosen clusters and the Phi / ESS work run on all lanes for all K datasets at once; every
//     per-id / per-column / per-(class, label) table is DIRECT-INDEXED in LDS (ids and columns are small here); what exceeds
//     the LDS capacities (columns >= cols_l, ids >= idcap) lives in the chain's arena with the same indexing (slower, rare);
//   * three workgroup barriers per swept observation (cluster phase | particle phase | bookkeeping phase);
//   * resampling (src/misc.jl:27-47, src/pmdi.jl:317-341): weights stay in registers, Julia's pairwise cumsum as lane-serial
//     leaf scans, the exact u += 1/P sequence per lane, slot counts by direct comparison, the gather through LDS, columns and
//     ids compacted per column / per id.
//
// A chain whose step does not fit the tables (more particle classes than the layout holds, cluster ids beyond 16 bits) is HANDED
// OVER: the bookkeeping phase checks before it commits, hand_over() writes the chain's state where the general kernel keeps it, and
// the same workgroup carries on with the general kernel's code from that observation (PM2_RESUME_GENERAL, defined by pmdi_sweep2.hip).
//
// Everything here is written against a small lane API (PM2_* macros).  This header defines it for gfx950; a translation unit that
// defines PM2_LANE_API_PROVIDED first brings its own -- tests/emu/ does, to run the same source on the CPU in a lock-step workgroup
// emulator (test infrastructure; the product build is hipcc for gfx950 only and never sees it).
// Compile with -ffp-contract=off.  Reference lines are file:line relative to /root/reference.
#pragma once
#include "pmdi_arith.h"
#include "pmdi_internal.h"

// The lane API of this file for gfx950.  (A translation unit that defines PM2_LANE_API_PROVIDED before including this header brings
// its own: that is how the test suite runs this source on a CPU, see tests/emu/.  The product build never does.)
#ifndef PM2_LANE_API_PROVIDED
#define PM2_DEV __device__ __forceinline__
#define PM2_COLD __device__ __noinline__          // out of line: a cold path must not cost the sweep loop registers
#define PM2_HD inline __host__ __device__
extern __shared__ __attribute__((aligned(16))) unsigned char pm2_smem_[];
#define PM2_SMEM (pm2_smem_)
#define PM2_TID() ((int)threadIdx.x)
#define PM2_BID() ((int)blockIdx.x)
#define PM2_BALLOT(p) __ballot(p)
#define PM2_SHFL64(v, src) pm2_shfl64((v), (src))
#define PM2_WAVE_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define PM2_BARRIER() __syncthreads()
#define PM2_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PM2_UNI(x) __builtin_amdgcn_readfirstlane(x)
#define PM2_CLOCK() ((long long)clock64())
#define PM2_WALLCLOCK() ((long long)wall_clock64())      // constant-rate counter shared by the whole device (100 MHz)
// arena / argument pointers are global memory: PM2_G types them so (global_* instead of flat_* instructions); what holds one is
// declared `auto` (a generic pointer variable would drop the address space again).  The host pass of a HIP compile only parses the
// device code: no address spaces there.
#if defined(__HIP_DEVICE_COMPILE__)
#define PM2_G(T, p) ((__attribute__((address_space(1))) T *)(p))
// the argument block is read through the constant address space (s_load, never a vector load); once per swept observation its
// address is passed through an empty asm: what is derived from it (array addresses of K datasets, LDS offsets) is then rebuilt
// from scalar loads where it is used instead of being hoisted out of the sweep loop into registers that do not exist
#define PM2_CONST __attribute__((address_space(4)))
#ifdef PM2_NO_LAUNDER
#define PM2_LAUNDER(ptr_, T) do { } while (0)
#else
#define PM2_LAUNDER(ptr_, T)                                                                                                      \
    do {                                                                                                                          \
        const unsigned long long v_ = (unsigned long long)(ptr_);                                                                 \
        unsigned lo_ = (unsigned)v_, hi_ = (unsigned)(v_ >> 32);                                                                  \
        asm volatile("" : "+s"(lo_), "+s"(hi_));                                                                                  \
        (ptr_) = (const PM2_CONST T *)(((unsigned long long)hi_ << 32) | (unsigned long long)lo_);                               \
    } while (0)
#endif
#define PM2_FRESH_VGPR(x_) asm volatile("" : "+v"(x_))
#else
#define PM2_G(T, p) ((T *)(p))
#define PM2_CONST
#define PM2_LAUNDER(ptr_, T) do { } while (0)
#define PM2_FRESH_VGPR(x_) do { } while (0)
#endif
__device__ __forceinline__ unsigned long long pm2_shfl64(unsigned long long v, int src)
{
    const int lo = __shfl((int)(unsigned)v, src), hi = __shfl((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}
template <class T> __device__ __forceinline__ T pm2_atomic_add(T *p, T v) { return atomicAdd(p, v); }
template <class T> __device__ __forceinline__ T pm2_atomic_min(T *p, T v) { return atomicMin(p, v); }
template <class T> __device__ __forceinline__ T pm2_atomic_or(T *p, T v) { return atomicOr(p, v); }
template <class T> __device__ __forceinline__ T pm2_atomic_max(T *p, T v) { return atomicMax(p, v); }
__device__ __forceinline__ int pm2_popc64(unsigned long long x) { return __popcll(x); }
__device__ __forceinline__ int pm2_ffs64(unsigned long long x) { return __ffsll((long long)x); }

namespace pmdi_s2 {
typedef unsigned long long u64;
PM2_DEV double shfl_d(double v, int src)
{
    union { double d; u64 u; } a, b;
    a.d = v;
    b.u = PM2_SHFL64(a.u, src);
    return b.d;
}
PM2_DEV int shfl_i(int v, int src) { return (int)(unsigned)PM2_SHFL64((u64)(unsigned)v, src); }
// the value of lane `src` (wave-uniform): v_readlane, no LDS round trip
PM2_DEV int readlane_i(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
PM2_DEV u64 readlane_u64(u64 v, int src)
{
    const int s_ = __builtin_amdgcn_readfirstlane(src);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, s_), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), s_);
    return ((u64)hi << 32) | (u64)lo;
}
// wave-level reductions of a double on the DPP network: xor-1 and xor-2 inside quads, half-row and row mirrors give every lane its
// 16-lane row total; the four row totals are read with v_readlane.  All 64 lanes must be active.
template <int CTRL> PM2_DEV double dpp_d(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
PM2_DEV double readlane_d(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
PM2_DEV double wave_sum_d(double v)
{
    v += dpp_d<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);     // row_half_mirror
    v += dpp_d<0x140>(v);     // row_mirror
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
PM2_DEV double wave_max_d(double v)
{
    double t;
    t = dpp_d<0xB1>(v); v = (t > v) ? t : v;
    t = dpp_d<0x4E>(v); v = (t > v) ? t : v;
    t = dpp_d<0x141>(v); v = (t > v) ? t : v;
    t = dpp_d<0x140>(v); v = (t > v) ? t : v;
    const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
    const double m01 = (r1 > r0) ? r1 : r0, m23 = (r3 > r2) ? r3 : r2;
    return (m23 > m01) ? m23 : m01;
}
// the value of the lane below (lane 0 keeps its own): one DPP shift across the whole wave, no LDS round trip
PM2_DEV double prev_lane_d(double v) { return dpp_d<0x138>(v); }      // wave_shr:1
PM2_DEV double wave_min_d(double v)
{
    double t;
    t = dpp_d<0xB1>(v); v = (t < v) ? t : v;
    t = dpp_d<0x4E>(v); v = (t < v) ? t : v;
    t = dpp_d<0x141>(v); v = (t < v) ? t : v;
    t = dpp_d<0x140>(v); v = (t < v) ? t : v;
    const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
    const double m01 = (r1 < r0) ? r1 : r0, m23 = (r3 < r2) ? r3 : r2;
    return (m23 < m01) ? m23 : m01;
}
// exclusive prefix sum of an int over the wave, and the total (the DPP network: shifts inside the rows of sixteen lanes, then the
// row totals broadcast into the rows above; all lanes active)
PM2_DEV int wave_excl_scan_i(int v, int &total)
{
    int inc = v;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, false);      // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, false);      // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, false);      // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, false);      // row_shr:8
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
    total = __builtin_amdgcn_readlane(inc, 63);
    return inc - v;
}
}  // namespace pmdi_s2
#endif  // PM2_LANE_API_PROVIDED

namespace pmdi_s2 {

// A/B builds only (-DPM2_DETAIL_TIMERS, scripts/s2_probe.py): where inside the phases does lane 0 of wave 0 spend its cycles?  The
// sixteen slots of the phase-timer record then hold: 0 need set, 1 terms, 2 ordered sums, 3 uncached clusters, 4 CDF, 5 class rows,
// 6 random draws, 7 chosen clusters + history, 8 census, 9 Phi + maximum, 10 ESS sums, 11 lists of the bookkeeping phase, 12 its
// fast path / clone, class and column work, 13 statistics, 15 clean-up, 14 the whole sweep
// -DPM2_DETAIL_TIMERS=2: the resampling event instead -- 0 weights + uniforms, 1 pairwise cumsum, 2 u table, 3 slot counts, 4 slot
// search, 5 ancestors, 6 per dataset: scatter + gather, 7 particles per column + class leaders, 8 column ranks, 9 id occupancy,
// 10 id ranks, 11 counts move, 12 statistics move, 13 cache + column compaction + classes; 15 everything outside resampling
#ifdef PM2_DETAIL_TIMERS
#define PHX(i_) do { if (ap->phase && tid == 0) { const long long t_ = PM2_CLOCK(); lds<long long>(L.ph)[phd_cur] += t_ - phd_last; phd_last = t_; phd_cur = (i_); } } while (0)
// -DPM2_DETAIL_TIMERS=3: the bookkeeping phase of dataset 0 -- 0 lists, 1 fast-path test, 2 clone-or-in-place, 3 class ids, 4 class
// representatives, 5 column splits, 6 class list, 7 statistics of the chosen clusters, 8 clean-up + counters, 9 the barrier after
#if PM2_DETAIL_TIMERS == 2
#define PHD(i_) PHX(15)
#define PHR(i_) PHX(i_)
#define PHC(i_) do { } while (0)
#elif PM2_DETAIL_TIMERS == 3
// (every owner wave adds its own times: the slots hold sums over the K datasets)
#define PHXW(i_) do { if (ap->phase && lane == 0 && wave < K) { const long long t_ = PM2_CLOCK(); pm2_atomic_add((u64 *)(lds<long long>(L.ph) + phd_cur), (u64)(t_ - phd_last)); phd_last = t_; phd_cur = (i_); } } while (0)
#define PHD(i_) PHXW(15)
#define PHR(i_) do { } while (0)
#define PHC(i_) PHXW(i_)
#else
#define PHD(i_) PHX(i_)
#define PHR(i_) do { } while (0)
#define PHC(i_) do { } while (0)
#endif
#else
#define PHD(i_) do { } while (0)
#define PHR(i_) do { } while (0)
#define PHC(i_) do { } while (0)
#endif

typedef unsigned long long u64;
typedef unsigned short u16;
typedef unsigned char u8;

constexpr int NWMAX = 8;        // waves of a workgroup at most (the workgroup has NW of them: a template parameter of the sweep)
#ifndef PM2_NS
#define PM2_NS 8
#endif
constexpr int NS = PM2_NS;      // cluster cache slots per dataset (registers of the owner wave)
constexpr int XR = 4;           // uncached clusters evaluated per round (their two term rows each borrow the cached clusters' tb rows)
#ifndef PM2_XCAP
#define PM2_XCAP 24
#endif
constexpr int XCAP = PM2_XCAP;  // uncached reachable clusters whose list entry and log-predictive live in LDS (the rest: arena)
constexpr int NR = (NS > 2 * XR) ? NS : 2 * XR;      // term rows per dataset (a round of uncached clusters needs two each)
constexpr int CLSMAX = 32;      // particle classes per dataset the class slots of a lane can name; how many of them a handle's LDS
                                // tables hold (16 .. 32, Layout::cls) is the host's choice for the LDS budget.  A step with more: hand-over
// bits of a class slot in a lane's register word(s): four where all K * PPL slots of the lane then share ONE 64-bit word (K = 4
// datasets x 4 particles per lane on four waves, the headline shape: its LDS budget holds 16 classes per dataset anyway, and the
// two-word form cost that kernel 5 % per step), else five -- one word per dataset
PM2_HD constexpr int class_slot_bits(int K, int PPL, int NW) { return (K == 4 && PPL == 4 && NW == 4) ? 4 : 5; }
PM2_HD constexpr int class_slots_max(int K, int PPL, int NW) { return 1 << class_slot_bits(K, PPL, NW); }
// builds whose mutation-CDF rows may continue in the chain's arena (class slots >= Layout::cdfl): the 8-wave ones, whose N = 50
// tables would otherwise hold 24 classes; the 4-wave builds keep every row in LDS and carry no code for the other case (the
// branches cost the headline kernel 3-5 % when they were there)
PM2_HD constexpr bool cdf_rows_in_arena(int NW) { return NW > 4; }
constexpr int KCAPMAX = 256;    // (class, label) keys a step may touch at most (the key lists of the bookkeeping phase; Layout::kcap)
constexpr int RI_CLSMIN = 64, RI_NEWSLOT = 96, RI_CLSVAL = 128;     // resampling: per-class scratch (CLSMAX ints each), int offsets into the reduction area
constexpr int RI_WCNT = 160;                                         // ... and per-wave counts of the block scans (NWMAX ints)
constexpr int RD_MAX = 0, RD_MIN = NWMAX, RD_SA = 2 * NWMAX, RD_SQ = 3 * NWMAX;      // doubles of the reduction area: per-wave maxima, minima, sums
constexpr int KMAX2 = 4;        // datasets (one owner wave each)
constexpr int NONE8 = 0xFF;
constexpr unsigned INFU = 0xFFFFFFFFu;
constexpr int PMDI_S2_REQUEUE = 1;   // err code: sweep this chain again with the general kernel

// per-dataset scalars (ints in LDS)
enum { DS_MAXID = 0, DS_NCLS, DS_NCOL, DS_NDX, DS_ND, DS_NS0, DS_NX, DS_NNEED, DS_NCLONE, DS_NFLAG, DS_FOLLOW, DS_DIRTY, DS_NEEDMASK, DS_CHANGED, DS_NDLOW, DS_HALT, DS_COUNT = 16 };
// shared scalars
enum { SC_FAIL = 0, SC_RES, SC_PSTAR, SC_NLEAF, SC_NPROG, SC_JS, SC_TMP0, SC_TMP1, SC_TMP2, SC_TMP3, SC_COUNT = 16 };

typedef S2Layout Layout;   // byte offsets into the workgroup's LDS: computed on the host (make_layout), read from the argument block

PM2_HD void make_layout(int K, int N, int P, int Dmax, int cols_l, int idcap, int cls, int cdfl, Layout &L)
{
    const int CLS = cls;                                                   // particle classes per dataset the tables hold
    const int kcap = (CLS * N < KCAPMAX) ? CLS * N : KCAPMAX;              // touched (class, label) keys per step the key lists hold
    if (cdfl > cls) cdfl = cls;                                            // class slots whose CDF rows live in LDS (the rest: the chain's arena)
    L.cls = cls; L.kcap = kcap; L.cdfl = cdfl;
    int o = 0;
    auto take = [&](int bytes) { const int at = o; o = (o + bytes + 15) & ~15; return at; };
    const int Dp = (Dmax + 1) & ~1;
    L.Dp = Dp; L.cols_l = cols_l; L.idcap = idcap;
    L.red = take(128 * 8);       // reductions: doubles [0,32) per-wave maxima / minima / sums (RD_*), ints [64,168) resampling scratch (RI_*), doubles [120] tie, [121] carry
    L.sc = take(SC_COUNT * 4);
    L.stat = take(8 * 8);
    L.wk = take(KMAX2 * 8 * 8);
    L.ph = take(16 * 8);
    L.leaf_i1 = 0; L.leaf_n = 0; L.leaf_prog = 0;
    L.leaf_tot = take(64 * 8); L.leaf_carry = take(64 * 8);      // block totals and the level totals of the cumsum's recursion tree
    L.xfl = take(KMAX2 * 64);                                   // feature flags per dataset (bytes)
    // transient region
    L.tr_tb = 0;
    // the tb rows are dead once the ordered sums are done: their first 1 KiB is the CDF stage's exchange area (and the prefix's label
    // table), the class CDF rows (alive until the particle phase ends) start right behind it
    L.tr_cdf = 128 * 8;
    L.tr_stride = L.tr_cdf + cdfl * (N + 2) * 8;
    if (L.tr_stride < NR * Dp * 8) L.tr_stride = NR * Dp * 8;
    L.tr_stride = (L.tr_stride + 15) & ~15;
    // resampling scratch (aliases the transient rows): u table (P doubles; once dead: the per-dataset gather tables scol, mult,
    // cmap (u16) and scsl (u8): 7 P bytes), slot counts, raw ancestors, ancestors (u16), id histogram (i32)
    L.rs_jtab = P * 8; L.rs_raw = P * 10; L.rs_anc = P * 12; L.rs_hist = P * 14;
    const int rs = P * 14 + idcap * 4 + 64;
    int trb = K * L.tr_stride;
    if (trb < rs) trb = rs;
    L.tr_bytes = trb;
    L.tr = take(trb);
    // dataset block
    const int b0 = o;
    o = 0;
    L.tab = take(cols_l * N * 2);
    L.cmask = take(cols_l * 8); L.wmask = take(cols_l * 8); L.cbi = take(cols_l * 4);
    L.counts = take(idcap * 4); L.cn = take(idcap * 4); L.ncop = take(idcap * 4); L.firstp = take(idcap * 4); L.tgt = take(idcap * 4);
    L.slotmap = take(idcap);
    L.ta = take(NS * Dp * 8); L.tax = L.ta;
    L.lp = take((NS + XCAP) * 8);
    L.slot_id = take(NS * 4); L.slot_cn = take(NS * 4); L.slot_g = take(NS * 8);
    L.clsval = take(CLS * 4); L.clslead = take(CLS * 4); L.leadcol = take(CLS * 4);
    L.minp = take(CLS * N * 4); L.nidv = take(CLS * N * 2); L.knew = take(CLS * N); L.itemj = take(CLS * N * 2);
    L.clist = take(idcap * 2); L.klist = take(kcap * 2); L.kval = take(kcap * 2); L.krep = take(kcap);
    L.bmc = take((P / 64 + 1) * 8); L.bmf = take((P / 64 + 1) * 8);
    L.cbm = take(((idcap + 63) / 64) * 8); L.kbm = take(((CLS * N + 63) / 64) * 8);
    L.xid = take(XCAP * 4);
    L.dsc = take(DS_COUNT * 4);
    L.ds_stride = o;
    L.ds0 = b0;
    L.total = b0 + K * L.ds_stride;
}

// ---- views ------------------------------------------------------------------------------------------------------------------------
template <class Tp> PM2_DEV Tp *lds(int off) { return (Tp *)(PM2_SMEM + off); }

struct Arena {   // the chain's arrays of one dataset in global memory (what exceeds the LDS tables; the pool; the history)
    char *b;
    const PM2_CONST DsetDev *d;
    PM2_DEV int *tabg() const { return (int *)(b + d->o_particle[0]); }
    PM2_DEV int *colg() const { return (int *)(b + d->o_col); }
    PM2_DEV int *pidg() const { return (int *)(b + d->o_pid); }
    PM2_DEV u64 *cmeta() const { return (u64 *)(b + d->o_cgrp); }
    PM2_DEV int *newid() const { return (int *)(b + d->o_newid); }
    PM2_DEV int *counts() const { return (int *)(b + d->o_counts); }
    PM2_DEV int *cn() const { return (int *)(b + d->o_cn); }
    PM2_DEV int *ncop() const { return (int *)(b + d->o_ncop); }
    PM2_DEV int *firstp() const { return (int *)(b + d->o_firstc); }
    PM2_DEV int *tgt() const { return (int *)(b + d->o_lp); }
    PM2_DEV double *sb() const { return (double *)(b + d->o_sb); }
    PM2_DEV int *cnt() const { return (int *)(b + d->o_cnt); }                          // Categorical: counts [id][feature][level]
    PM2_DEV long long *nbs() const { return (long long *)(b + d->o_nbs); }              // NegBinom: sums [id][feature]
    PM2_DEV int *dl() const { return (int *)(b + d->o_dl); }
    PM2_DEV u8 *sstar() const { return (u8 *)(b + d->o_sstar); }
    PM2_DEV double *lpx() const { return (double *)(b + d->o_s2x); }                      // log-predictives of the uncached clusters beyond XCAP
    PM2_DEV int *xidx() const { return (int *)(b + d->o_s2x) + 2 * CLSMAX * 64; }
    PM2_DEV double *cdfg() const { return (double *)(b + d->o_cdf); }                    // CDF rows of the class slots beyond the LDS rows: [slot][N + 2]        // ... and their ids (at most CLSMAX * N <= CLSMAX * 64 entries each)
};

struct DV {      // view of one dataset: LDS block + arena
    int base;            // LDS offset of the dataset block
    int trb;             // LDS offset of its transient rows
    int N, P, D, Dp, cols_l, idcap, cdfl;
    int kind, Lc;        // cluster type of the dataset (K_GAUSSIAN / K_CATEGORICAL / K_NEGBINOM); Categorical: levels per feature in the pool
    const PM2_CONST Layout *lay;
    Arena ar;
    PM2_DEV int *dsc() const { return lds<int>(base + lay->dsc); }
    // particle[label, column]: the distinct columns of particle[:, :, k]
    PM2_DEV int tab_get(int c, int nn) const
    {
        if (c < cols_l) return (int)lds<u16>(base + lay->tab)[c * N + nn];
        return PM2_G(int, ar.tabg())[(size_t)c * N + nn];
    }
    PM2_DEV void tab_set(int c, int nn, int v) const
    {
        if (c < cols_l) lds<u16>(base + lay->tab)[c * N + nn] = (u16)v; else PM2_G(int, ar.tabg())[(size_t)c * N + nn] = v;
    }
    // per-column and per-id tables: direct-indexed, the first cols_l columns / idcap ids in LDS, the rest in the chain's arena.  Every
    // access is an explicit branch between a typed LDS and a typed global access (a pointer select would make it a flat_* instruction)
#define PM2_TABLE(name, Tp, cap, ldsoff, gexpr)                                                                                   \
    PM2_DEV Tp name##_get(int i) const { if (i < cap) return lds<Tp>(base + lay->ldsoff)[i]; return PM2_G(Tp, gexpr)[i]; }          \
    PM2_DEV void name##_set(int i, Tp x) const { if (i < cap) lds<Tp>(base + lay->ldsoff)[i] = x; else PM2_G(Tp, gexpr)[i] = x; }
    PM2_TABLE(cmask, u64, cols_l, cmask, ar.cmeta())
    PM2_TABLE(wmask, u64, cols_l, wmask, ar.cmeta() + P)
    PM2_TABLE(cbi, int, cols_l, cbi, (int *)(ar.cmeta() + 2 * (size_t)P))
    PM2_TABLE(counts, int, idcap, counts, ar.counts())
    PM2_TABLE(cn, int, idcap, cn, ar.cn())
    PM2_TABLE(ncop, int, idcap, ncop, ar.ncop())
    PM2_TABLE(firstp, int, idcap, firstp, ar.firstp())
    PM2_TABLE(tgt, int, idcap, tgt, ar.tgt())
#undef PM2_TABLE
    PM2_DEV void cmask_or(int c, u64 bits) const { if (c < cols_l) pm2_atomic_or(lds<u64>(base + lay->cmask) + c, bits); else pm2_atomic_or(ar.cmeta() + c, bits); }
    PM2_DEV void ncop_add(int id, int x) const { if (id < idcap) pm2_atomic_add(lds<int>(base + lay->ncop) + id, x); else pm2_atomic_add(ar.ncop() + id, x); }
    PM2_DEV int firstp_min(int id, int x) const { if (id < idcap) return pm2_atomic_min(lds<int>(base + lay->firstp) + id, x); return pm2_atomic_min(ar.firstp() + id, x); }
    PM2_DEV int slot_of(int id) const { return id < idcap ? (int)lds<u8>(base + lay->slotmap)[id] : NONE8; }
    PM2_DEV void slot_set(int id, int s) const { if (id < idcap) lds<u8>(base + lay->slotmap)[id] = (u8)s; }
    PM2_DEV double *ta_row(int j) const { return lds<double>(base + lay->ta) + j * Dp; }
    PM2_DEV double *tb_row(int j) const { return lds<double>(trb + lay->tr_tb) + j * Dp; }
    // mutation CDF of class slot r: entries [0, N) the CDF, [N] the log-increment, [N + 1] the one-hot label or -1.  The first cdfl slots'
    // rows live in LDS, the others (a step with that many classes is rare) in the chain's arena
    PM2_DEV double *cdf_row_l(int r) const { return lds<double>(trb + lay->tr_cdf) + r * (N + 2); }
    PM2_DEV double *lp() const { return lds<double>(base + lay->lp); }
    // row j of the step's log-predictives: cache slots [0, NS), then the uncached clusters in the order they were listed
    PM2_DEV double lp_get(int j) const { if (j < NS + XCAP) return lp()[j]; return PM2_G(const double, ar.lpx())[j - NS - XCAP]; }
    PM2_DEV void lp_set(int j, double x) const { if (j < NS + XCAP) lp()[j] = x; else PM2_G(double, ar.lpx())[j - NS - XCAP] = x; }
    PM2_DEV int xid_get(int e) const { if (e < XCAP) return lds<int>(base + lay->xid)[e]; return PM2_G(const int, ar.xidx())[e - XCAP]; }
    PM2_DEV void xid_set(int e, int id) const { if (e < XCAP) lds<int>(base + lay->xid)[e] = id; else PM2_G(int, ar.xidx())[e - XCAP] = id; }
};

// set bits strictly below bit p of a bitmap of 64-bit words
PM2_DEV int popc_below64(const u64 *bm, int p)
{
    int n = 0;
    const int w = p >> 6;
    for (int i = 0; i < w; ++i) n += pm2_popc64(bm[i]);
    return n + pm2_popc64(bm[w] & ((1ull << (p & 63)) - 1ull));
}

// A small array that lives in registers: NN separate scalar fields (no array type, so nothing can ever turn an access into a
// run-time-indexed memory access); a run-time index is a chain of selects, a compile-time one folds away.
template <class Tp, int NN>
struct RegArr {
    Tp head;
    RegArr<Tp, NN - 1> tail;
    PM2_DEV Tp operator[](int i) const { return i == 0 ? head : tail[i - 1]; }
    PM2_DEV void set(int i, Tp x) { head = (i == 0) ? x : head; tail.set(i - 1, x); }
};
template <class Tp>
struct RegArr<Tp, 1> {
    Tp head;
    PM2_DEV Tp operator[](int) const { return head; }
    PM2_DEV void set(int i, Tp x) { head = (i == 0) ? x : head; }
};

// (the levels per feature of a Categorical dataset's pool rows: read here, ahead of the layout shorthand `L` of the struct below)
template <class Ds> PM2_DEV int ds_levels(const Ds &d) { return d.L; }

template <int K, int PPL, int NW, bool GO> struct Sweep2;
// (cold paths of the sweep as out-of-line functions on a COPY of the sweep's per-lane state: inlined, their code cost the sweep loop
// twenty spilled registers)
template <int K, int PPL, int NW, bool GO> PM2_COLD void hand_over_cold(Sweep2<K, PPL, NW, GO> s, long long pos, int fcode, long long t_start);

// ------------------------------------------------------------------------------------------------------------------------------------
// GO ("Gaussian only"): a build for handles whose datasets are all Gaussian -- every cluster-type switch folds at compile time and the
// integer types' code is not in the kernel (the headline shape's kernel is as it was before those types arrived)
template <int K, int PPL, int NW, bool GO>
struct Sweep2 {
    static constexpr int T = 64 * NW;       // threads of the workgroup: NW waves, the first K of them own a dataset each
    static_assert(NW >= K && NW <= NWMAX, "one owner wave per dataset");
    // ---- per-lane state ----
    RegArr<double, PPL> lw;
    static constexpr int NCP = (PPL + 1) / 2;
    RegArr<unsigned, K * NCP> colp;      // [k * NCP + j]: column index of the lane's particles per dataset, 16 bits each
    // class slot of the lane's particles, CSB bits each: in ONE 64-bit word (bit offset CSB * (k * PPL + u)) when they all fit,
    // else one word per dataset (32 bits while that fits; bit offset CSB * u)
    static constexpr bool CDFA = cdf_rows_in_arena(NW);
    PM2_DEV double cdf_get(const DV &v, int r, int j) const { if (!CDFA || r < v.cdfl) return v.cdf_row_l(r)[j]; return PM2_G(const double, v.ar.cdfg())[(size_t)r * (N + 2) + j]; }
    PM2_DEV void cdf_set(const DV &v, int r, int j, double x) const { if (!CDFA || r < v.cdfl) v.cdf_row_l(r)[j] = x; else PM2_G(double, v.ar.cdfg())[(size_t)r * (N + 2) + j] = x; }
    static constexpr int CSB = class_slot_bits(K, PPL, NW);
    static constexpr bool ONEWORD = CSB * K * PPL <= 64;
    static constexpr int NCSW = ONEWORD ? 1 : K;
    template <bool Wide, class Dummy = void> struct CslWord { typedef unsigned type; };
    template <class Dummy> struct CslWord<true, Dummy> { typedef u64 type; };
    typedef typename CslWord<(ONEWORD || CSB * PPL > 32)>::type csl_t;
    RegArr<csl_t, NCSW> cslp;
    static_assert(CSB * PPL <= 64, "class slots of a lane's particles of one dataset must fit one 64-bit word");
    RegArr<double, NS> c_mu, c_lam;     // owner wave: the cluster cache of its dataset (mu, lambda per feature; Sigma, beta stay in the pool), lane = feature
#ifdef PM2_DETAIL_TIMERS
    long long phd_last;
    int phd_cur;
#endif
    double pend_g;              // owner wave, lane 0: the prefix constant of the slot refreshed by the last step's fast path, on its way
    int pend_slot;              // ... and which slot it belongs to (-1: none)
    // ---- uniform ----
    const PM2_CONST SweepArgs *ap;
    int tid, lane, wave, chain;
    int N, P;
    unsigned long long seed;
    unsigned iter;
    long long n, n1;

#define L (ap->s2)
#define CLS (ap->s2.cls)        // particle classes per dataset this handle's LDS tables hold (<= CLSMAX)
    // the packed registers are only ever indexed by compile-time constants: a run-time dataset index goes through these selects
    PM2_DEV void col_get(int k, int (&ck)[PPL]) const
    {
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            if (kk == k) {
#pragma unroll
                for (int u = 0; u < PPL; ++u) ck[u] = (int)((colp[kk * NCP + (u >> 1)] >> ((u & 1) * 16)) & 0xffffu);
            }
    }
    PM2_DEV void col_put(int k, const int (&ck)[PPL])
    {
        unsigned pk[NCP];
#pragma unroll
        for (int j = 0; j < NCP; ++j) pk[j] = 0;
#pragma unroll
        for (int u = 0; u < PPL; ++u) pk[u >> 1] |= (unsigned)ck[u] << ((u & 1) * 16);
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            if (kk == k) {
#pragma unroll
                for (int j = 0; j < NCP; ++j) colp.set(kk * NCP + j, pk[j]);
            }
    }
    // lw[u] for a run-time u (registers: selects, never a run-time index)
    PM2_DEV double lw_get(int u) const
    {
        return lw[u];
    }
    PM2_DEV void lw_set(int u, double x)
    {
        lw.set(u, x);
    }
    PM2_DEV int csl_get(int k, int u) const
    {
        if (ONEWORD) return (int)((cslp[0] >> (CSB * (k * PPL + u))) & (csl_t)((1 << CSB) - 1));
        return (int)((cslp[k] >> (CSB * u)) & (csl_t)((1 << CSB) - 1));
    }
    PM2_DEV void csl_put(int k, int u, int r)
    {
        const int sh = ONEWORD ? CSB * (k * PPL + u) : CSB * u;
        const int w = ONEWORD ? 0 : k;
        cslp.set(w, (csl_t)((cslp[w] & ~((csl_t)((1 << CSB) - 1) << sh)) | ((csl_t)r << sh)));
    }
    PM2_DEV DV view(int k) const
    {
        DV v;
        v.base = L.ds0 + k * L.ds_stride; v.trb = L.tr + k * L.tr_stride;
        v.N = N; v.P = P; v.D = ap->ds[k].D; v.kind = GO ? (int)K_GAUSSIAN : ap->ds[k].kind; v.Lc = ds_levels(ap->ds[k]); v.Dp = L.Dp; v.cols_l = L.cols_l; v.idcap = L.idcap; v.cdfl = L.cdfl; v.lay = &ap->s2;
        v.ar.d = &ap->ds[k];
        v.ar.b = ap->ds[k].arena + (size_t)chain * ap->ds[k].stride;
        return v;
    }
    PM2_DEV int *sc() const { return lds<int>(L.sc); }
    PM2_DEV long long *stat() const { return lds<long long>(L.stat); }
    PM2_DEV long long *wk(int k) const { return lds<long long>(L.wk) + k * 8; }
    PM2_DEV const u8 *flk(int k) const { return lds<u8>(L.xfl) + k * 64; }

    // ---- the three cluster types (src/datatypes/{gaussian,categorical,negbinom}_cluster.jl), lane = feature ------------------------------
    // An observation's value of the lane's feature travels as a double for every type (levels and counts are integers below 2^30:
    // exact); so do the statistics a step touches: Gaussian (Sigma, beta); Categorical: the count of the observed level;
    // NegBinom: the feature's sum (below 2^53: exact).
    struct St { double a, b; };
    PM2_DEV double obs_x(int k, int i) const
    {
        const auto &d = ap->ds[k];
        if (lane >= d.D) return 0.0;
        if (GO || d.kind == K_GAUSSIAN) return PM2_G(const double, d.xf)[(size_t)i * d.D + lane];
        return (double)PM2_G(const int, d.xi)[(size_t)i * d.D + lane];
    }
    PM2_DEV St st_load(const DV &v, int id, double x, bool act) const
    {
        St st;
        st.a = 0.0; st.b = 0.5;
        if (!act) return st;
        if ((GO || v.kind == K_GAUSSIAN)) {
            auto sb = PM2_G(const double, v.ar.sb());
            st.a = sb[((size_t)id * v.D + lane) * 2]; st.b = sb[((size_t)id * v.D + lane) * 2 + 1];
        } else if ((!GO && v.kind == K_CATEGORICAL)) {
            st.a = (double)PM2_G(const int, v.ar.cnt())[((size_t)id * v.D + lane) * v.Lc + ((int)x - 1)];
        } else {
            st.a = (double)PM2_G(const long long, v.ar.nbs())[(size_t)id * v.D + lane];
        }
        return st;
    }
    // cluster_add! of the observation, in place, for the lane's feature (gaussian_cluster.jl:54-66, categorical_cluster.jl:43-51,
    // negbinom_cluster.jl:43-51): the statistics as st_load returned them, the cluster's new size
    PM2_DEV void st_add_store(const DV &v, int id, St &st, double x, int nnew, bool on) const
    {
        if (!on) return;
        if ((GO || v.kind == K_GAUSSIAN)) {
            pmdi_arith::gauss_add_sb(x, nnew, st.a, st.b);
            auto sb = PM2_G(double, v.ar.sb());
            sb[((size_t)id * v.D + lane) * 2] = st.a; sb[((size_t)id * v.D + lane) * 2 + 1] = st.b;
        } else if ((!GO && v.kind == K_CATEGORICAL)) {
            st.a = st.a + 1.0;
            PM2_G(int, v.ar.cnt())[((size_t)id * v.D + lane) * v.Lc + ((int)x - 1)] = (int)st.a;
        } else {
            st.a = st.a + x;
            PM2_G(long long, v.ar.nbs())[(size_t)id * v.D + lane] = (long long)st.a;
        }
    }
    // the per-feature term(s) of calc_logprob for a cluster of size cn whose statistics (as st_load returned them) are st.
    // Gaussian: both terms of gaussian_cluster.jl:45-48; Categorical: log(nlevels_q + n) and log(0.5 + counts[x_q, q])
    // (categorical_cluster.jl:30,35-38; the n == 0 branch reads log(0.5)); NegBinom: the six loggammas of negbinom_cluster.jl:33-37 (tb).
    // Integer types read the host-built tables, so their log-predictives are the bits a CPU evaluation with the same libm gives.
    PM2_DEV void st_terms(const DV &v, int k, int cn, const St &st, double x, double &ta, double &tb) const
    {
        const auto &d = ap->ds[k];
        if ((GO || v.kind == K_GAUSSIAN)) {
            double mu, lam;
            pmdi_arith::gauss_ml(cn, st.a, st.b, mu, lam);
            const double nd_ = (double)cn, dd = x - mu;
            ta = 0.5 * log(lam / (nd_ + 1.0));
            tb = (0.5 * nd_ + 1.0) * log(1.0 + (1.0 / (nd_ + 1.0)) * (dd * dd) * lam);
        } else if ((!GO && v.kind == K_CATEGORICAL)) {
            auto lh = PM2_G(const double, d.lhtab);
            ta = lh[PM2_G(const int, d.maxcol)[lane] + 2 * cn];
            tb = (cn == 0) ? lh[1] : lh[2 * (int)st.a + 1];
        } else {
            auto lg = PM2_G(const double, d.lgtab);
            const long long n_ = cn, x_ = (long long)x, S_ = (long long)st.a;
            ta = 0.0;
            tb = lg[1 + n_ + 1] + lg[1 + x_ + S_] + lg[1 + n_ + 1 + S_] - lg[1 + n_ + 1 + 1 + x_ + S_] - lg[1 + n_] - lg[1 + S_];
        }
    }
    // calc_logprob's sum over the features that are switched on, in feature order (one lane): the same terms in the same order as
    // the reference's loops (gaussian_cluster.jl:41-50; categorical_cluster.jl:30 then :35-38; negbinom_cluster.jl:25-40)
    PM2_DEV double ordered_sum(int kind, const double *ta, const double *tb, int D, const u8 *fl, bool all_on, double g0) const
    {
        double out;
        if (kind == K_GAUSSIAN) {
            out = g0;
            if (all_on) {                       // all features on: fetch eight features' terms, then add them in order
                for (int q0 = 0; q0 < D; q0 += 8) {
                    double ra[8], rb[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int q = (q0 + u < D) ? q0 + u : D - 1; ra[u] = ta[q]; rb[u] = tb[q]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (q0 + u < D) { out += ra[u]; out -= rb[u]; }
                }
            } else {
                for (int q = 0; q < D; ++q) if (fl[q]) { out += ta[q]; out -= tb[q]; }
            }
        } else if (kind == K_CATEGORICAL) {
            double acc = 0.0;
            for (int q = 0; q < D; ++q) if (fl[q]) acc += ta[q];
            out = -acc;
            for (int q = 0; q < D; ++q) if (fl[q]) out += tb[q];
        } else {
            out = 0.0;
            for (int q = 0; q < D; ++q) if (fl[q]) out += tb[q];
        }
        return out;
    }

    // ---- cluster cache (owner wave) -------------------------------------------------------------------------------------------
    // cluster `id` of size cnv takes slot s0 (uniform) of the owner wave, lane = feature.  Gaussian: (mu, lambda) from its statistics
    // into the wave's registers, the first term of every feature and the prefix constant into LDS.  Categorical: the first term
    // (it depends on the size only).  The integer types read their statistics from the pool at every step.
    PM2_DEV void cache_fill(const DV &v, int k, int s0, double sg, double bt, int cnv, bool defer = false, bool have_g = false, double g_in = 0.0)
    {
        if ((!GO && v.kind != K_GAUSSIAN)) {
            if ((!GO && v.kind == K_CATEGORICAL) && lane < v.D)
                v.ta_row(s0)[lane] = PM2_G(const double, ap->ds[k].lhtab)[PM2_G(const int, ap->ds[k].maxcol)[lane] + 2 * cnv];
            if (lane == 0) lds<int>(v.base + L.slot_cn)[s0] = cnv;
            return;
        }
        double mu, lam;
        pmdi_arith::gauss_ml(cnv, sg, bt, mu, lam);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s == s0) { c_mu.set(s, mu); c_lam.set(s, lam); }
        if (lane < v.D) v.ta_row(s0)[lane] = 0.5 * log(lam / ((double)cnv + 1.0));            // gaussian_cluster.jl:45
        if (lane == 0) {
            lds<int>(v.base + L.slot_cn)[s0] = cnv;
            const double g = have_g ? g_in : PM2_G(const double, ap->ds[k].gtab)[cnv];            // gaussian_cluster.jl:38-40
            // (deferred: the load stays in flight; the next cluster phase of this wave writes the value where the ordered sums read it)
            if (defer) { pend_g = g; pend_slot = s0; }
            else lds<double>(v.base + L.slot_g)[s0] = (double)v.dsc()[DS_NFLAG] * g;
        }
    }

    // ---- reset (src/pmdi.jl:165-171) and known prefix (:188-207), by the owner wave of dataset k ----------------------------------
    PM2_DEV void prefix(int k)
    {
        const DV v = view(k);
        const auto &d = ap->ds[k];
        const int D = d.D;
        int *dsc = v.dsc();
        auto s_in = PM2_G(const int, ap->s_in) + ((size_t)chain * K + k) * n;
        auto order = PM2_G(const int, ap->order) + (size_t)chain * n;
        const u8 *flags = ap->flags ? PM2_G(const u8, ap->flags) + (size_t)chain * ap->sumD + d.flag_off : (const u8 *)nullptr;
        u8 *fl = lds<u8>(L.xfl) + k * 64;
        int *lab = lds<int>(v.trb + L.tr_tb);               // [0,64) first position, [64,128) id, [128,192) count (the tb rows are idle)
        for (int e = lane; e < v.idcap; e += 64) {
            lds<int>(v.base + L.counts)[e] = 0; lds<int>(v.base + L.cn)[e] = 0; lds<int>(v.base + L.ncop)[e] = 0;
            lds<int>(v.base + L.firstp)[e] = 0x7fffffff; lds<int>(v.base + L.tgt)[e] = 0; lds<u8>(v.base + L.slotmap)[e] = NONE8;
        }
        for (int e = lane; e < v.cols_l; e += 64) { lds<u64>(v.base + L.cmask)[e] = 0; lds<u64>(v.base + L.wmask)[e] = 0; lds<int>(v.base + L.cbi)[e] = 0; }
        for (int e = lane; e < CLS * N; e += 64) lds<unsigned>(v.base + L.minp)[e] = INFU;
        for (int e = lane; e < P / 64 + 1; e += 64) { lds<u64>(v.base + L.bmc)[e] = 0; lds<u64>(v.base + L.bmf)[e] = 0; }
        for (int e = lane; e < NS; e += 64) lds<int>(v.base + L.slot_id)[e] = 0;
        for (int e = lane; e < (v.idcap + 63) / 64; e += 64) lds<u64>(v.base + L.cbm)[e] = 0;
        for (int e = lane; e < (CLS * N + 63) / 64; e += 64) lds<u64>(v.base + L.kbm)[e] = 0;
        if (lane < 64) { lab[lane] = 0x7fffffff; lab[64 + lane] = 0; lab[128 + lane] = 0; fl[lane] = (lane < D) ? (flags ? flags[lane] : (u8)1) : (u8)0; }
        // what lives in the arena: new_id (:167), the per-id scratch beyond idcap, the column tables beyond cols_l
        {
            auto nid = PM2_G(int, v.ar.newid());
            for (int e = lane; e < N * P; e += 64) nid[e] = 0;
            auto g1 = PM2_G(int, v.ar.ncop());
            auto g2 = PM2_G(int, v.ar.firstp());
            auto g3 = PM2_G(int, v.ar.counts());
            for (int e = v.idcap + lane; e <= ap->cap; e += 64) { g1[e] = 0; g2[e] = 0x7fffffff; g3[e] = 0; }
            auto cm = PM2_G(u64, v.ar.cmeta());
            for (int e = v.cols_l + lane; e < P; e += 64) { cm[e] = 0; cm[P + e] = 0; }
        }
        PM2_WAVE_BARRIER();
        // unique(s[order_obs[1:n1-1], k]) in first-appearance order (:192)
        for (long long j = lane; j < n1 - 1; j += 64) {
            const int u = s_in[order[j]];
            pm2_atomic_min(&lab[u], (int)j);
            pm2_atomic_add(&lab[128 + u], 1);
        }
        PM2_WAVE_BARRIER();
        int nu = 0;
        for (int u = 0; u < N; ++u) nu += (lab[u] != 0x7fffffff) ? 1 : 0;
        if (lane < N) {
            const int fp = lab[lane];
            int id = 1;
            if (fp != 0x7fffffff) {
                int r = 0;
                for (int u = 0; u < N; ++u) r += (lab[u] < fp) ? 1 : 0;
                id = 2 + r;                                    // cluster id of label u (:197)
                lab[64 + lane] = id;
                v.counts_set(id, P);
                v.cn_set(id, lab[128 + lane]);
            }
            v.tab_set(0, lane, id);                            // particle[u, :, k] .= id (:195): one column
        }
        if (lane == 0) { v.counts_set(1, P * N - nu * P); v.cn_set(1, 0); }
        PM2_WAVE_BARRIER();
        // fresh clusters and the first n1-1 shuffled observations joining their previous cluster, sequentially in shuffled
        // order (:189,:194,:201-206): lane = feature, one label after the other
        if (GO || d.kind == K_GAUSSIAN) {
            auto sb = PM2_G(double, v.ar.sb());
            auto xf = PM2_G(const double, d.xf);
            if (lane < D) { sb[((size_t)1 * D + lane) * 2] = 0.0; sb[((size_t)1 * D + lane) * 2 + 1] = 0.5; }
            for (int u = 0; u < N; ++u) {
                const int id = lab[64 + u];
                if (!id) continue;
                double sg = 0.0, bt = 0.5;
                if (lane < D && fl[lane]) {
                    int c = 0;
                    for (long long j = 0; j < n1 - 1; ++j) {
                        const int i = order[j];
                        if (s_in[i] != u) continue;
                        ++c;
                        pmdi_arith::gauss_add_sb(xf[(size_t)i * D + lane], c, sg, bt);
                    }
                }
                if (lane < D) { sb[((size_t)id * D + lane) * 2] = sg; sb[((size_t)id * D + lane) * 2 + 1] = bt; }
            }
        } else if ((!GO && d.kind == K_CATEGORICAL)) {
            // counts[level, feature] (categorical_cluster.jl:43-51): the lane's feature, one level after the other (the count of a level
            // is a sum over the label's observations: no order to keep)
            auto cn_ = PM2_G(int, v.ar.cnt());
            auto xi = PM2_G(const int, d.xi);
            const int Lc = v.Lc;
            if (lane < D) for (int l = 0; l < Lc; ++l) cn_[((size_t)1 * D + lane) * Lc + l] = 0;
            for (int u = 0; u < N; ++u) {
                const int id = lab[64 + u];
                if (!id) continue;
                if (lane < D) for (int l = 0; l < Lc; ++l) cn_[((size_t)id * D + lane) * Lc + l] = 0;
                if (lane < D && fl[lane]) {
                    for (long long j = 0; j < n1 - 1; ++j) {
                        const int i = order[j];
                        if (s_in[i] != u) continue;
                        cn_[((size_t)id * D + lane) * Lc + (xi[(size_t)i * D + lane] - 1)] += 1;
                    }
                }
            }
        } else {
            auto nb_ = PM2_G(long long, v.ar.nbs());
            auto xi = PM2_G(const int, d.xi);
            if (lane < D) nb_[(size_t)1 * D + lane] = 0;
            for (int u = 0; u < N; ++u) {
                const int id = lab[64 + u];
                if (!id) continue;
                long long S = 0;
                if (lane < D && fl[lane]) {
                    for (long long j = 0; j < n1 - 1; ++j) {
                        const int i = order[j];
                        if (s_in[i] != u) continue;
                        S += xi[(size_t)i * D + lane];                                 // negbinom_cluster.jl:43-51
                    }
                }
                if (lane < D) nb_[(size_t)id * D + lane] = S;
            }
        }
        if (lane == 0) {
            int nf = 0;
            for (int q = 0; q < D; ++q) nf += fl[q];
            dsc[DS_NFLAG] = nf; dsc[DS_MAXID] = nu + 1; dsc[DS_NCLS] = 1; dsc[DS_NCOL] = 1; dsc[DS_NDX] = 0; dsc[DS_NX] = 0;
            dsc[DS_DIRTY] = 1; dsc[DS_CHANGED] = 0; dsc[DS_FOLLOW] = 0; dsc[DS_NEEDMASK] = 0; dsc[DS_NCLONE] = 0; dsc[DS_ND] = 0; dsc[DS_NDLOW] = 0; dsc[DS_HALT] = 0;
            lds<int>(v.base + L.clsval)[0] = 1; lds<int>(v.base + L.clslead)[0] = 0; lds<int>(v.base + L.leadcol)[0] = 0;
        }
        PM2_WAVE_BARRIER();
    }

    // ---- phase A (owner wave of dataset k): log-predictives of the clusters the class leaders can reach (src/pmdi.jl:218-220,:232),
    //      the mutation CDF of every particle class (:231-248) -------------------------------------------------------------------------
    PM2_DEV void phase_a(int k, double x, int ns0_cur, long long pos)
    {
        const DV v = view(k);
        const auto &d = ap->ds[k];
        const int D = d.D;
        int *dsc = v.dsc();
        const int ncls = PM2_UNI(dsc[DS_NCLS]);
        const u8 *fl = flk(k);
        const bool on = lane < D && fl[lane];
        u16 *itemj = lds<u16>(v.base + L.itemj);
        int *slot_id = lds<int>(v.base + L.slot_id);
        PHD(0);
        if (lane == 0 && pend_slot >= 0) { lds<double>(v.base + L.slot_g)[pend_slot] = (double)dsc[DS_NFLAG] * pend_g; pend_slot = -1; }
        // -- A1: the clusters the class leaders' columns hold; their cache slots (stable while a cluster stays reachable).  Skipped
        // while nothing it depends on has changed (phase C and the resampling say so): most steps of a settled chain
        unsigned needmask = 0;
        int nx = 0;
        if (PM2_UNI(dsc[DS_DIRTY]) == 0) { needmask = (unsigned)PM2_UNI(dsc[DS_NEEDMASK]); nx = PM2_UNI(dsc[DS_NX]); }
        else {
        // the leaders' columns, four classes at a time: cache slots in use; new_id of the (class, label) keys, fetched now (four loads in
        // flight) and needed after the draw (:266)
        for (int r0 = 0; r0 < ncls; r0 += 4) {
            int nid_pref[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + j;
                nid_pref[j] = 0;
                if (r < ncls) {
                    const int lc = lds<int>(v.base + L.leadcol)[r];
                    const int id = (lane < N) ? v.tab_get(lc, lane) : 0;
                    const int s = (lane < N) ? v.slot_of(id) : NONE8;
                    for (int sb_ = 0; sb_ < NS; ++sb_) if (PM2_BALLOT(s == sb_)) needmask |= 1u << sb_;
                    if (lane < N && ap->q1 == 0) nid_pref[j] = PM2_G(const int, v.ar.newid())[(size_t)(lds<int>(v.base + L.clsval)[r] - 1) * N + lane];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (r0 + j < ncls && lane < N) lds<u16>(v.base + L.nidv)[(r0 + j) * N + lane] = (u16)nid_pref[j];
        }
        int nneed_new = 0;
#pragma nounroll
        for (int r = 0; r < ncls; ++r) {
            {
                const int id = (lane < N) ? v.tab_get(lds<int>(v.base + L.leadcol)[r], lane) : 0;
                int j = (lane < N) ? v.slot_of(id) : 0;
                bool missing = lane < N && j == NONE8;
                if (missing)                                   // an uncached cluster another class already asked for
                    for (int e = 0; e < nx; ++e) if (v.xid_get(e) == id) { j = NS + e; missing = false; }
                u64 m;
                while ((m = PM2_BALLOT(missing)) != 0) {
                    const int l0 = pm2_ffs64(m) - 1;
                    const int id0 = readlane_i(id, l0);
                    const unsigned freem = ~needmask & ((1u << NS) - 1u);
                    int row;
                    if (id0 < v.idcap && freem) {
                        const int s0 = __builtin_ffs((int)freem) - 1;
                        if (lane == 0) {
                            const int old = slot_id[s0];
                            if (old) v.slot_set(old, NONE8);
                            slot_id[s0] = id0;
                            v.slot_set(id0, s0);
                        }
                        needmask |= 1u << s0;
                        const St st0 = st_load(v, id0, x, lane < D && (GO || v.kind == K_GAUSSIAN));
                        cache_fill(v, k, s0, st0.a, st0.b, v.cn_get(id0));
                        row = s0;
                    } else {
                        if (lane == 0) v.xid_set(nx, id0);
                        row = NS + nx;
                        nx += 1;
                    }
                    nneed_new += 1;
                    if (missing && id == id0) { j = row; missing = false; }
                    PM2_WAVE_BARRIER();
                }
                if (lane < N) itemj[r * N + lane] = (u16)j;
            }
        }
        (void)nneed_new;
        // slots that are no longer reachable are given up (their ids leave the map): the cache holds what the step reads
        {
            const unsigned dead = ~needmask & ((1u << NS) - 1u);
            if (lane < NS && ((dead >> lane) & 1u)) { const int old = slot_id[lane]; if (old) { v.slot_set(old, NONE8); slot_id[lane] = 0; } }
        }
        const int nneed = __builtin_popcount(needmask) + nx;
        if (lane == 0) { dsc[DS_NX] = nx; dsc[DS_NNEED] = nneed; dsc[DS_NEEDMASK] = (int)needmask; }
        PM2_WAVE_BARRIER();
        }
        PHD(1);
        // -- A2: the per-feature terms, lane = feature.  Gaussian: cached clusters need one log per feature (gaussian_cluster.jl:46-48).
        // Integer types: the statistics of all needed slots from the pool (the loads in flight together), then the table look-ups
        if ((GO || v.kind == K_GAUSSIAN)) {
#pragma nounroll
            for (int s = 0; s < NS; ++s) {
                if ((needmask >> s) & 1u) {
                    const double mu = c_mu[s], lam = c_lam[s];
                    const double nd_ = (double)lds<int>(v.base + L.slot_cn)[s];
                    const double dd = x - mu;
                    const double tb = (0.5 * nd_ + 1.0) * log(1.0 + (1.0 / (nd_ + 1.0)) * (dd * dd) * lam);
                    if (lane < D) v.tb_row(s)[lane] = tb;
                }
            }
        } else {
            St sv[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) sv[s] = st_load(v, slot_id[s], x, on && ((needmask >> s) & 1u));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (on && ((needmask >> s) & 1u)) {
                    double ta_, tb_;
                    st_terms(v, k, lds<int>(v.base + L.slot_cn)[s], sv[s], x, ta_, tb_);       // (Categorical: the first term was stored when the slot was filled)
                    v.tb_row(s)[lane] = tb_;
                }
            }
        }
        PHD(2);
        // -- A3: ordered sums, one lane per cluster row (calc_logprob's loop, same terms, same order)
        double *lp = v.lp();
        PM2_WAVE_BARRIER();
        if (lane < NS && ((needmask >> lane) & 1u))
            lp[lane] = ordered_sum(GO ? (int)K_GAUSSIAN : v.kind, v.ta_row(lane), v.tb_row(lane), D, fl, dsc[DS_NFLAG] == D, lds<double>(v.base + L.slot_g)[lane]);
        PHD(3);
        // uncached reachable clusters, XR per round: statistics from the pool, both terms on the fly into the tb rows (the cached
        // clusters' sums are done with them): rows 2j, 2j + 1 for the j-th cluster of the round; then one lane per cluster adds
        for (int e0 = 0; e0 < nx; e0 += XR) {
            PM2_WAVE_BARRIER();
            {
                int idj[XR], cnj[XR];
                St stj[XR];
#pragma unroll
                for (int j = 0; j < XR; ++j) {
                    idj[j] = (e0 + j < nx) ? v.xid_get(e0 + j) : 0;
                    cnj[j] = idj[j] ? v.cn_get(idj[j]) : 0;
                    stj[j] = st_load(v, idj[j], x, on && idj[j] != 0);
                }
#pragma unroll
                for (int j = 0; j < XR; ++j) {
                    if (on && idj[j]) {
                        double ta_, tb_;
                        st_terms(v, k, cnj[j], stj[j], x, ta_, tb_);
                        v.tb_row(2 * j)[lane] = ta_;
                        v.tb_row(2 * j + 1)[lane] = tb_;
                    }
                }
            }
            PM2_WAVE_BARRIER();
            if (lane < XR && e0 + lane < nx) {
                const int id = v.xid_get(e0 + lane);
                const double g0 = ((GO || v.kind == K_GAUSSIAN)) ? (double)dsc[DS_NFLAG] * PM2_G(const double, d.gtab)[v.cn_get(id)] : 0.0;
                v.lp_set(NS + e0 + lane, ordered_sum(GO ? (int)K_GAUSSIAN : v.kind, v.tb_row(2 * lane), v.tb_row(2 * lane + 1), D, fl, false, g0));
            }
        }
        PM2_WAVE_BARRIER();
        PHD(4);
        // -- A4: mutation CDF per particle class (:231-248): lanes = (class, label); max / cumsum / normalise through a per-wave
        // exchange area.  The cumsum follows Julia's accumulate_pairwise!: c[n] = e[0] + (e[1] + ... + e[n]) (n < 128).
        {
            auto pik = PM2_G(const double, ap->Pi) + ((size_t)chain * K + k) * N;
            double *wv = lds<double>(v.trb + L.tr_tb);          // (the tb rows are dead: the sums are done)
            const int G = 64 / N;
            const int g = lane / N, nn = lane - g * N;
            const int gbase = (g < G) ? g * N : 0;
            for (int r0 = 0; r0 < ncls; r0 += G) {
                const int r = r0 + g;
                const bool valid = (g < G) && (r < ncls);
                double val = 0.0;
                if (valid) val = v.lp_get(itemj[r * N + nn]);
                wv[lane] = val;
                PM2_WAVE_BARRIER();
                double m = val;
                {
                    int j = 0;
                    for (; j + 4 <= N; j += 4) {      // four LDS reads in flight
                        const double t0 = wv[gbase + j], t1 = wv[gbase + j + 1], t2 = wv[gbase + j + 2], t3 = wv[gbase + j + 3];
                        m = (t0 > m) ? t0 : m; m = (t1 > m) ? t1 : m; m = (t2 > m) ? t2 : m; m = (t3 > m) ? t3 : m;
                    }
                    for (; j < N; ++j) { const double t = wv[gbase + j]; m = (t > m) ? t : m; }
                }
                double e = val - m;
                e = exp(e);
                e = e * (valid ? pik[nn] : 0.0);
                wv[64 + lane] = e;
                PM2_WAVE_BARRIER();
                const double e0v = wv[64 + gbase];
                double s_ = 0.0;
                {
                    int j = 1;
                    for (; j + 4 <= N; j += 4) {      // loads first, then the ordered adds
                        const double t0 = wv[64 + gbase + j], t1 = wv[64 + gbase + j + 1], t2 = wv[64 + gbase + j + 2], t3 = wv[64 + gbase + j + 3];
                        if (j <= nn) s_ = (j == 1) ? t0 : s_ + t0;
                        if (j + 1 <= nn) s_ = s_ + t1;
                        if (j + 2 <= nn) s_ = s_ + t2;
                        if (j + 3 <= nn) s_ = s_ + t3;
                    }
                    for (; j < N; ++j) { const double t = wv[64 + gbase + j]; if (j <= nn) s_ = (j == 1) ? t : s_ + t; }
                }
                const double c = (nn == 0) ? e : e0v + s_;
                PM2_WAVE_BARRIER();
                wv[lane] = c;
                PM2_WAVE_BARRIER();
                const double fN = wv[gbase + N - 1];
                const double cd = c / fN;
                // one-hot to working precision?  every uniform is an odd multiple of 2^-53, so a label whose CDF is < 2^-53 is never
                // chosen and one whose CDF is 1.0 always stops the search: the draw (:252-260) is then the same label for every u
                const u64 m_one = PM2_BALLOT(valid && (cd == 1.0 || nn == N - 1));
                const u64 m_tiny = PM2_BALLOT(valid && cd < 0x1p-53);
                if (valid) {
                    cdf_set(v, r, nn, cd);
                    if (nn == N - 1) {
                        cdf_set(v, r, N, log(fN) + m);
                        const u64 grp = (N == 64) ? ~0ull : (((1ull << N) - 1ull) << gbase);
                        const int nstar = pm2_ffs64((m_one & grp) >> gbase) - 1;
                        const u64 below = (nstar == 0) ? 0ull : ((1ull << nstar) - 1ull);
                        const bool onehot = (((m_tiny & grp) >> gbase) & below) == below;
                        cdf_set(v, r, N + 1, onehot ? (double)nstar : -1.0);
                    }
                }
                PM2_WAVE_BARRIER();
            }
        }
        if (lane == 0) dsc[DS_NS0] = ns0_cur;             // reference trajectory (:262), fetched a step ago
        PHD(15);
        (void)pos;
    }

    // ---- phase C (owner wave): what the draws of this step mean for the tables -- copy-on-write decisions (:275-299), class ids
    //      (:266-272), column splits (:301-308), sufficient statistics (:297,:300) ------------------------------------------------------
    PM2_DEV bool phase_c(int k, double x, long long pos)
    {
        const DV v = view(k);
        const auto &d = ap->ds[k];
        const int D = d.D;
        int *dsc = v.dsc();
        const u8 *fl = flk(k);
        const int maxid = PM2_UNI(dsc[DS_MAXID]), ncol = PM2_UNI(dsc[DS_NCOL]), ncls = PM2_UNI(dsc[DS_NCLS]);
        u16 *clist = lds<u16>(v.base + L.clist);
        u16 *klist = lds<u16>(v.base + L.klist);
        u16 *kval = lds<u16>(v.base + L.kval);          // (class ids are at most P)
        u8 *krep = lds<u8>(v.base + L.krep);
        PHD(11);
        PHC(0);
        // the chosen clusters and the touched keys, from the bitmaps the particle phase marked, as dense lists
        int nd = 0, nk = 0, nd_low = 0;
        {
            u64 *cbm = lds<u64>(v.base + L.cbm), *kbm = lds<u64>(v.base + L.kbm);
            const u64 below = (1ull << lane) - 1ull;
            for (int w = 0; w < (v.idcap + 63) / 64; ++w) {
                const u64 bits = cbm[w];
                if ((bits >> lane) & 1ull) clist[nd + pm2_popc64(bits & below)] = (u16)(w * 64 + lane);
                nd += pm2_popc64(bits);
            }
            const int ndx = PM2_UNI(dsc[DS_NDX]);
            nd_low = nd;                      // (entries nd_low.. are the overflow list in the arena: ids beyond the LDS tables)
            nd += ndx;
            for (int w = 0; w < (CLS * N + 63) / 64; ++w) {
                const u64 bits = kbm[w];
                const int at = nk + pm2_popc64(bits & below);
                if (((bits >> lane) & 1ull) && at < L.kcap) klist[at] = (u16)(w * 64 + lane);
                nk += pm2_popc64(bits);
            }
            PM2_WAVE_BARRIER();
            for (int w = lane; w < (v.idcap + 63) / 64; w += 64) cbm[w] = 0;
            for (int w = lane; w < (CLS * N + 63) / 64; w += 64) kbm[w] = 0;
        }
        if (nk > L.kcap) {                  // more touched (class, label) keys than the lists hold: as good as too many classes
            if (lane == 0) { sc()[SC_FAIL] = 4; sc()[SC_TMP1] = nk; dsc[DS_HALT] = 1; dsc[DS_ND] = 0; dsc[DS_FOLLOW] = 0; }
            return false;
        }
        auto chosen = [&](int e) -> int { if (e < nd_low) return (int)clist[e]; return PM2_G(const int, v.ar.dl())[e - nd_low]; };
        unsigned *minp = lds<unsigned>(v.base + L.minp);
        u64 *bmc = lds<u64>(v.base + L.bmc), *bmf = lds<u64>(v.base + L.bmf);
        // the statistics of the first chosen cluster (the only one in most steps), on their way while the bookkeeping runs
        const St pf = st_load(v, nd > 0 ? chosen(0) : 0, x, nd > 0 && lane < D);
        PHD(12);
        PHC(1);
        // -- C0: the step of a settled chain, most of the time: one class, every particle drew the same label and the same cluster,
        // all references of that cluster were chosen (so it is updated in place, :286-290), the (class, label) key is known and keeps
        // the class its value: no clone, no new class, no column changes -- the statistics, the cache and the idle state of the census
        if (nd == 1 && nk == 1 && ncls == 1 && ap->q1 == 0) {
            const int c = chosen(0), key = klist[0];
            const int v0 = lds<u16>(v.base + L.nidv)[key];
            if (v.ncop_get(c) == v.counts_get(c) && v0 > 0 && v0 == lds<int>(v.base + L.clsval)[0]) {
                const int nnew = v.cn_get(c) + 1;
                const int s0 = v.slot_of(c);
                St st = pf;
                st_add_store(v, c, st, x, nnew, lane < D && fl[lane]);
                if (s0 != NONE8) cache_fill(v, k, s0, st.a, st.b, nnew, true);
                for (int cc = lane; cc < ncol; cc += 64) v.cmask_set(cc, 0);
                PM2_WAVE_BARRIER();
                if (lane == 0) {
                    v.cn_set(c, nnew); v.ncop_set(c, 0); v.firstp_set(c, 0x7fffffff);
                    minp[key] = INFU;
                    long long *st = stat();
                    pm2_atomic_add((u64 *)&st[0], (u64)maxid);
                    pm2_atomic_add((u64 *)&st[4], (u64)1);
                    pm2_atomic_add((u64 *)&st[5], (u64)1);
                    pm2_atomic_max((u64 *)&st[3], (u64)maxid);
                    long long *w_ = wk(k);
                    w_[WK_EVAL] += dsc[DS_NNEED]; w_[WK_UPD] += 1;
                    dsc[DS_NDX] = 0; dsc[DS_NCLONE] = 0; dsc[DS_FOLLOW] = 0; dsc[DS_DIRTY] = 0; dsc[DS_CHANGED] = 0; dsc[DS_ND] = 0;
                }
                return true;
            }
        }
        // Nothing below changes the chain's state before the step is known to fit this kernel's tables (pool capacity, 16-bit ids,
        // CLS particle classes): a dataset whose step does not fit stops here with its state as the observation found it, and the
        // general kernel carries the chain on from this observation (hand_over).
        // -- C1: clone or in place (:286-299): a chosen cluster all of whose references were chosen is updated in place
        PHC(2);
        for (int e0 = 0; e0 < nd; e0 += 64) {
            const int e = e0 + lane;
            if (e < nd) {
                const int c = chosen(e);
                if (v.ncop_get(c) != v.counts_get(c)) { const int fp = v.firstp_get(c); pm2_atomic_or(&bmc[fp >> 6], 1ull << (fp & 63)); }
            }
        }
        PM2_WAVE_BARRIER();
        int nclone = 0;
        for (int w = 0; w < P / 64; ++w) nclone += pm2_popc64(bmc[w]);
        if (maxid + nclone > ap->cap) { if (lane == 0) { sc()[SC_FAIL] = 1; sc()[SC_TMP2] = 1; } return false; }      // PMDI_E_POOL
        // (ids travel as 16-bit values in the LDS tables: beyond that the chain is the general kernel's)
        if (maxid + nclone > 0xFFFF) { if (lane == 0) { sc()[SC_FAIL] = 3; dsc[DS_HALT] = 1; dsc[DS_ND] = 0; dsc[DS_FOLLOW] = 0; } return false; }
        // -- C2: class ids of the next step (:266-272): a (class, label) key met for the first time in this Gibbs iteration gets
        // the next id in particle order (curr_id restarts at 0 every step: Q1), others reuse the id stored under the key
        PHC(3);
        for (int j0 = 0; j0 < nk; j0 += 64) {
            const int j = j0 + lane;
            if (j < nk) {
                const int key = klist[j];
                const int v0 = (ap->q1 == 1) ? 0 : (int)lds<u16>(v.base + L.nidv)[key];
                if (v0 <= 0) { const int pf = (int)(minp[key] >> 16); pm2_atomic_or(&bmf[pf >> 6], 1ull << (pf & 63)); }
            }
        }
        PM2_WAVE_BARRIER();
        for (int j0 = 0; j0 < nk; j0 += 64) {
            const int j = j0 + lane;
            if (j < nk) {
                const int key = klist[j];
                const int r = key / N, ns = key - r * N;
                int v0 = (ap->q1 == 1) ? 0 : (int)lds<u16>(v.base + L.nidv)[key];
                if (v0 <= 0) v0 = 1 + popc_below64(bmf, (int)(minp[key] >> 16));     // curr_id += 1 (:267-269); stored below, once the step is known to fit
                (void)r; (void)ns;
                kval[j] = (u16)v0;
            }
        }
        PM2_WAVE_BARRIER();
        // classes of the next step: one per distinct value, leader = lowest first particle, slots in leader order
        PHC(4);
        int nrep = 0;
        int kv_r = 0;                    // nk <= 64: lane j keeps its key's class value, first particle | column, and representative flag
        unsigned mp_r = INFU;
        bool rep_r = false;
        if (nk <= 64) {
            if (lane < nk) { kv_r = kval[lane]; mp_r = minp[klist[lane]]; }
            rep_r = lane < nk;
            for (int j2 = 0; j2 < nk; ++j2) {
                const int v2 = readlane_i(kv_r, j2);
                const unsigned m2 = (unsigned)readlane_i((int)mp_r, j2);
                if (v2 == kv_r && m2 < mp_r) rep_r = false;
            }
            if (lane < nk) krep[lane] = rep_r ? (u8)1 : (u8)0;
            nrep = pm2_popc64(PM2_BALLOT(rep_r));
        } else
        for (int j0 = 0; j0 < nk; j0 += 64) {
            const int j = j0 + lane;
            bool rep = false;
            if (j < nk) {
                const int v0 = kval[j];
                const unsigned mp = minp[klist[j]];
                rep = true;
                for (int j2 = 0; j2 < nk; ++j2) if (kval[j2] == v0 && minp[klist[j2]] < mp) rep = false;
                krep[j] = rep ? (u8)1 : (u8)0;
            }
            nrep += pm2_popc64(PM2_BALLOT(rep));
        }
        if (nrep > CLS) {                                                                                        // too many particle classes
            if (lane == 0) { sc()[SC_FAIL] = 4; sc()[SC_TMP1] = nrep; dsc[DS_HALT] = 1; dsc[DS_ND] = 0; dsc[DS_FOLLOW] = 0; }
            return false;
        }
        PM2_WAVE_BARRIER();
        // -- the step fits: from here on the tables change.  Targets, reference counts and sizes of the chosen clusters (:290-294) ...
        for (int e0 = 0; e0 < nd; e0 += 64) {
            const int e = e0 + lane;
            if (e < nd) {
                const int c = chosen(e);
                const int ncp = v.ncop_get(c), fp = v.firstp_get(c);
                const int cnt_c = v.counts_get(c);
                const bool needs = ncp != cnt_c;
                const int t = needs ? maxid + 1 + popc_below64(bmc, fp) : c;       // first-appearance order over the particles (:290-292)
                v.tgt_set(c, t);
                const int nnew = v.cn_get(c) + 1;
                if (needs) { v.counts_set(c, cnt_c - ncp); v.counts_set(t, ncp); }         // (:293-294)
                v.cn_set(t, nnew);
            }
        }
        // ... and new_id under the keys met for the first time (:267-269)
        for (int j0 = 0; j0 < nk; j0 += 64) {
            const int j = j0 + lane;
            if (j < nk) {
                const int key = klist[j];
                const int r = key / N, ns = key - r * N;
                const int v_old = (ap->q1 == 1) ? 0 : (int)lds<u16>(v.base + L.nidv)[key];
                const int v0 = kval[j];
                if (v_old <= 0) {
                    if (ap->q1 == 0) PM2_G(int, v.ar.newid())[(size_t)(lds<int>(v.base + L.clsval)[r] - 1) * N + ns] = v0;
                    dsc[DS_CHANGED] = 1;                                            // (new_id changed under a key of this class)
                }
                if (v0 != lds<int>(v.base + L.clsval)[r]) dsc[DS_CHANGED] = 1;       // (the class does not map to itself)
            }
        }
        PM2_WAVE_BARRIER();
        // -- C3: column splits (:301-308): particles of one column that chose the same label move together; the group whose chosen
        // cluster was cloned takes a copy of the column with that entry replaced -- or the column itself when nobody stays behind
        PHC(5);
        int ncol_new = ncol;
        if (nclone == 0) {                                   // nothing was cloned: no column changes, only the chosen-label masks go back to idle
            for (int cc = lane; cc < ncol; cc += 64) v.cmask_set(cc, 0);
        } else
        for (int c0 = 0; c0 < ncol; c0 += 64) {
            const int cc = c0 + lane;
            u64 wm = 0;
            bool keeper = false;
            if (cc < ncol) {
                u64 cm = v.cmask_get(cc);
                v.cmask_set(cc, 0);
                while (cm) {
                    const int ns = pm2_ffs64(cm) - 1;
                    cm &= cm - 1;
                    const int c = v.tab_get(cc, ns);
                    if (v.tgt_get(c) != c) wm |= 1ull << ns; else keeper = true;
                }
            }
            const int nw = pm2_popc64(wm);
            const int nnew = keeper ? nw : (nw > 0 ? nw - 1 : 0);
            const int inpl = (!keeper && nw > 0) ? pm2_ffs64(wm) - 1 : -1;
            int tot;
            const int base = ncol_new + wave_excl_scan_i(nnew, tot);
            if (cc < ncol) { v.wmask_set(cc, wm); v.cbi_set(cc, (base << 8) | (inpl + 1)); }
            // the copies, then the in-place entries (the copies read the columns as they were).  The copies of the chunk are independent
            // of each other: they are listed (source column, label) in the dataset's transient rows -- dead in this phase -- and made
            // 64 / N at a time, lanes = (copy, label); only a chunk with more copies than the list holds takes the column-by-column loop
            unsigned *jobs = lds<unsigned>(v.trb + L.tr_tb);
            const int jobcap = L.tr_stride / 4;
            if (tot <= jobcap) {
                if (cc < ncol && nnew > 0) {
                    int idx = base - ncol_new;
                    u64 wm0 = wm;
                    while (wm0) {
                        const int ns = pm2_ffs64(wm0) - 1;
                        wm0 &= wm0 - 1;
                        if (ns != inpl) jobs[idx++] = ((unsigned)cc << 8) | (unsigned)ns;
                    }
                }
                PM2_WAVE_BARRIER();
                const int G = 64 / N;
                const int g = lane / N, nn = lane - g * N;
                for (int j0 = 0; j0 < tot; j0 += G) {
                    if (g < G && j0 + g < tot) {
                        const unsigned job = jobs[j0 + g];
                        const int cs = (int)(job >> 8), ns = (int)(job & 0xffu);
                        int val = v.tab_get(cs, nn);
                        if (nn == ns) val = v.tgt_get(val);
                        v.tab_set(ncol_new + j0 + g, nn, val);
                    }
                }
                PM2_WAVE_BARRIER();                            // (the copies have read the columns as they were)
                if (cc < ncol && inpl >= 0) v.tab_set(cc, inpl, v.tgt_get(v.tab_get(cc, inpl)));
            } else {
            u64 m;
            bool todo = wm != 0;
            while ((m = PM2_BALLOT(todo)) != 0) {
                const int l0 = pm2_ffs64(m) - 1;
                const int cs = c0 + l0;
                u64 wm0 = readlane_u64(wm, l0);
                const int base0 = readlane_i(base, l0), inpl0 = readlane_i(inpl, l0);
                int rnk = 0;
                while (wm0) {
                    const int ns = pm2_ffs64(wm0) - 1;
                    wm0 &= wm0 - 1;
                    if (ns != inpl0) {
                        const int newc = base0 + rnk - (inpl0 >= 0 ? 1 : 0);
                        const int t = v.tgt_get(v.tab_get(cs, ns));
                        if (lane < N) v.tab_set(newc, lane, lane == ns ? t : v.tab_get(cs, lane));
                    }
                    rnk += 1;
                }
                PM2_WAVE_BARRIER();                            // (the copies have read the column as it was)
                if (inpl0 >= 0 && lane == 0) v.tab_set(cs, inpl0, v.tgt_get(v.tab_get(cs, inpl0)));
                if (lane == l0) todo = false;
            }
            }
            ncol_new += tot;
        }
        PM2_WAVE_BARRIER();
        // the class list of the next step and every key's class slot; the leader's column after the split
        PHC(6);
        unsigned mpr_r = INFU;            // nk <= 64: the first particle | column of the lane's class representative, and the class slot
        int slot_r = 0;
        if (nk <= 64) {
            const u64 repb = PM2_BALLOT(rep_r);
            for (u64 rb = repb; rb; rb &= rb - 1) {
                const int j2 = pm2_ffs64(rb) - 1;
                const int v2 = readlane_i(kv_r, j2);
                const unsigned m2 = (unsigned)readlane_i((int)mp_r, j2);
                if (v2 == kv_r) mpr_r = m2;
            }
            for (u64 rb = repb; rb; rb &= rb - 1) {
                const int j2 = pm2_ffs64(rb) - 1;
                const unsigned m2 = (unsigned)readlane_i((int)mp_r, j2);
                if (m2 < mpr_r) ++slot_r;
            }
        }
        for (int j0 = 0; j0 < nk; j0 += 64) {
            const int j = j0 + lane;
            if (j < nk) {
                const int key = klist[j];
                const int v0 = kval[j];
                unsigned mpr = mpr_r;
                int slot = slot_r;
                if (nk > 64) {
                    int jr = j;
                    for (int j2 = 0; j2 < nk; ++j2) if (krep[j2] && kval[j2] == v0) jr = j2;
                    mpr = minp[klist[jr]];
                    slot = 0;
                    for (int j2 = 0; j2 < nk; ++j2) if (krep[j2] && minp[klist[j2]] < mpr) ++slot;
                }
                lds<u8>(v.base + L.knew)[key] = (u8)slot;
                if (krep[j]) {
                    const int pl = (int)(mpr >> 16), cl = (int)(mpr & 0xffffu);
                    const int ns = key - (key / N) * N;
                    const u64 wm = nclone ? v.wmask_get(cl) : 0ull;          // (without a clone the column tables were not written)
                    int newc = cl;
                    if ((wm >> ns) & 1ull) {
                        const int bi = v.cbi_get(cl);
                        const int inpl = (bi & 0xff) - 1, base = bi >> 8;
                        const int rnk = pm2_popc64(wm & ((1ull << ns) - 1ull));
                        newc = (inpl >= 0) ? (ns == inpl ? cl : base + rnk - 1) : base + rnk;
                    }
                    lds<int>(v.base + L.clsval)[slot] = v0; lds<int>(v.base + L.clslead)[slot] = pl; lds<int>(v.base + L.leadcol)[slot] = newc;
                }
            }
        }
        PM2_WAVE_BARRIER();
        PHD(13);
        PHC(7);
        int ndefer = 0;
        // -- C4: deepcopy + cluster_add! of every distinct chosen cluster (:297,:300), lane = feature: (Sigma, beta) from the pool (the
        // first cluster's were fetched at the top of the phase), written to the pool; a cached cluster gets its mu, lambda and first
        // term refreshed in the owner wave's registers
        {
            // The owner wave updates the clusters its register cache holds and that are updated in place (it has to refresh mu, lambda and
            // the first term anyway); every other chosen cluster -- uncached, or cloned -- is left to the statistics phase behind the
            // next workgroup barrier, where all four waves share them (help_stats).  Four clusters at a time: table entries, then the
            // statistics and prefix constants (all loads in flight together), then the arithmetic.
            const bool on = lane < D && fl[lane];
            for (int e0 = 0; e0 < nd; e0 += 4) {
                int cj[4], nj[4], sj[4];
                bool mine[4];
                St stj[4];
                double gj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cj[j] = 0; nj[j] = 0; sj[j] = NONE8; mine[j] = false;
                    if (e0 + j < nd) {
                        cj[j] = chosen(e0 + j);
                        sj[j] = v.slot_of(cj[j]);
                        mine[j] = sj[j] != NONE8 && v.tgt_get(cj[j]) == cj[j];
                        if (mine[j]) nj[j] = v.cn_get(cj[j]); else ndefer += 1;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    stj[j] = pf; gj[j] = 0.0;
                    if (mine[j]) {
                        if (e0 + j > 0) stj[j] = st_load(v, cj[j], x, lane < D);
                        if (lane == 0 && (GO || v.kind == K_GAUSSIAN)) gj[j] = PM2_G(const double, d.gtab)[nj[j]];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (mine[j]) {
                        st_add_store(v, cj[j], stj[j], x, nj[j], on);
                        cache_fill(v, k, sj[j], stj[j].a, stj[j].b, nj[j], false, true, gj[j]);
                    }
                }
            }
        }
        PHD(15);
        PHC(8);
        // -- C5: the step's scratch back to its idle state; counters
        PM2_WAVE_BARRIER();
        for (int e0 = 0; e0 < nd; e0 += 64) {
            const int e = e0 + lane;
            if (e < nd) { const int c = chosen(e); v.ncop_set(c, 0); v.firstp_set(c, 0x7fffffff); }
        }
        for (int j0 = 0; j0 < nk; j0 += 64) { const int j = j0 + lane; if (j < nk) minp[klist[j]] = INFU; }
        for (int w = lane; w < P / 64 + 1; w += 64) { bmc[w] = 0; bmf[w] = 0; }
        if (lane == 0) {
            long long *st = stat();
            // (the K owner waves add to the shared counters one after the other in phase order: atomics keep them exact)
            pm2_atomic_add((u64 *)&st[0], (u64)maxid);                   // src/__pmdi.jl:187
            pm2_atomic_add((u64 *)&st[4], (u64)ncls);
            pm2_atomic_add((u64 *)&st[2], (u64)nclone);
            pm2_atomic_add((u64 *)&st[5], (u64)1);
            pm2_atomic_max((u64 *)&st[3], (u64)(maxid + nclone));        // max_id: after the copy-on-write, before any renumbering
            long long *w_ = wk(k);
            w_[WK_EVAL] += dsc[DS_NNEED]; w_[WK_UPD] += nd; w_[WK_CLONE] += nclone; w_[WK_SPLITS] += ncol_new - ncol;
            dsc[DS_MAXID] = maxid + nclone; dsc[DS_NCLS] = nrep; dsc[DS_NCOL] = ncol_new; dsc[DS_NDX] = 0; dsc[DS_NCLONE] = nclone;
            dsc[DS_ND] = ndefer ? nd : 0; dsc[DS_NDLOW] = nd_low;
            // the particles have something to follow when a column was split or written, or when the class slots move: not when the
            // step had one class, one (class, label) key and no clone (the key's class is slot 0 again)
            dsc[DS_FOLLOW] = (nclone != 0 || ncls != 1 || nk != 1) ? 1 : 0;
            // ... and the next step of this dataset reads the same clusters through the same class, under unchanged new_id entries,
            // when on top of that the one key was known and kept its class value: the need set of phase A stands as it is
            dsc[DS_DIRTY] = (nclone != 0 || ncls != 1 || nk != 1 || dsc[DS_CHANGED] != 0) ? 1 : 0;
            dsc[DS_CHANGED] = 0;
        }
        (void)pos;
        return true;
    }

    // ---- statistics phase (all waves, behind the barrier that ends the bookkeeping phase): deepcopy + cluster_add! (:297,:300) of the
    //      chosen clusters the owner waves left -- every one that is not (cached and updated in place).  The batches of all
    //      datasets are dealt round-robin to the four waves: a chain whose particles sit on dozens of private columns in ONE dataset
    //      updates dozens of clusters per step there, and the waves of the quiet datasets would only wait for it.  Returns whether
    //      there was anything to do (uniform): the caller then closes the phase with a workgroup barrier.
    PM2_DEV bool help_stats(int i_cur)
    {
        constexpr int HB = 2;                  // clusters per batch (two: the phase runs with every lane's particle state live)
        int item = 0;
        bool any = false;
#pragma nounroll
        for (int k = 0; k < K; ++k) {
            const DV v = view(k);
            const int *dsc = v.dsc();
            const int nd = PM2_UNI(dsc[DS_ND]);
            if (nd == 0) continue;
            any = true;
            const int nd_low = PM2_UNI(dsc[DS_NDLOW]);
            const auto &d = ap->ds[k];
            const int D = d.D;
            const u8 *fl = flk(k);
            const bool on = lane < D && fl[lane];
            const u16 *clist = lds<u16>(v.base + L.clist);
            auto sb = PM2_G(double, v.ar.sb());
            double x = 0.0;
            bool have_x = false;
            for (int e0 = 0; e0 < nd; e0 += HB, ++item) {
                if (item % NW != wave) continue;
                if (!have_x) { x = obs_x(k, i_cur); have_x = true; }
                int cj[HB], tj[HB], nj[HB];
                bool mine[HB];
                double sgj[HB], btj[HB];
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    cj[j] = 0; tj[j] = 0; nj[j] = 0; mine[j] = false;
                    if (e0 + j < nd) {
                        const int e = e0 + j;
                        cj[j] = (e < nd_low) ? (int)clist[e] : PM2_G(const int, v.ar.dl())[e - nd_low];
                        tj[j] = v.tgt_get(cj[j]);
                        mine[j] = !(v.slot_of(cj[j]) != NONE8 && tj[j] == cj[j]);
                        if (mine[j]) nj[j] = v.cn_get(tj[j]);
                    }
                }
                if ((GO || v.kind == K_GAUSSIAN)) {
#pragma unroll
                    for (int j = 0; j < HB; ++j) {
                        sgj[j] = 0.0; btj[j] = 0.5;
                        if (mine[j] && lane < D) { sgj[j] = sb[((size_t)cj[j] * D + lane) * 2]; btj[j] = sb[((size_t)cj[j] * D + lane) * 2 + 1]; }
                    }
#pragma unroll
                    for (int j = 0; j < HB; ++j) {
                        if (mine[j]) {
                            double sg = sgj[j], bt = btj[j];
                            if (on) pmdi_arith::gauss_add_sb(x, nj[j], sg, bt);
                            if (lane < D && (on || tj[j] != cj[j])) { sb[((size_t)tj[j] * D + lane) * 2] = sg; sb[((size_t)tj[j] * D + lane) * 2 + 1] = bt; }
                        }
                    }
                } else if ((!GO && v.kind == K_CATEGORICAL)) {
                    // deepcopy: every level's count of the lane's feature; cluster_add!: the observed level's (categorical_cluster.jl:43-51)
                    auto cn_ = PM2_G(int, v.ar.cnt());
                    const int Lc = v.Lc;
#pragma unroll
                    for (int j = 0; j < HB; ++j) {
                        if (mine[j] && lane < D) {
                            const size_t so = ((size_t)cj[j] * D + lane) * Lc, to = ((size_t)tj[j] * D + lane) * Lc;
                            const int xl = (int)x - 1;
                            const int cx = cn_[so + xl];
                            if (tj[j] != cj[j]) for (int l = 0; l < Lc; ++l) cn_[to + l] = cn_[so + l];
                            if (on) cn_[to + xl] = cx + 1;
                        }
                    }
                } else {
                    auto nb_ = PM2_G(long long, v.ar.nbs());
#pragma unroll
                    for (int j = 0; j < HB; ++j)
                        if (mine[j] && lane < D && (on || tj[j] != cj[j]))
                            nb_[(size_t)tj[j] * D + lane] = nb_[(size_t)cj[j] * D + lane] + (on ? (long long)x : 0ll);       // negbinom_cluster.jl:43-51
                }
            }
        }
        return any;
    }

    // ---- the census of a lane's draw (`mult` of its particles drew the same class, column and label; p the first of them),
    //      aggregated over the lanes of the wave that drew the same.  Nothing here waits for an answer: the chosen clusters and the
    //      touched (class, label) keys are marked in bitmaps that phase C reads back; only a cluster id beyond the LDS tables
    //      (rare) takes the returning path into the overflow list. ----------------------------------------------------------------
    PM2_DEV void census(const DV &v, bool active, int mult, int r, int cl, int ns, int c, int p)
    {
        int *dsc = v.dsc();
        const int key3 = ((r * P + cl) * N + ns) * 16 + mult;
        const u64 act = PM2_BALLOT(active);
        if (act == 0) return;
        // the whole wave drew the same (the rule in a settled chain): one lane speaks for it; otherwise every lane for itself -- the
        // LDS atomics sort out equal addresses faster than an election loop over the distinct draws would
        const int l0 = pm2_ffs64(act) - 1;
        const int k0 = readlane_i(key3, l0);
        const bool uni = PM2_BALLOT(active && key3 == k0) == act;
        if (uni ? (lane == l0) : active) {
            const int cnt = uni ? pm2_popc64(act) * mult : mult;
            // chosen cluster: copies and first particle (:279)
            if (c < v.idcap) {
                pm2_atomic_add(lds<int>(v.base + L.ncop) + c, cnt);
                pm2_atomic_min(lds<int>(v.base + L.firstp) + c, p);
                pm2_atomic_or(lds<u64>(v.base + L.cbm) + (c >> 6), 1ull << (c & 63));
            } else {
                const int old = pm2_atomic_min(v.ar.firstp() + c, p);       // (idle value INF: the first toucher lists the cluster)
                pm2_atomic_add(v.ar.ncop() + c, cnt);
                if (old == 0x7fffffff) { const int idx = pm2_atomic_add(&dsc[DS_NDX], 1); PM2_G(int, v.ar.dl())[idx] = c; }
            }
            // (class, label) key: first particle, with its column
            const int key = r * N + ns;
            pm2_atomic_min(&lds<unsigned>(v.base + L.minp)[key], ((unsigned)p << 16) | (unsigned)cl);
            pm2_atomic_or(lds<u64>(v.base + L.kbm) + (key >> 6), 1ull << (key & 63));
            // labels chosen on this column
            v.cmask_or(cl, 1ull << ns);
        }
    }

    // ---- Julia's accumulate_pairwise! over the P weights held PPL per lane (src/misc.jl:29): returns c[p] for the lane's particles --
    // leaves: [1, 128) and the 64-element blocks above it (P a power of two >= 256); a leaf's running sums are a serial chain
    // that walks from lane to lane; the carries come from replaying the recursion over the leaf totals (one lane).
    PM2_DEV void cumsum_pairwise(const double (&w)[PPL], double (&c)[PPL])
    {
        constexpr int LPB = 64 / PPL;                // lanes per 64-element block
        double sl[PPL];
#pragma unroll
        for (int u = 0; u < PPL; ++u) sl[u] = 0.0;
        const int blk = (tid * PPL) >> 6;            // 64-element block of the lane's particles
        const int lib = tid % LPB;                   // lane inside the block
        double *tot = lds<double>(L.leaf_tot), *car = lds<double>(L.leaf_carry);
        double acc = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: every block but block 1 (whose chain continues block 0's: one leaf of 127 elements); pass 1: block 1
            const bool mine = (pass == 0) ? (blk != 1) : (blk == 1);
            double carry_in = 0.0;
            if (pass == 1) carry_in = tot[0];
            for (int t = 0; t < LPB; ++t) {
                const double cin = prev_lane_d(acc);
                if (mine && lib == t) {
                    double s = (t == 0) ? carry_in : cin;
#pragma unroll
                    for (int u = 0; u < PPL; ++u) {
                        const int p = tid * PPL + u;
                        if (p == 0) continue;                                      // c[1] = v1 is the recursion's seed
                        const bool first = (p == 1) || ((p & 63) == 0 && p >= 128);
                        s = first ? w[u] : s + w[u];
                        sl[u] = s;
                    }
                    acc = s;
                }
            }
            if (mine && lib == LPB - 1) tot[blk] = acc;
            PM2_BARRIER();
        }
        // The carries.  Above the leaves the recursion is a fixed binary tree over the 64-element blocks (blocks 0 and 1 are the one leaf
        // [1, 128)): a node hands its carry s to its left child and s + total(left) to its right child, and total(node) = total(left) +
        // total(right).  Level totals first (log2(P / 64) - 1 levels, lanes = nodes), then every lane adds up its own block's path.
        const int nb = P >> 6;
        double *lv = car;                              // lv[off(h) + g]: total of blocks [g 2^h, (g + 1) 2^h), h >= 1; off(h) = nb - (nb >> (h - 1))
        if (wave == 0) {
            if (tid == 0) lds<double>(L.red)[121] = w[0];                    // c[1] = v1, the carry of everything on the left spine
            for (int h = 1; (nb >> h) >= 2; ++h) {
                const int cnt = nb >> h, off = nb - (nb >> (h - 1)), offp = (h >= 2) ? nb - (nb >> (h - 2)) : 0;
                if (lane < cnt) {
                    double val;
                    if (h == 1) val = (lane == 0) ? tot[1] : tot[2 * lane] + tot[2 * lane + 1];
                    else val = lv[offp + 2 * lane] + lv[offp + 2 * lane + 1];
                    lv[off + lane] = val;
                }
                PM2_WAVE_BARRIER();
            }
        }
        PM2_BARRIER();
        double carry = lds<double>(L.red)[121];
        if (blk >= 2) {
            const int j = 32 - __builtin_clz((unsigned)blk);                  // 2^(j-1) <= blk < 2^j: the right child of spine node j
            carry = carry + lv[nb - (nb >> (j - 2))];                          // ... whose carry is v1 + total(blocks [0, 2^(j-1)))
            for (int h = j - 2; h >= 0; --h)
                if ((blk >> h) & 1) {
                    const int g = (blk >> h) - 1;                              // the left sibling at level h
                    carry = carry + (h == 0 ? tot[g] : lv[nb - (nb >> (h - 1)) + g]);
                }
        }
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const int p = tid * PPL + u;
            c[u] = (p == 0) ? w[u] : carry + sl[u];
        }
    }

    // ---- draw_partstar + gather + compact renumbering (src/misc.jl:27-47, src/pmdi.jl:318-340) ---------------------------------------
    PM2_DEV void resample(long long pos, double mx)
    {
        PHR(0);
        double w[PPL];
#pragma unroll
        for (int u = 0; u < PPL; ++u) w[u] = exp(lw[u] - mx);        // pprob before the cumsum (src/misc.jl:29), as calc_ESS formed it
        const double u01 = pmdi_arith::uniform01(seed, iter, (unsigned)pos, 0, 0, SITE_RESAMPLE_U);
        const double usl = pmdi_arith::uniform01(seed, iter, (unsigned)pos, 0, 0, SITE_RESAMPLE_SLOT);
        double c[PPL];
        PHR(1);
        cumsum_pairwise(w, c);
        PHR(2);
        double *utab = lds<double>(L.tr);
        u16 *jtab = lds<u16>(L.tr + L.rs_jtab), *raw = lds<u16>(L.tr + L.rs_raw), *anc = lds<u16>(L.tr + L.rs_anc);
        double *red = lds<double>(L.red);
        if (tid == T - 1) red[0] = c[PPL - 1];
        // u += 1/particles by repeated addition (:34), exactly: h = 1/P is a power of two, so inside a binade every u + h is exact
        // and only the additions that cross into the next binade round.  The lane replays the crossings up to its first slot
        // and adds h for the following ones as the reference does.
        {
            const double h = 1.0 / (double)P;
            const int j = tid * PPL;
            double u = u01 / (double)P;
            int done = 0;
            while (done < j) {
                int e;
                (void)frexp(u, &e);                          // u in [2^(e-1), 2^e)
                const double B = ldexp(1.0, e);
                const double xx = (B - u) * (double)P;       // exact: B - u (same binade) and the power-of-two scale
                double m = floor(xx);
                if (m == xx) m -= 1.0;                       // largest m with u + m*h < B
                if (m > (double)(j - done)) m = (double)(j - done);
                if (m >= 1.0) { u = u + m * h; done += (int)m; }
                if (done < j) { u = u + h; done += 1; }
            }
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) { utab[j + uu] = u; u = u + h; }
        }
        PM2_BARRIER();
        PHR(3);
        // slots taken up to and including particle p: J_p = #{ j : pprob[p] / last >= u_j } (:32-36)
        {
            const double last = red[0];
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) {
                const double q = c[uu] / last;
                int J = (int)((q - utab[0]) * (double)P) + 1;
                if (J < 0) J = 0;
                if (J > P) J = P;
                while (J < P && q >= utab[J]) ++J;
                while (J > 0 && !(q >= utab[J - 1])) --J;
                jtab[tid * PPL + uu] = (u16)J;
            }
        }
        PM2_BARRIER();
        PHR(4);
        // slot j belongs to the first particle with J_p > j
        {
            // (the lane's slots are consecutive: the answer never moves down, and the same heavy particle usually owns the next slot too)
            int lo = 0;
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) {
                const int j = tid * PPL + uu;
                if (uu == 0 || !((int)jtab[lo] > j)) {
                    int hi = P - 1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)jtab[mid] > j) hi = mid; else lo = mid + 1; }
                }
                raw[j] = (u16)lo;
            }
        }
        int js = (int)(usl * (double)P);                      // shuffle!, partstar[1] = 1, sort! (:43-45)
        if (js >= P) js = P - 1;
        PM2_BARRIER();
        PHR(5);
#pragma unroll
        for (int uu = 0; uu < PPL; ++uu) {
            const int p = tid * PPL + uu;
            anc[p] = (p == 0) ? (u16)0 : (p <= js ? raw[p - 1] : raw[p]);
            lw.set(uu, 1.0);                                  // src/pmdi.jl:319
        }
        PM2_BARRIER();
        // per dataset (:320-340); the u table is dead: its LDS holds the gather tables
        u16 *scol = lds<u16>(L.tr), *mult = lds<u16>(L.tr + P * 2), *cmap = lds<u16>(L.tr + P * 4);
        u8 *scsl = lds<u8>(L.tr + P * 6);
        int *hist = lds<int>(L.tr + L.rs_hist);
#pragma nounroll
        for (int k = 0; k < K; ++k) {
            const DV v = view(k);
            int *dsc = v.dsc();
            const int ncol_old = dsc[DS_NCOL], oldmax = dsc[DS_MAXID], ncls_old = dsc[DS_NCLS];
            PHR(6);
            int ck[PPL], rk[PPL];                           // this dataset's column / class slot of the lane's particles
            col_get(k, ck);
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) rk[uu] = csl_get(k, uu);
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) { const int p = tid * PPL + uu; scol[p] = (u16)ck[uu]; scsl[p] = (u8)rk[uu]; }
            for (int e = tid; e < ncol_old; e += T) mult[e] = 0;
            for (int e = tid; e <= oldmax && e < v.idcap; e += T) hist[e] = 0;
            if (tid < CLSMAX) lds<int>(L.red)[RI_CLSMIN + tid] = 0x7fffffff;
            if (tid == 0) sc()[SC_TMP0] = 0;                 // ids moved by this dataset's renumbering (counted below, barriers away)
            PM2_BARRIER();
            // particle[:, partstar, k], particle_id[partstar, k] (:322-323): a particle takes its ancestor's column index and class
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) {
                const int p = tid * PPL + uu;
                const int an = anc[p];
                ck[uu] = scol[an]; rk[uu] = scsl[an];
            }
            PHR(7);
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) {
                // particles per old column and the lowest particle of every class.  The ancestors are sorted, so the particles of a lane --
                // and mostly of a whole wave -- sit on the same column and class: runs inside the lane are counted once, and a wave whose
                // runs all agree speaks through one lane
                const int cl = ck[uu], r = rk[uu];
                bool start = true;
                if (uu > 0) start = cl != ck[uu - 1] || r != rk[uu - 1];
                int cnt = 1;
                {
                    bool go = true;
#pragma unroll
                    for (int u2 = uu + 1; u2 < PPL; ++u2) { go = go && ck[u2] == cl && rk[u2] == r; cnt += go ? 1 : 0; }
                }
                const u64 act = PM2_BALLOT(start);
                if (act) {
                    const int l0 = pm2_ffs64(act) - 1;
                    const int key = cl * CLSMAX + r;
                    const int key0 = readlane_i(key, l0);             // (every lane reads it: a cross-lane read must not sit behind `start &&`)
                    const bool uni = PM2_BALLOT(start && key == key0) == act;
                    int tot = cnt;
                    if (uni) {
                        tot = 0;
#pragma unroll
                        for (int c_ = 1; c_ <= PPL; ++c_) tot += c_ * pm2_popc64(PM2_BALLOT(start && cnt == c_));
                    }
                    if (uni ? (lane == l0) : start) {
                        // (u16 table, 32-bit LDS atomics: two columns share a word)
                        pm2_atomic_add((unsigned *)mult + (cl >> 1), (unsigned)tot << ((cl & 1) * 16));
                        pm2_atomic_min(&lds<int>(L.red)[RI_CLSMIN + r], tid * PPL + uu);
                    }
                }
            }
            PM2_BARRIER();
            PHR(8);
            // new index of every column that kept a particle
            int ncol_new = 0;
            for (int b = 0; b < ncol_old; b += T) {
                const int cc = b + tid;
                const bool live = cc < ncol_old && mult[cc] != 0;
                const u64 bal = PM2_BALLOT(live);
                if (lane == 0) lds<int>(L.red)[RI_WCNT + wave] = pm2_popc64(bal);
                PM2_BARRIER();
                int basew = ncol_new, tot = 0;
                for (int w_ = 0; w_ < T / 64; ++w_) { const int cnt = lds<int>(L.red)[RI_WCNT + w_]; if (w_ < wave) basew += cnt; tot += cnt; }
                if (live) cmap[cc] = (u16)(basew + pm2_popc64(bal & ((1ull << lane) - 1ull)));
                ncol_new += tot;
                PM2_BARRIER();
            }
            PHR(9);
            // occupancy of every old id = sum over the live columns of (particles on the column) x (entries holding the id) (:338)
            for (int idx = tid; idx < ncol_old * N; idx += T) {
                const int cc = idx / N;
                const int m = mult[cc];
                if (m) {
                    const int id = v.tab_get(cc, idx - cc * N);
                    if (id < v.idcap) pm2_atomic_add(&hist[id], m); else pm2_atomic_add(v.ar.ncop() + id, m);
                }
            }
            PM2_BARRIER();
            PHR(10);
            // sort(unique(particle)) ascending -> 1..U' (:329): ranks of the live ids; the map goes to the tgt table
            int newmax = 0;
            for (int b = 0; b < oldmax; b += T) {
                const int id = 1 + b + tid;
                int occ = 0;
                if (id <= oldmax) occ = (id < v.idcap) ? hist[id] : PM2_G(int, v.ar.ncop())[id];
                const bool live = occ != 0;
                const u64 bal = PM2_BALLOT(live);
                if (lane == 0) lds<int>(L.red)[RI_WCNT + wave] = pm2_popc64(bal);
                PM2_BARRIER();
                int basew = newmax, tot = 0;
                for (int w_ = 0; w_ < T / 64; ++w_) { const int cnt = lds<int>(L.red)[RI_WCNT + w_]; if (w_ < wave) basew += cnt; tot += cnt; }
                if (id <= oldmax) v.tgt_set(id, live ? basew + pm2_popc64(bal & ((1ull << lane) - 1ull)) + 1 : 0);
                newmax += tot;
                PM2_BARRIER();
            }
            // counts and cluster sizes move down with their ids, ascending (:336,:338): read a batch, barrier, write it
            PHR(11);
            for (int b = 0; b < oldmax; b += T) {
                const int id = 1 + b + tid;
                int nid = 0, occ = 0, cnv = 0;
                if (id <= oldmax) {
                    nid = v.tgt_get(id);
                    occ = (id < v.idcap) ? hist[id] : PM2_G(int, v.ar.ncop())[id];
                    cnv = v.cn_get(id);
                    if (id >= v.idcap) PM2_G(int, v.ar.ncop())[id] = 0;
                }
                {
                    const u64 mvb = PM2_BALLOT(nid != 0 && nid != id);
                    if (lane == 0 && mvb) pm2_atomic_add(&sc()[SC_TMP0], pm2_popc64(mvb));
                }
                PM2_BARRIER();
                if (id <= oldmax) {
                    if (nid) { v.counts_set(nid, occ); v.cn_set(nid, cnv); }
                    if (id > newmax) v.counts_set(id, 0);
                }
                PM2_BARRIER();
            }
            // ... and the statistics: clusters[k][i] = deepcopy(clusters[k][id]) for id > i, ascending batches (:336)
            PHR(12);
            if (sc()[SC_TMP0] != 0) {                         // (uniform: written before the barriers of the loop above)
                // one pool row = W words per id: Gaussian D pairs of doubles, Categorical D x L counts, NegBinom D 64-bit sums
                const int D = ap->ds[k].D;
                if ((GO || v.kind == K_GAUSSIAN)) {
                    auto sb = PM2_G(double, v.ar.sb());
                    const long long nitems = (long long)oldmax * D;
                    for (long long b = 0; b < nitems; b += T) {
                        const long long it = b + tid;
                        const int id = 1 + (int)(it / D), q = (int)(it - (long long)(id - 1) * D);
                        const int nid = (it < nitems) ? v.tgt_get(id) : 0;
                        const bool mv = nid != 0 && nid != id;
                        double sg = 0.0, bt = 0.0;
                        if (mv) { sg = sb[((size_t)id * D + q) * 2]; bt = sb[((size_t)id * D + q) * 2 + 1]; }
                        PM2_BARRIER();
                        if (mv) { sb[((size_t)nid * D + q) * 2] = sg; sb[((size_t)nid * D + q) * 2 + 1] = bt; }
                    }
                } else if ((!GO && v.kind == K_CATEGORICAL)) {
                    auto cn_ = PM2_G(int, v.ar.cnt());
                    const int W = D * v.Lc;
                    const long long nitems = (long long)oldmax * W;
                    for (long long b = 0; b < nitems; b += T) {
                        const long long it = b + tid;
                        const int id = 1 + (int)(it / W), q = (int)(it - (long long)(id - 1) * W);
                        const int nid = (it < nitems) ? v.tgt_get(id) : 0;
                        const bool mv = nid != 0 && nid != id;
                        int c_ = 0;
                        if (mv) c_ = cn_[(size_t)id * W + q];
                        PM2_BARRIER();
                        if (mv) cn_[(size_t)nid * W + q] = c_;
                    }
                } else {
                    auto nb_ = PM2_G(long long, v.ar.nbs());
                    const long long nitems = (long long)oldmax * D;
                    for (long long b = 0; b < nitems; b += T) {
                        const long long it = b + tid;
                        const int id = 1 + (int)(it / D), q = (int)(it - (long long)(id - 1) * D);
                        const int nid = (it < nitems) ? v.tgt_get(id) : 0;
                        const bool mv = nid != 0 && nid != id;
                        long long S_ = 0;
                        if (mv) S_ = nb_[(size_t)id * D + q];
                        PM2_BARRIER();
                        if (mv) nb_[(size_t)nid * D + q] = S_;
                    }
                }
            }
            PM2_BARRIER();
            PHR(13);
            // the cache follows the renumbering (same clusters under new ids)
            if (wave == k) {
                int *slot_id = lds<int>(v.base + L.slot_id);
                int olds = 0, news = 0;
                if (lane < NS) { olds = slot_id[lane]; news = olds ? v.tgt_get(olds) : 0; }
                PM2_WAVE_BARRIER();
                if (lane < NS && olds) v.slot_set(olds, NONE8);
                PM2_WAVE_BARRIER();
                if (lane < NS) { slot_id[lane] = news; if (news) v.slot_set(news, lane); }
            }
            // the live columns, compacted and relabelled (:331-337), in rounds of whole columns (a round's targets lie at or below
            // its sources, later rounds' sources above them)
            {
                const int cpr = (2 * T) / N;                       // columns per round: two entries per lane
                for (int c0 = 0; c0 < ncol_old; c0 += cpr) {
                    int nv[2], dc[2], dn[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int idx = h * T + tid;
                        const int cc = c0 + idx / N, nn = idx - (idx / N) * N;
                        nv[h] = 0; dc[h] = -1; dn[h] = nn;
                        if (idx < cpr * N && cc < ncol_old && mult[cc] != 0) { nv[h] = v.tgt_get(v.tab_get(cc, nn)); dc[h] = cmap[cc]; }
                    }
                    PM2_BARRIER();
#pragma unroll
                    for (int h = 0; h < 2; ++h) if (dc[h] >= 0) v.tab_set(dc[h], dn[h], nv[h]);
                    PM2_BARRIER();
                }
            }
#pragma unroll
            for (int uu = 0; uu < PPL; ++uu) ck[uu] = cmap[ck[uu]];
            // classes that kept a particle, in leader order; leaders' columns
            {
                int *redi = lds<int>(L.red);
                int nc2 = 0;
                for (int r = 0; r < ncls_old; ++r) nc2 += (redi[RI_CLSMIN + r] != 0x7fffffff) ? 1 : 0;
                if (tid < ncls_old) {
                    int s = 0;
                    const int mine = redi[RI_CLSMIN + tid];
                    for (int r2 = 0; r2 < ncls_old; ++r2) if (redi[RI_CLSMIN + r2] < mine) ++s;
                    redi[RI_NEWSLOT + tid] = s;                             // (a class that lost every particle is never looked up)
                    redi[RI_CLSVAL + tid] = lds<int>(v.base + L.clsval)[tid];
                }
                PM2_BARRIER();
#pragma unroll
                for (int uu = 0; uu < PPL; ++uu) {
                    const int p = tid * PPL + uu;
                    const int r = rk[uu];
                    const int ns_ = redi[RI_NEWSLOT + r];
                    rk[uu] = ns_;
                    if (redi[RI_CLSMIN + r] == p) {
                        lds<int>(v.base + L.clsval)[ns_] = redi[RI_CLSVAL + r]; lds<int>(v.base + L.clslead)[ns_] = p; lds<int>(v.base + L.leadcol)[ns_] = ck[uu];
                    }
                }
                col_put(k, ck);
#pragma unroll
                for (int uu = 0; uu < PPL; ++uu) csl_put(k, uu, rk[uu]);
                if (tid == 0) {
                    const int moved = sc()[SC_TMP0];
                    dsc[DS_NCLS] = nc2; dsc[DS_NCOL] = ncol_new; dsc[DS_MAXID] = newmax; dsc[DS_DIRTY] = 1;
                    long long *w_ = wk(k);
                    w_[WK_COLS] += ncol_old; w_[WK_MOVED] += moved; w_[WK_MOVE_EVENTS] += moved ? 1 : 0;
                }
            }
            PM2_BARRIER();
        }
        PHR(15);
    }

    // ---- the whole sweep -----------------------------------------------------------------------------------------------------------
    PM2_DEV void run(const SweepArgs *ap_, int chain_)
    {
        ap = (const PM2_CONST SweepArgs *)ap_; chain = chain_;
        const auto &a = *ap;
        tid = PM2_TID(); lane = tid & 63; wave = tid >> 6;
        N = a.N; P = a.P; n = a.n; n1 = a.n1; iter = a.iter;
        seed = a.seed + (unsigned long long)chain;
        const long long t_start = PM2_CLOCK();
        const long long t_wall = a.phase ? PM2_WALLCLOCK() : 0ll;
        const bool owner = wave < K;
        auto order = PM2_G(const int, a.order) + (size_t)chain * n;
        auto logphi = PM2_G(const double, a.logphi) + (size_t)chain * a.npairs;
        if (tid < SC_COUNT) sc()[tid] = 0;
        if (tid < 8) stat()[tid] = 0;
        if (tid < KMAX2 * 8) lds<long long>(L.wk)[tid] = 0;
        if (tid < 16) lds<long long>(L.ph)[tid] = 0;
        // phase timers (PMDI_PHASE_TIMERS): shader-clock totals of lane 0 -- [0] prefix, [1] cluster phase (its own dataset), [2] wait
        // for the other owner waves, [3] particle phase, [4] wait, [5] ESS half + bookkeeping phase, [6] wait, [7] follow-up + ESS
        // decision, [8] resampling, [9] finish, [14] whole sweep
        long long ph_last = 0;
        int ph_cur = 0;
#ifdef PM2_DETAIL_TIMERS
        phd_last = PM2_CLOCK(); phd_cur = 15;
#define PH2(i_) do { } while (0)
#else
#define PH2(i_) do { if (a.phase && tid == 0) { const long long t_ = PM2_CLOCK(); lds<long long>(L.ph)[ph_cur] += t_ - ph_last; ph_last = t_; ph_cur = (i_); } } while (0)
#endif
        if (a.phase && tid == 0) ph_last = PM2_CLOCK();
        PM2_BARRIER();
#pragma unroll
        for (int u = 0; u < PPL; ++u) lw.set(u, a.lw_init);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < NCP; ++j) colp.set(k * NCP + j, 0u);
#pragma unroll
        for (int k = 0; k < NCSW; ++k) cslp.set(k, (csl_t)0);
#pragma unroll
        for (int s = 0; s < NS; ++s) { c_mu.set(s, 0.0); c_lam.set(s, 1.0); }
        PM2_BARRIER();
        if (owner) prefix(wave);
        PM2_BARRIER();
        PH2(1);

        // the observation row of the owner wave's dataset and the reference trajectory's label there, one step ahead (lane = feature);
        // the shuffled order two steps ahead, so that the row's address never waits for it
        double xnext = 0.0;
        int ns0_next = 0;
        int i_next = order[n1 - 1];
        int i_next2 = (n1 < n) ? order[n1] : 0;
        if (owner) {
            xnext = obs_x(wave, i_next);
            ns0_next = PM2_G(const int, a.s_in)[((size_t)chain * K + wave) * n + i_next];
        }
        pend_slot = -1; pend_g = 0.0;
        int failed = 0, handed = 0;
        long long pos_handed = 0;
        for (long long pos = n1 - 1; pos < n; ++pos) {
            PM2_LAUNDER(ap, SweepArgs);
            PM2_FRESH_VGPR(tid); PM2_FRESH_VGPR(lane);
            const double x = xnext;
            const int ns0_cur = ns0_next;
            const int i_cur = i_next;
            if (pos + 1 < n) {
                i_next = i_next2;
                if (owner) {
                    xnext = obs_x(wave, i_next);
                    ns0_next = PM2_G(const int, a.s_in)[((size_t)chain * K + wave) * n + i_next];
                }
                if (pos + 2 < n) i_next2 = order[pos + 2];
            }
            // ---- cluster phase: the K datasets side by side, one owner wave each
            PH2(1);
            if (owner) phase_a(wave, x, ns0_cur, pos);
            PH2(2);
            PM2_BARRIER();
            PH2(3);
            if (sc()[SC_FAIL]) { failed = sc()[SC_FAIL]; break; }
            // ---- particle phase (all lanes, all datasets): allocation draw (:251-265), census, weights (:227,:245), Phi (:312-314).
            // Per dataset: the class rows of the lane's particles are read together; only particles whose row is not one-hot go
            // through the random draw (one instance of the generator, not unrolled: the step has to stay inside the instruction
            // cache); the chosen clusters are read together; particles of the lane that drew the same are counted once.
            unsigned nsp[PPL];                                                  // the labels a particle drew, a byte per dataset
#pragma unroll
            for (int u = 0; u < PPL; ++u) nsp[u] = 0;
#pragma nounroll
            for (int k = 0; k < K; ++k) {
                const DV v = view(k);
                const int ns0 = v.dsc()[DS_NS0];
                PHD(5);
                int cl_[PPL], r_[PPL], c_[PPL];
                RegArr<int, PPL> nsv;
                double inc_[PPL];
                bool draw_any = false;
#pragma unroll
                for (int u = 0; u < PPL; ++u) {
                    const int p = tid * PPL + u;
                    r_[u] = csl_get(k, u);
                    cl_[u] = (int)((colp[k * NCP + (u >> 1)] >> ((u & 1) * 16)) & 0xffffu);
                    const int hot = (int)cdf_get(v, r_[u], N + 1);
                    inc_[u] = cdf_get(v, r_[u], N);
                    const int ns = (p == 0) ? ns0 : hot;                        // reference trajectory (:262); one-hot CDF: no random number needed
                    nsv.set(u, ns);
                    draw_any |= ns < 0;
                }
                PHD(6);
                if (PM2_BALLOT(draw_any)) {
#pragma nounroll
                    for (int u = 0; u < PPL; ++u) {
                        if (nsv[u] < 0) {
                            const int p = tid * PPL + u;
                            const int rr = csl_get(k, u);
                            const double u01 = pmdi_arith::uniform01(seed, iter, (unsigned)pos, (unsigned)k, (unsigned)p, SITE_DRAW);
                            // first label whose CDF exceeds u (:252-260) = the number of leading entries that do not exceed it
                            int ns = 0, t = 0;
                            if (!CDFA || rr < v.cdfl) {
                                const double *row = v.cdf_row_l(rr);
                                for (; t + 4 <= N - 1; t += 4) {          // four LDS reads in flight
                                    const double a0 = row[t], a1 = row[t + 1], a2 = row[t + 2], a3 = row[t + 3];
                                    ns += ((a0 > u01) ? 0 : 1) + ((a1 > u01) ? 0 : 1) + ((a2 > u01) ? 0 : 1) + ((a3 > u01) ? 0 : 1);
                                }
                                for (; t < N - 1; ++t) ns += (row[t] > u01) ? 0 : 1;
                            } else {                                      // (a class slot beyond the LDS rows: its CDF from the arena)
                                auto row = PM2_G(const double, v.ar.cdfg()) + (size_t)rr * (N + 2);
                                for (; t < N - 1; ++t) ns += (row[t] > u01) ? 0 : 1;
                            }
                            nsv.set(u, ns);
                        }
                    }
                }
                PHD(7);
                unsigned packed = 0;
#pragma unroll
                for (int u = 0; u < PPL; ++u) {
                    const int ns = nsv[u];
                    lw.set(u, lw[u] + inc_[u]);                                 // logweight[p] += increment (:227,:245), dataset order
                    c_[u] = v.tab_get(cl_[u], ns);                              // sstar_id (:264)
                    nsp[u] |= (unsigned)ns << (8 * k);
                    packed |= (unsigned)ns << (8 * (u & 3));
                    if ((u & 3) == 3 || u == PPL - 1) {                         // sstar[p, i, k] (:265), four particles per store
                        u8 *ss = v.ar.sstar() + (size_t)pos * P + (size_t)tid * PPL + (u & ~3);
                        if (PPL >= 4) *PM2_G(unsigned, ss) = packed;
                        else if (PPL == 2) *PM2_G(u16, ss) = (u16)packed;
                        else *PM2_G(u8, ss) = (u8)packed;
                        packed = 0;
                    }
                }
                PHD(8);
                {
                    // particles of the lane that drew the same class, column and label are counted with the first of them; when that is all
                    // of them in every lane of the wave (the rule between resampling events), one call does it
                    int keyu[PPL];
#pragma unroll
                    for (int u = 0; u < PPL; ++u) keyu[u] = (r_[u] * P + cl_[u]) * N + nsv[u];
                    bool allsame = true;
#pragma unroll
                    for (int u = 1; u < PPL; ++u) allsame = allsame && keyu[u] == keyu[0];
                    if (PM2_BALLOT(!allsame) == 0) {
                        census(v, true, PPL, r_[0], cl_[0], nsv[0], c_[0], tid * PPL);
                    } else {
#pragma unroll
                        for (int u = 0; u < PPL; ++u) {
                            bool dup = false;
                            int mult = 0;
#pragma unroll
                            for (int u2 = 0; u2 < PPL; ++u2) {
                                const bool same = keyu[u2] == keyu[u];
                                if (u2 < u) dup |= same;
                                if (u2 >= u) mult += same ? 1 : 0;
                            }
                            census(v, !dup, mult, r_[u], cl_[u], nsv[u], c_[u], tid * PPL + u);
                        }
                    }
                }
            }
            PHD(9);
            if (K > 1) {                                                        // Phi_upweight! (src/misc.jl:50-59)
#pragma unroll
                for (int u = 0; u < PPL; ++u) {
                    int pr = 0;
                    double wv_ = lw[u];
#pragma unroll
                    for (int k1 = 0; k1 < K - 1; ++k1)
#pragma unroll
                        for (int k2 = k1 + 1; k2 < K; ++k2) {
                            wv_ += (((nsp[u] >> (8 * k1)) & 0xffu) == ((nsp[u] >> (8 * k2)) & 0xffu)) ? logphi[pr] : 0.0;
                            ++pr;
                        }
                    lw.set(u, wv_);
                }
            }
            double *red = lds<double>(L.red);
            {
                double mx = lw[0], mn = lw[0];
#pragma unroll
                for (int u = 1; u < PPL; ++u) { mx = (lw[u] > mx) ? lw[u] : mx; mn = (lw[u] < mn) ? lw[u] : mn; }
                mx = wave_max_d(mx); mn = wave_min_d(mn);
                if (lane == 0) { red[RD_MAX + wave] = mx; red[RD_MIN + wave] = mn; }
            }
            PH2(4);
            PM2_BARRIER();
            PH2(5);
            // ---- calc_ESS (src/misc.jl:15-25), first half; the bookkeeping phase of the owner waves.  If every log-weight is the
            // same number the sums are exact (P ones): ESS == P, no exps
            double mx = red[RD_MAX], mn = red[RD_MIN];
            for (int w_ = 1; w_ < NW; ++w_) { mx = (red[RD_MAX + w_] > mx) ? red[RD_MAX + w_] : mx; mn = (red[RD_MIN + w_] < mn) ? red[RD_MIN + w_] : mn; }
            PHD(10);
            const bool lw_flat = mx == mn;
            if (!lw_flat) {
                double sa = 0.0, sq = 0.0;
#pragma nounroll
                for (int u = 0; u < PPL; ++u) { const double w = exp(lw_get(u) - mx); sa += w; sq += w * w; }
                sa = wave_sum_d(sa); sq = wave_sum_d(sq);
                if (lane == 0) { red[RD_SA + wave] = sa; red[RD_SQ + wave] = sq; }
            }
            if (owner) phase_c(wave, x, pos);
            PHC(9);
            PH2(6);
            PM2_BARRIER();
            PHC(15);
            PH2(7);
            int fcode = sc()[SC_FAIL];
            if (fcode && sc()[SC_TMP2]) fcode = 1;                              // (the pool ran out in some dataset: an error whatever the others say)
            if (fcode && (fcode == 1 || !a.resume)) { failed = fcode; break; }
            // (fcode != 0 from here on: a dataset's step does not fit this kernel's tables and that dataset has changed nothing; the
            // datasets whose step did fit finish it below, then the chain is handed to the general kernel at this observation)
            // ---- statistics of the chosen clusters the owner waves left to everybody
            if (help_stats(i_cur)) PM2_BARRIER();
            // ---- every particle follows its group: new column, new class slot
#pragma nounroll
            for (int k = 0; k < K; ++k) {
                const DV v = view(k);
                if (!v.dsc()[DS_FOLLOW]) continue;                              // nothing was cloned, the classes stand as they are
                const bool cloned = v.dsc()[DS_NCLONE] != 0;
                int ck[PPL];
                col_get(k, ck);
#pragma unroll
                for (int u = 0; u < PPL; ++u) {
                    const int cl = ck[u], ns = (int)((nsp[u] >> (8 * k)) & 0xffu);
                    const u64 wm = cloned ? v.wmask_get(cl) : 0ull;
                    if ((wm >> ns) & 1ull) {
                        const int bi = v.cbi_get(cl);
                        const int inpl = (bi & 0xff) - 1, base = bi >> 8;
                        const int rnk = pm2_popc64(wm & ((1ull << ns) - 1ull));
                        ck[u] = (inpl >= 0) ? (ns == inpl ? cl : base + rnk - 1) : base + rnk;
                    }
                    csl_put(k, u, (int)lds<u8>(v.base + L.knew)[csl_get(k, u) * N + ns]);
                }
                col_put(k, ck);
            }
            if (__builtin_expect(fcode != 0, 0)) { handed = fcode; pos_handed = pos; break; }       // (the calls are made outside the loop)
            double ess = (double)P;
            if (!lw_flat) {
                double sa = 0.0, sq = 0.0;
                for (int w_ = 0; w_ < NW; ++w_) { sa += red[RD_SA + w_]; sq += red[RD_SQ + w_]; }
                ess = (sa * sa) / sq;
            }
            // The tree-ordered sums agree with calc_ESS's sequential loop to ~1e-13 relative; the decision is a comparison, so when
            // ESS lands that close to P/2 (k equal weights and the rest negligible give exactly k in the reference's order, and
            // k = P/2 does happen) the sums are redone in the reference's order by one lane.
            if (fabs(ess - 0.5 * (double)P) <= 1e-9 * (double)P) {
                double *wt = lds<double>(L.tr);
                PM2_BARRIER();
#pragma unroll
                for (int u = 0; u < PPL; ++u) wt[tid * PPL + u] = exp(lw[u] - mx);
                PM2_BARRIER();
                if (tid == 0) {
                    double na = 0.0, nb = 0.0;
                    for (int p = 0; p < P; ++p) { na += wt[p]; nb += wt[p] * wt[p]; }
                    red[120] = (na * na) / nb;
                }
                PM2_BARRIER();
                ess = red[120];
                PM2_BARRIER();
            }
            const bool res = ess <= 0.5 * (double)P;              // src/pmdi.jl:317
            if (res) {
                PH2(8);
                if (tid == 0) stat()[1] += 1;
                resample(pos, mx);
                PH2(7);
            }
            if (a.trace_on && tid == 0) {
                auto tr = PM2_G(double, a.trace) + ((size_t)chain * (n - n1 + 1) + (pos - (n1 - 1))) * (2 + 2 * K);
                tr[0] = ess; tr[1] = res ? 1.0 : 0.0;
                for (int k = 0; k < K; ++k) { tr[2 + k] = (double)view(k).dsc()[DS_MAXID]; tr[2 + K + k] = (double)view(k).dsc()[DS_NCLS]; }
            }
        }
        if (handed) {
            hand_over_cold(*this, pos_handed, handed, t_start);      // (out of line, on a copy of the lane's state)
#ifdef PM2_RESUME_GENERAL
            // ... and the general kernel's code carries the chain on, in this workgroup (what was written above is read back by other
            // lanes of it: every store is out and visible before anybody goes on)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            PM2_BARRIER();
            PM2_RESUME_GENERAL(K, NW, ap_);
#endif
            return;
        }
        if (failed) {
            if (tid == 0) {
                if (failed == 1) PM2_G(int, a.err)[chain] = -4;                            // PMDI_E_POOL
                else if (a.requeue) {
                    PM2_G(int, a.requeue)[chain] = 1;                                      // sweep again with the general kernel
                    if (!a.err_keep) PM2_G(int, a.err)[chain] = 0;                         // (its K cooperating workgroups only ever write an error)
                    if (a.handed) PM2_G(int, a.handed)[chain] = a.sweep_no;
                    if (a.requeue_total) {
                        pm2_atomic_add((u64 *)a.requeue_total + 3, (u64)1); pm2_atomic_add((u64 *)a.requeue_total + (failed - 2), (u64)1);
                        if (failed == 4 && sc()[SC_TMP1] > 2 * CLS) pm2_atomic_add((u64 *)a.requeue_total + 1, (u64)1);      // (how many of them would not fit twice the classes either)
                    }
                }
                else PM2_G(int, a.err)[chain] = PMDI_S2_REQUEUE;                           // (no requeue list: report it)
                // (the general kernel may sweep the chain with K cooperating workgroups that ADD their counters)
                for (int e = 0; e < 8; ++e) PM2_G(long long, a.stats)[(size_t)chain * 8 + e] = 0;
                PM2_G(long long, a.stats)[(size_t)chain * 8 + 7] = (a.requeue && failed != 1) ? 0 : failed;
                PM2_G(long long, a.cost)[chain] = PM2_CLOCK() - t_start;
            }
            if (failed == 1) {
                auto s_in = PM2_G(const int, a.s_in) + (size_t)chain * K * n;
                auto s_out = PM2_G(int, a.s_out) + (size_t)chain * K * n;
                for (long long e = tid; e < (long long)K * n; e += T) s_out[e] = s_in[e];
            }
            return;
        }
        PH2(9);
        finish(t_start);
        PH2(10);
        if (a.phase && tid == 0) {
            lds<long long>(L.ph)[14] = PM2_CLOCK() - t_start;
            lds<long long>(L.ph)[12] = t_wall; lds<long long>(L.ph)[13] = PM2_WALLCLOCK();      // when the chain ran (launch timeline)
            for (int e = 0; e < 16; ++e) PM2_G(long long, a.phase)[(size_t)chain * 16 + e] = lds<long long>(L.ph)[e];
        }
#undef PH2
    }

    // ---- the tables of dataset k as the arena holds them for everybody else (pmdi_export_state, the general kernel): the columns, the
    //      column and class of every particle, reference counts, cluster sizes -------------------------------------------------------
    PM2_DEV void export_dataset(int k)
    {
        const auto &a = *ap;
        const DV v = view(k);
        const int ncol = v.dsc()[DS_NCOL], maxid = v.dsc()[DS_MAXID];
        auto tg = PM2_G(int, v.ar.tabg());
        for (int e = tid; e < ncol * N && e < v.cols_l * N; e += T) tg[e] = (int)lds<u16>(v.base + L.tab)[e];
        auto cg = PM2_G(int, v.ar.colg());
        auto pg = PM2_G(int, v.ar.pidg());
        int ck[PPL];
        col_get(k, ck);
#pragma unroll
        for (int u = 0; u < PPL; ++u) { cg[tid * PPL + u] = ck[u]; pg[tid * PPL + u] = lds<int>(v.base + L.clsval)[csl_get(k, u)]; }
        auto cng = PM2_G(int, v.ar.cn());
        auto ctg = PM2_G(int, v.ar.counts());
        for (int id = tid; id < v.idcap && id <= a.cap; id += T) { cng[id] = (id <= maxid) ? lds<int>(v.base + L.cn)[id] : cng[id]; ctg[id] = (id <= maxid) ? lds<int>(v.base + L.counts)[id] : 0; }
        for (int id = maxid + 1 + tid; id <= a.cap; id += T) ctg[id] = 0;
        if (tid == 0) {
            PM2_G(int, a.kstate)[((size_t)chain * PMDI_KMAX_I + k) * 2] = maxid;
            PM2_G(int, a.kstate)[((size_t)chain * PMDI_KMAX_I + k) * 2 + 1] = 0;
        }
    }

    // ---- a step of observation `pos` does not fit this kernel's tables in some dataset (more than CLS particle classes, or cluster ids
    //      beyond 16 bits): the chain goes to the general kernel HERE, not back to the start of the sweep.  The datasets that halted
    //      have changed nothing (phase_c checks before it commits); the others have finished the step.  Everything the general kernel
    //      keeps between steps is written where it looks for it: tables and per-particle columns / classes (export_dataset), the
    //      step scratch of the per-id tables back to idle, log-weights (they hold this observation's increments and the Phi term: the
    //      draws of ALL datasets are done and recorded in the history), counters, and the resume record: position, which datasets are
    //      done, live columns per dataset.  pmdi_sweep.hip replays the bookkeeping of the halted datasets from the recorded draws and
    //      carries on with calc_ESS of this observation.
    PM2_DEV void hand_over(long long pos, int fcode, long long t_start)
    {
        const auto &a = *ap;
        PM2_BARRIER();
        int done_mask = 0;
#pragma nounroll
        for (int k = 0; k < K; ++k) {
            const DV v = view(k);
            int *dsc = v.dsc();
            const bool halted = dsc[DS_HALT] != 0;
            if (!halted) done_mask |= 1 << k;
            export_dataset(k);
            auto g1 = PM2_G(int, v.ar.ncop());
            auto g2 = PM2_G(int, v.ar.firstp());
            for (int e = tid; e < v.idcap && e <= a.cap; e += T) { g1[e] = 0; g2[e] = 0x7fffffff; }
            if (halted) {                 // what its census marked beyond the LDS tables
                const int ndx = dsc[DS_NDX];
                for (int e = tid; e < ndx; e += T) { const int id = PM2_G(const int, v.ar.dl())[e]; g1[id] = 0; g2[id] = 0x7fffffff; }
            }
            if (tid == 0) {
                PM2_G(int, a.resume)[(size_t)chain * 16 + 2 + k] = dsc[DS_NCOL];
                if (halted) wk(k)[WK_EVAL] += dsc[DS_NNEED];          // (its clusters were evaluated; the replay will not do that again)
            }
        }
#pragma unroll
        for (int u = 0; u < PPL; ++u) PM2_G(double, a.uscratch)[(size_t)chain * K * P + tid * PPL + u] = lw[u];
        PM2_BARRIER();
        if (tid == 0) {
            auto st = PM2_G(long long, a.stats) + (size_t)chain * 8;
            const long long *s = stat();
            st[ST_NOPS] = s[0]; st[ST_NRESAMPLE] = s[1]; st[ST_NCLONES] = s[2]; st[ST_MAXID] = s[3]; st[ST_SUMCLASSES] = s[4];
            st[5] = s[5]; st[6] = 0; st[7] = 0;
            PM2_G(int, a.resume)[(size_t)chain * 16] = (int)pos;
            PM2_G(int, a.resume)[(size_t)chain * 16 + 1] = done_mask;
            PM2_G(int, a.requeue)[chain] = 0;             // (not for another launch: this workgroup carries the chain on)
            if (!a.err_keep) PM2_G(int, a.err)[chain] = 0;
            if (a.handed) PM2_G(int, a.handed)[chain] = a.sweep_no;
            if (a.requeue_total) {
                pm2_atomic_add((u64 *)a.requeue_total + 3, (u64)1); pm2_atomic_add((u64 *)a.requeue_total + (fcode - 2), (u64)1);
                if (fcode == 4 && sc()[SC_TMP1] > 2 * CLS) pm2_atomic_add((u64 *)a.requeue_total + 1, (u64)1);
            }
            PM2_G(long long, a.cost)[chain] = PM2_CLOCK() - t_start;
        }
        if (a.work && tid < K * 8) PM2_G(long long, a.work)[((size_t)chain * PMDI_KMAX_I) * 8 + tid] = lds<long long>(L.wk)[tid];
    }

    // ---- particle pick (src/pmdi.jl:345-350), s = sstar[p_star, :, :] (:373), counters, the state the debug export reads --------------
    PM2_DEV void finish(long long t_start)
    {
        const auto &a = *ap;
        double *red = lds<double>(L.red);
        double *wt = lds<double>(L.tr);
        auto order = PM2_G(const int, a.order) + (size_t)chain * n;
        PM2_BARRIER();
        {
            double mx = lw[0];
#pragma unroll
            for (int u = 1; u < PPL; ++u) mx = (lw[u] > mx) ? lw[u] : mx;
            mx = wave_max_d(mx);
            if (lane == 0) red[wave] = mx;
        }
        PM2_BARRIER();
        double mx = red[0];
        for (int w_ = 1; w_ < T / 64; ++w_) mx = (red[w_] > mx) ? red[w_] : mx;
#pragma unroll
        for (int u = 0; u < PPL; ++u) wt[tid * PPL + u] = exp(lw[u] - mx);
        PM2_BARRIER();
        if (tid == 0) {   // StatsBase.sample(::Weights): sequential sum and scan, as the oracle
            double sum = 0.0;
            for (int p = 0; p < P; ++p) sum += wt[p];
            const double t = pmdi_arith::uniform01(seed, iter, 0, 0, 0, SITE_PSTAR) * sum;
            int ip = 0;
            double cw = wt[0];
            while (cw < t && ip < P - 1) { ++ip; cw += wt[ip]; }
            sc()[SC_PSTAR] = ip;
        }
        PM2_BARRIER();
        const int pstar = sc()[SC_PSTAR];
        for (long long pp = tid; pp < n; pp += T) {
            const int i = order[pp];
            for (int k = 0; k < K; ++k) {
                int val;
                if (pp < n1 - 1) val = PM2_G(const int, a.s_in)[((size_t)chain * K + k) * n + i];          // sstar[:, i, k] .= s[i, k] (:204)
                else val = PM2_G(const u8, view(k).ar.sstar())[(size_t)pp * P + pstar];
                PM2_G(int, a.s_out)[((size_t)chain * K + k) * n + i] = val;
            }
        }
        if (a.lw_out)
#pragma unroll
            for (int u = 0; u < PPL; ++u) PM2_G(double, a.lw_out)[(size_t)chain * P + tid * PPL + u] = lw[u];
        // what pmdi_export_state reads: the columns, the column and class of every particle, counts, cluster sizes
#pragma nounroll
        for (int k = 0; k < K; ++k) export_dataset(k);
        if (tid == 0) {
            auto st = PM2_G(long long, a.stats) + (size_t)chain * 8;
            const long long *s = stat();
            PM2_G(int, a.pstar)[chain] = pstar;
            st[ST_NOPS] = s[0]; st[ST_NRESAMPLE] = s[1]; st[ST_NCLONES] = s[2]; st[ST_MAXID] = s[3]; st[ST_SUMCLASSES] = s[4];
            st[5] = s[5]; st[6] = 0; st[7] = 0;
            if (!a.err_keep) PM2_G(int, a.err)[chain] = 0;
            if (a.requeue) PM2_G(int, a.requeue)[chain] = 0;
            if (a.swept_by) PM2_G(int, a.swept_by)[chain] = 1;
            PM2_G(long long, a.cost)[chain] = PM2_CLOCK() - t_start;
        }
        if (a.work && tid < K * 8) PM2_G(long long, a.work)[((size_t)chain * PMDI_KMAX_I) * 8 + tid] = lds<long long>(L.wk)[tid];
    }
#undef CLS
#undef L
};

template <int K, int PPL, int NW, bool GO> PM2_COLD void hand_over_cold(Sweep2<K, PPL, NW, GO> s, long long pos, int fcode, long long t_start)
{
    s.hand_over(pos, fcode, t_start);
}

}  // namespace pmdi_s2
