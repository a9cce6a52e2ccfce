// pmdi_device.h -- device helpers shared by the sweep kernel (pmdi_sweep.hip) and the
// unit / feature-selection kernels (pmdi_kernels.hip).  Compile with -ffp-contract=off.
// Reference lines are cited as file:line relative to /root/reference.
#pragma once
#include "pmdi_internal.h"

namespace pmdi_dev {

// ---------------------------------------------------------------------------
// Counter-based RNG (specification shared with oracle/pmdi_oracle.c):
// Philox4x32-10, key = (seed lo, seed hi), ctr = (p, pos, site<<16|k, iter).
__device__ __forceinline__ double uniform01(unsigned long long seed, unsigned iter, unsigned pos,
                                            unsigned k, unsigned p, unsigned site)
{
    unsigned c0 = p, c1 = pos, c2 = (site << 16) | k, c3 = iter;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // 52 random bits -> odd multiple of 2^-53: uniform on the open interval (0,1), never 0
    unsigned long long m = ((unsigned long long)(c0 >> 6) << 26) | (unsigned long long)(c1 >> 6);
    return (double)(2 * m + 1) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------------------
// Arena pointers carry the global address space: the compiler then emits global_load/global_store
// (vmcnt only) instead of flat instructions, which also tick the LDS counter and make every LDS
// read behind them wait for the HBM round trip.
#define PMDI_GLOBAL __attribute__((address_space(1)))
typedef PMDI_GLOBAL int *gint;
typedef const PMDI_GLOBAL int *gcint;
typedef PMDI_GLOBAL double *gdbl;
typedef const PMDI_GLOBAL double *gcdbl;
typedef double dbl2v __attribute__((ext_vector_type(2)));   // plain 16-byte pair: loads and stores work in any address space
typedef PMDI_GLOBAL dbl2v *gdbl2;
typedef PMDI_GLOBAL long long *gi64;
typedef PMDI_GLOBAL unsigned char *gu8;
typedef const PMDI_GLOBAL unsigned char *gcu8;
template <class Tp> __device__ __forceinline__ PMDI_GLOBAL Tp *glob(Tp *p) { return (PMDI_GLOBAL Tp *)p; }
template <class Tp> __device__ __forceinline__ Tp *gen(PMDI_GLOBAL Tp *p) { return (Tp *)p; }
__device__ __forceinline__ double2 ld2(const PMDI_GLOBAL dbl2v *p, size_t i) { const dbl2v v = p[i]; return make_double2(v.x, v.y); }
__device__ __forceinline__ void st2(PMDI_GLOBAL dbl2v *p, size_t i, double2 v) { dbl2v w; w.x = v.x; w.y = v.y; p[i] = w; }

#if defined(PMDI_SWEEP_TU)
// The 22 array pointers of a (chain, dataset) are wave-uniform, but the step loop cannot keep them in SGPRs: computed at the top
// of every step they were parked in VGPRs and spilled -- 22 scratch stores per lane and step in the 256-register build, about half
// of the spill write-back per chain and step, and a scratch reload in front of every phase (profiles/README.md, round 2).  Here a
// field is a (dataset, chain) pair and the address is rebuilt where it is used, from scalar loads of the argument block (the asm
// keeps the loads from being hoisted back out of the step loop).  Round 3 A/B on the GPU, same box back to back: HL 459.9 ->
// 495.6 iterations/s, parity suite equal; kernel body 36 -> 5 scratch stores.
typedef const __attribute__((address_space(4))) DsetDev *cdsptr;   // the argument block through the constant address space: s_load
__device__ __forceinline__ cdsptr opaque_ds(const DsetDev *d)
{
    // wave-uniform by construction (argument block + blockIdx-derived dataset index); out-of-line device functions receive
    // their arguments in VGPRs, so say so explicitly
    const unsigned long long v = (unsigned long long)d;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (cdsptr)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
template <class Tp, size_t DsetDev::*OFF>
struct LazyArr {
    const DsetDev *d;
    int chain;
    __device__ __forceinline__ PMDI_GLOBAL Tp *p() const
    {
        const cdsptr dd = opaque_ds(d);
        return glob((Tp *)(dd->arena + (size_t)chain * dd->stride + dd->*OFF));
    }
    __device__ __forceinline__ operator PMDI_GLOBAL Tp *() const { return p(); }
    template <class I> __device__ __forceinline__ PMDI_GLOBAL Tp &operator[](I i) const { return p()[i]; }
    template <class I> __device__ __forceinline__ PMDI_GLOBAL Tp *operator+(I i) const { return p() + i; }
};
struct LazyPart {
    const DsetDev *d;
    int chain;
    __device__ __forceinline__ gint operator[](int cur) const
    {
        const cdsptr dd = opaque_ds(d);
        return glob((int *)(dd->arena + (size_t)chain * dd->stride + dd->o_particle[cur]));
    }
};
template <class Tp, size_t DsetDev::*OFF> __device__ __forceinline__ PMDI_GLOBAL Tp *raw(const LazyArr<Tp, OFF> &x) { return x.p(); }
struct KS {
    LazyPart part;
    LazyArr<int, &DsetDev::o_pid> pid;
    LazyArr<int, &DsetDev::o_sid> sid;
    LazyArr<int, &DsetDev::o_kv> kv;
    LazyArr<int, &DsetDev::o_newid> newid;
    LazyArr<int, &DsetDev::o_counts> counts;
    LazyArr<int, &DsetDev::o_ncop> ncop;
    LazyArr<int, &DsetDev::o_firstc> firstc;
    LazyArr<int, &DsetDev::o_cn> cn;
    LazyArr<int, &DsetDev::o_clslead> clslead;
    LazyArr<int, &DsetDev::o_clsval> clsval;
    LazyArr<int, &DsetDev::o_dl> dl;
    LazyArr<int, &DsetDev::o_col> col;
    LazyArr<unsigned long long, &DsetDev::o_cgrp> cgrp;
    LazyArr<double, &DsetDev::o_lp> lp;
    LazyArr<double, &DsetDev::o_cdf> cdf;
    LazyArr<dbl2v, &DsetDev::o_ml> ml;
    LazyArr<dbl2v, &DsetDev::o_sb> sb;
    LazyArr<int, &DsetDev::o_cnt> cnt;
    LazyArr<long long, &DsetDev::o_nbs> nbs;
    LazyArr<unsigned char, &DsetDev::o_sstar> sstar;
};
__device__ __forceinline__ KS make_ks(const DsetDev &d, int chain)
{
    const DsetDev *dp = &d;
    return KS{{dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain},
              {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain}, {dp, chain},
              {dp, chain}};
}
#else
struct KS {  // pointers of one (chain, dataset)
    gint part[2];
    gint pid, sid, kv, newid, counts, ncop, firstc, cn, clslead, clsval, dl, col;
    PMDI_GLOBAL unsigned long long *cgrp;
    gdbl lp, cdf;
    gdbl2 ml, sb;
    gint cnt;
    gi64 nbs;
    gu8 sstar;
};

__device__ __forceinline__ KS make_ks(const DsetDev &d, int chain)
{
    KS s;
    char *b = d.arena + (size_t)chain * d.stride;
    s.part[0] = glob((int *)(b + d.o_particle[0]));
    s.part[1] = glob((int *)(b + d.o_particle[1]));
    s.col = glob((int *)(b + d.o_col));
    s.cgrp = glob((unsigned long long *)(b + d.o_cgrp));
    s.pid = glob((int *)(b + d.o_pid));
    s.sid = glob((int *)(b + d.o_sid));
    s.kv = glob((int *)(b + d.o_kv));
    s.newid = glob((int *)(b + d.o_newid));
    s.counts = glob((int *)(b + d.o_counts));
    s.ncop = glob((int *)(b + d.o_ncop));
    s.firstc = glob((int *)(b + d.o_firstc));
    s.lp = glob((double *)(b + d.o_lp));
    s.cn = glob((int *)(b + d.o_cn));
    s.ml = glob((dbl2v *)(b + d.o_ml));
    s.sb = glob((dbl2v *)(b + d.o_sb));
    s.cnt = glob((int *)(b + d.o_cnt));
    s.nbs = glob((long long *)(b + d.o_nbs));
    s.sstar = glob((unsigned char *)(b + d.o_sstar));
    s.clslead = glob((int *)(b + d.o_clslead));
    s.clsval = glob((int *)(b + d.o_clsval));
    s.cdf = glob((double *)(b + d.o_cdf));
    s.dl = glob((int *)(b + d.o_dl));
    return s;
}

#endif
template <class Tp> __device__ __forceinline__ Tp raw(Tp x) { return x; }   // (a KS field as a plain pointer: identity in the default build)

// ---------------------------------------------------------------------------
// Block-level primitives (64-wide waves).
template <int T>
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v,
                                                              unsigned long long &total,
                                                              unsigned long long *scr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) scr[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
        unsigned long long s = scr[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// Exclusive ranks of up to three 1-bit flags over the block in thread order, by wave ballots:
// returns packed (f0 | f1<<20 | f2<<40) exclusive prefix and block total, like block_excl_scan.
template <int T>
__device__ __forceinline__ unsigned long long block_flag_scan(bool f0, bool f1, bool f2,
                                                              unsigned long long &total,
                                                              unsigned long long *scr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1), b2 = __ballot(f2);
    const unsigned long long mine = (unsigned long long)__popcll(b0 & lt) |
                                    ((unsigned long long)__popcll(b1 & lt) << 20) |
                                    ((unsigned long long)__popcll(b2 & lt) << 40);
    if (lane == 0)
        scr[wave] = (unsigned long long)__popcll(b0) | ((unsigned long long)__popcll(b1) << 20) |
                    ((unsigned long long)__popcll(b2) << 40);
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) {
        const unsigned long long v = scr[w];
        if (w < wave) base += v;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return base + mine;
}

// Wave-level all-reduce of a double on the DPP network (no LDS round trips): xor-1 and xor-2 inside
// quads, half-row and row mirrors give every lane its 16-lane row total; the four row totals are
// read with v_readlane and combined as (r0 op r1) op (r2 op r3).  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max_f64(double v)
{
    double t;
    t = dpp_f64<0xB1>(v); v = (t > v) ? t : v;
    t = dpp_f64<0x4E>(v); v = (t > v) ? t : v;
    t = dpp_f64<0x141>(v); v = (t > v) ? t : v;
    t = dpp_f64<0x140>(v); v = (t > v) ? t : v;
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    const double m01 = (r1 > r0) ? r1 : r0, m23 = (r3 > r2) ? r3 : r2;
    return (m23 > m01) ? m23 : m01;
}

// Block-wide max / pair of sums.  scr: 48 doubles; the max uses [32,48), the sums [0,32), so the two
// can follow each other with one barrier each (the caller's next use of scr is barriers away).
template <int T>
__device__ __forceinline__ double block_max(double v, double *scr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max_f64(v);
    if (lane == 0) scr[32 + wave] = v;
    __syncthreads();
    double m = scr[32];
#pragma unroll
    for (int w = 1; w < T / 64; ++w) { double t = scr[32 + w]; m = (t > m) ? t : m; }
    return m;
}

template <int T>
__device__ __forceinline__ void block_sum2(double &a, double &b, double *scr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    if (lane == 0) { scr[wave] = a; scr[16 + wave] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) { sa += scr[w]; sb += scr[16 + w]; }
    a = sa; b = sb;
}

// Lanes of a wave holding equal `key` (among `valid` lanes) elect the lowest
// lane as group leader; returns true on the leader with the group's size.
__device__ __forceinline__ bool wave_group(int key, bool valid, int &count)
{
    const int lane = threadIdx.x & 63;
    unsigned long long active = __ballot(valid);
    bool leader = false;
    count = 0;
    while (active) {
        int l0 = __ffsll((long long)active) - 1;
        int k0 = __builtin_amdgcn_readlane(key, l0);   // l0 is wave-uniform: v_readlane, not a ds_bpermute round trip
        unsigned long long m = __ballot(valid && key == k0);
        if (lane == l0) { leader = true; count = __popcll(m); }
        active &= ~m;
    }
    return leader;
}

// As wave_group, with at most `rounds` elections: lanes whose key has not come up by then speak for themselves (leader,
// count 1).  For callers whose leader action is idempotent or additive (a flag, a counter, a compare-and-swap that one lane
// wins): when a wave holds dozens of distinct keys the election loop costs more than the atomics it saves.
__device__ __forceinline__ bool wave_group_capped(int key, bool valid, int &count, int rounds)
{
    const int lane = threadIdx.x & 63;
    unsigned long long active = __ballot(valid);
    bool leader = false;
    count = 0;
    for (int r = 0; r < rounds && active; ++r) {
        const int l0 = __ffsll((long long)active) - 1;
        const int k0 = __builtin_amdgcn_readlane(key, l0);
        const unsigned long long m = __ballot(valid && key == k0);
        if (lane == l0) { leader = true; count = __popcll(m); }
        active &= ~m;
    }
    if ((active >> lane) & 1ull) { leader = true; count = 1; }
    return leader;
}

// As wave_group, but every lane learns the lane id of its group's leader (lowest lane of the
// group); `count` is set on leaders.
__device__ __forceinline__ int wave_group_lead(int key, bool valid, int &count)
{
    const int lane = threadIdx.x & 63;
    unsigned long long active = __ballot(valid);
    int lead = lane;
    count = 0;
    while (active) {
        const int l0 = __ffsll((long long)active) - 1;
        const int k0 = __builtin_amdgcn_readlane(key, l0);
        const bool mine = valid && key == k0;
        const unsigned long long m = __ballot(mine);
        if (mine) lead = l0;
        if (lane == l0) count = __popcll(m);
        active &= ~m;
    }
    return lead;
}

// number of set bits strictly below bit p of a bitmap (32-bit words, an even number of them
// allocated, 8-byte aligned); read as 64-bit words, four loads in flight
__device__ __forceinline__ int popc_below(const unsigned *bm, int p)
{
    const unsigned long long *b64 = (const unsigned long long *)bm;
    const int w = p >> 6;
    int n = 0, i = 0;
    for (; i + 4 <= w; i += 4) {
        const unsigned long long x0 = b64[i], x1 = b64[i + 1], x2 = b64[i + 2], x3 = b64[i + 3];
        n += __popcll(x0) + __popcll(x1) + __popcll(x2) + __popcll(x3);
    }
    for (; i < w; ++i) n += __popcll(b64[i]);
    return n + __popcll(b64[w] & ((1ull << (p & 63)) - 1ull));
}

// Julia Base.accumulate_pairwise! (base/accumulate.jl), the algorithm behind
// cumsum(::Vector{Float64}) at src/misc.jl:29: in place on c[0..n).  Run by
// one lane; the recursion is unrolled onto a small explicit stack.
__device__ void jl_cumsum_inplace(double *c, int n)
{
    if (n <= 1) return;
    const double v1 = c[0];
    int f_i1[24], f_n[24], f_stage[24];
    double f_s[24], f_left[24];
    int sp = 0;
    f_i1[0] = 1; f_n[0] = n - 1; f_s[0] = v1; f_stage[0] = 0; f_left[0] = 0.0;
    double ret = 0.0;
    while (sp >= 0) {
        const int i1 = f_i1[sp], nn = f_n[sp];
        const double s = f_s[sp];
        if (nn < 128) {
            double s_ = c[i1];
            c[i1] = s + s_;
            for (int i = i1 + 1; i < i1 + nn; ++i) {
                s_ = s_ + c[i];
                c[i] = s + s_;
            }
            ret = s_;
            --sp;
        } else if (f_stage[sp] == 0) {
            f_stage[sp] = 1;
            ++sp;
            f_i1[sp] = i1; f_n[sp] = nn >> 1; f_s[sp] = s; f_stage[sp] = 0;
        } else if (f_stage[sp] == 1) {
            f_left[sp] = ret;
            f_stage[sp] = 2;
            const int n2 = nn >> 1;
            ++sp;
            f_i1[sp] = i1 + n2; f_n[sp] = nn - n2; f_s[sp] = s + ret; f_stage[sp] = 0;
        } else {
            ret = f_left[sp] + ret;
            --sp;
        }
    }
}

// ---------------------------------------------------------------------------
// Per-type arithmetic, restating the reference expression by expression.

// cluster_add!(::GaussianCluster): gaussian_cluster.jl:54-66 (one feature)
__device__ __forceinline__ void gauss_add(double x, int nnew, double2 &ml, double2 &sb)
{
    const double n = (double)nnew;
    sb.x = sb.x + x;
    const double d = x - ml.x;
    sb.y = sb.y + ((double)(nnew - 1) + 0.001) * (d * d) / (2.0 * (n + 0.001));
    ml.x = sb.x / (n + 0.001);
    ml.y = ((0.5 * n + 0.5) * (n + 0.001)) / (sb.y * (n + 1.001));
}

// The sweep's pool stores only (Sigma, beta) per feature: mu and lambda are pure functions of
// (Sigma, beta, n) -- gaussian_cluster.jl:60-63 recomputes them from exactly these values at
// every add -- so they are derived on demand, bit-identically, and an update touches 16 bytes
// instead of 64.  (Valid while the feature flags are fixed for the lifetime of the pool, which
// holds inside a sweep: a feature is either updated at every add or never read.)
__device__ __forceinline__ void gauss_add_sb(double x, int nnew, double2 &sb)
{
    const double n = (double)nnew;
    const double mu_prev = (nnew == 1) ? 0.0 : sb.x / ((double)(nnew - 1) + 0.001);
    sb.x = sb.x + x;
    const double d = x - mu_prev;
    sb.y = sb.y + ((double)(nnew - 1) + 0.001) * (d * d) / (2.0 * (n + 0.001));
}

__device__ __forceinline__ double2 gauss_ml(int cn, double2 sb)
{
    if (cn == 0) return make_double2(0.0, 1.0);          // GaussianCluster(dataFile): mu = 0, lambda = 1
    const double n = (double)cn;
    return make_double2(sb.x / (n + 0.001), ((0.5 * n + 0.5) * (n + 0.001)) / (sb.y * (n + 1.001)));
}

// the two per-feature terms of calc_logprob(::GaussianCluster): gaussian_cluster.jl:45-48
__device__ __forceinline__ void gauss_terms(double x, double n, double2 ml, double &ta, double &tb)
{
    ta = 0.5 * log(ml.y / (n + 1.0));
    const double d = x - ml.x;
    tb = (0.5 * n + 1.0) * log(1.0 + (1.0 / (n + 1.0)) * (d * d) * ml.y);
}

// deepcopy (src/pmdi.jl:297) + cluster_add! (:300) of one feature of one chosen cluster:
// statistics of pool id `src` plus the observation go to pool id `dst` (dst == src: in place).
__device__ __forceinline__ void stats_update_one(const DsetDev &d, const KS &s, bool on, const double *xs,
                                                 int src, int dst, int nnew, int D, int q)
{
    if (d.kind == K_GAUSSIAN) {
        double2 sb = ld2(s.sb, (size_t)src * D + q);
        if (on) gauss_add_sb(xs[q], nnew, sb);
        if (on || dst != src) st2(s.sb, (size_t)dst * D + q, sb);
    } else if (d.kind == K_CATEGORICAL) {
        const int x = ((const int *)xs)[q];
        const gcint cs = s.cnt + ((size_t)src * D + q) * d.L;
        const gint cd = s.cnt + ((size_t)dst * D + q) * d.L;
        if (dst != src) for (int l = 0; l < d.L; ++l) cd[l] = cs[l];
        if (on) cd[x - 1] = cs[x - 1] + 1;
    } else {
        const int x = ((const int *)xs)[q];
        s.nbs[(size_t)dst * D + q] = s.nbs[(size_t)src * D + q] + (on ? x : 0);
    }
}

// calc_logprob(::NegBinomCluster) per-feature term: negbinom_cluster.jl:33-37;
// loggamma of integers comes from the host-built table LG[m] = lgamma(m)
__device__ __forceinline__ double negbin_term(gcdbl lg, long long n, long long x, long long S)
{
    return lg[1 + n + 1] + lg[1 + x + S] + lg[1 + n + 1 + S] - lg[1 + n + 1 + 1 + x + S] -
           lg[1 + n] - lg[1 + S];
}
}  // namespace pmdi_dev
