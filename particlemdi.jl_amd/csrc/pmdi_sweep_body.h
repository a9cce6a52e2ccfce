// pmdi_sweep_body.h -- the device code of the general sweep kernel (see pmdi_sweep.hip for the account of the mapping): every
// function of it and the kernel's body as a device function template, so that two translation units can carry it -- pmdi_sweep.hip
// (the __global__ kernels of the general sweep) and pmdi_sweep2.hip (the settled-chain kernel, whose workgroup carries a chain on
// with this code, in place, from the observation where its own tables no longer hold the chain: RESUME = true).
// Compile with -ffp-contract=off.  Reference lines are cited as file:line relative to /root/reference.
#pragma once
#define PMDI_SWEEP_TU 1     // (this file gets the KS whose array addresses are rebuilt from the argument block where they are used: pmdi_device.h)
#include "pmdi_device.h"

using namespace pmdi_dev;

namespace {

// LDS arrays are held as 32-bit address-space-3 pointers: half the registers of generic
// pointers and guaranteed ds_* instructions.
typedef __attribute__((address_space(3))) int *lint;
typedef __attribute__((address_space(3))) unsigned *lu32;
typedef __attribute__((address_space(3))) double *ldbl;
typedef __attribute__((address_space(3))) unsigned long long *lu64;
typedef __attribute__((address_space(3))) long long *li64;
typedef __attribute__((address_space(3))) unsigned char *lu8;

// explicit LDS -> generic pointer conversion (after inlining the compiler still sees the
// address space and emits ds_* instructions)
template <class Tp>
__device__ __forceinline__ Tp *gen(__attribute__((address_space(3))) Tp *p) { return (Tp *)p; }
using pmdi_dev::gen;   // the overload for arena (global) pointers

// ---------------------------------------------------------------------------
// Open-addressing hash table in LDS: key = cluster id (0 = empty), two payload words.
struct HT {
    lint key, a, b;
    unsigned mask;
};

__device__ __forceinline__ unsigned ht_hash(int id) { return ((unsigned)id * 2654435761u) >> 12; }

// slot of `id`, creating it if absent (won = this lane created it); -1 if too crowded
__device__ __forceinline__ int ht_insert(const HT &h, int id, bool &won, int maxprobe)
{
    unsigned s = ht_hash(id) & h.mask;
    won = false;
    for (int pr = 0; pr < maxprobe; ++pr) {
        const int k = __hip_atomic_load(gen(&h.key[s]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == id) return (int)s;
        if (k == 0) {
            const int old = atomicCAS(gen(&h.key[s]), 0, id);
            if (old == 0) { won = true; return (int)s; }
            if (old == id) return (int)s;
        }
        s = (s + 1) & h.mask;
    }
    return -1;
}

__device__ __forceinline__ int ht_find(const HT &h, int id)
{
    unsigned s = ht_hash(id) & h.mask;
    for (unsigned pr = 0; pr <= h.mask; ++pr) {
        if (h.key[s] == id) return (int)s;
        s = (s + 1) & h.mask;
    }
    return 0;
}

// Barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global
// load and store of the wave (s_waitcnt vmcnt(0)); inside a fast-path step, global data written
// by one lane is read by other lanes in the NEXT step at the earliest, so only the step's last
// barrier needs that -- the others let global loads (e.g. the prefetched observation row) and
// stores stay in flight.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Inlining policy of the cold paths (A/B-tested: see DESIGN.md section 6)
// waves per SIMD the 256-thread build is register-capped for.  2 = the full 256 VGPRs (no spills).
// Measured: 3 (168 VGPRs, 73 spill slots) makes a converged chain 18% slower (156 vs 132 ms), which
// a third chain per CU would not win back after the LDS tables had been shrunk to fit.
#ifndef PMDI_LIGHT_WPS
#define PMDI_LIGHT_WPS 2
#endif
// pool reads in flight per lane in the statistics update of the 256-register wide build
#ifndef PMDI_VH_U
#define PMDI_VH_U 8
#endif
#ifndef PMDI_COLD_PREFIX
#define PMDI_COLD_PREFIX __noinline__
#endif
#ifndef PMDI_COLD_SLOW
#define PMDI_COLD_SLOW __forceinline__   // out-of-line cost 12% on converged chains (call-site spills in the step loop)
#endif
#ifndef PMDI_COLD_RESAMPLE
#define PMDI_COLD_RESAMPLE __noinline__
#endif
#ifndef PMDI_COLD_FINAL
#define PMDI_COLD_FINAL __noinline__
#endif

// ---------------------------------------------------------------------------
struct Carve {  // byte offsets of the LDS arrays (shared by host sizing and the kernel)
    size_t xs, pis, lw, term, lpl, cdf, scan, red, pid, sid, kv, lead_of, slot_of, cl_lead, cl_val,
        need, need_slot, item_id, dl, dl_slot, h1k, h1a, h2k, h2a, h2b, ktab_minp, ktab_val, klist, kl_v, wk,
        kl_key, fl_p, fl_slot, fl_nnew, fl_tgt, bm_fresh, bm_clone, bm_keep, bm_reuse, col, kncol, leaf_i1, leaf_n, leaf_tot, leaf_carry, leaf_prog, kmaxid, kncls, kcur, knflag, khint, misc, ph, stat,
        fl, news, total;
};

__host__ __device__ inline void carve_lds(const SweepArgs &a, Carve &c)
{
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
    const int Dp = (a.Dmax + 15) & ~15;
    // ---- fixed-size tables: compile-time offsets ----
    c.scan = take(16 * 8);
    c.red = take(48 * 8);
    c.misc = take(16 * 4);
    c.ph = take(16 * 8);
#ifdef PMDI_RESAMPLE_TIMERS
    c.stat = take(24 * 8);
#else
    c.stat = take(8 * 8);
#endif
    c.wk = take(PMDI_KMAX_I * 8 * 8);
    c.kmaxid = take(PMDI_KMAX_I * 4);
    c.kncls = take(PMDI_KMAX_I * 4);
    c.kcur = take(PMDI_KMAX_I * 4);
    c.knflag = take(PMDI_KMAX_I * 4);
    c.khint = take(PMDI_KMAX_I * 4);
    c.kncol = take(PMDI_KMAX_I * 4);
    c.leaf_i1 = take(64 * 4);
    c.leaf_n = take(64 * 4);
    c.leaf_tot = take(64 * 8);
    c.leaf_carry = take(64 * 8);
    c.leaf_prog = take(256);
    c.dl_slot = take((size_t)PMDI_DL_LDS * 4);
    c.h1k = take((size_t)PMDI_HT_SIZE * 4);
    c.h1a = take((size_t)PMDI_HT_SIZE * 4);
    c.h2k = take((size_t)PMDI_HT_SIZE * 4);
    c.h2a = take((size_t)PMDI_HT_SIZE * 4);
    c.h2b = take((size_t)PMDI_HT_SIZE * 4);
    c.fl_p = take((size_t)PMDI_HT_SIZE * 4);
    c.fl_slot = take((size_t)PMDI_HT_SIZE * 4);
    c.fl_nnew = take((size_t)PMDI_HT_SIZE * 4);
    c.fl_tgt = take((size_t)PMDI_HT_SIZE * 4);
    // ---- tables of (class, label) items: a.item_cap entries each (256, or 384 when N > 32 so that more than
    // five classes stay on the fast path); need .. dl are contiguous (the fallback step's cluster list) ----
    const size_t icap = (size_t)a.item_cap;
    const int KL = a.ksplit ? 1 : a.K;                       // datasets this workgroup sweeps
    c.lpl = take(icap * 8);
    c.cdf = take((2 * icap + 4) * 8);                        // rows of N + 2: CDF, log-increment, one-hot label
    c.need = take(icap * 4);
    c.need_slot = take(icap * 4);
    c.item_id = take(icap * 4);
    c.ktab_minp = take(icap * 4);
    c.ktab_val = take(icap * 4);
    c.klist = take(icap * 4);
    c.kl_v = take(icap * 4);
    c.kl_key = take(icap * 4);
    c.dl = take((size_t)3 * PMDI_DL_LDS * 4);
    // ---- sizes that depend on the configuration ----
    c.xs = take((size_t)a.Dmax * 8);
    c.pis = take((size_t)KL * a.N * 8);
    c.term = take((size_t)a.terms_cap * 8);
    c.lw = take((size_t)a.P * 8);
    c.pid = take(a.pid_lds ? (size_t)KL * a.P * 4 : 0);
    c.col = take(a.col_lds ? (size_t)KL * a.P * 4 : 0);
    c.sid = take(a.pp_lds ? (size_t)a.P * 4 : 0);
    c.kv = take(a.pp_lds ? (size_t)a.P * 4 : 0);
    c.lead_of = take((size_t)(a.P + 1) * 4);
    c.slot_of = take((size_t)(a.P + 1) * 4);
    c.cl_lead = take((size_t)KL * PMDI_CLS_LDS * 4);
    c.cl_val = take((size_t)KL * PMDI_CLS_LDS * 4);
    c.bm_fresh = take((size_t)((a.P >> 6) + 1) * 8);
    c.bm_clone = take((size_t)((a.P >> 6) + 1) * 8);
    c.bm_keep = take((size_t)((a.P >> 6) + 1) * 8);
    c.bm_reuse = take((size_t)((a.P >> 6) + 1) * 8);
    c.fl = take((size_t)KL * Dp);
    c.news = take((size_t)KL * a.P);
    c.total = o;
}

struct Sh {
    ldbl xs, pis, lw, term, lpl, cdf, red;
    lu64 scan;
    lint pid, col, sid, kv, lead_of, slot_of, cl_lead, cl_val, need, need_slot, item_id, dl, dl_slot;
    HT h1, h2;
    lint ktab_minp, ktab_val, klist, kl_v, kl_key, fl_p, fl_slot, fl_nnew, fl_tgt;
    lu32 bm_fresh, bm_clone, bm_keep, bm_reuse;
    lint leaf_i1, leaf_n;
    ldbl leaf_tot, leaf_carry;
    lu8 leaf_prog;
    lint kmaxid, kncls, kcur, knflag, khint, kncol, lab, misc;
    li64 ph, stat;   // phase timers; the sweep's counters (n_operations, ...), kept by lane 0
    li64 wk;         // work counters per dataset (WK_*), kept by lane 0: what the dedup-aware byte model of bench.py is built from
    lu8 fl, news;
};

enum { M_NEED = 0, M_OVF = 1, M_PSTAR = 2, M_NK = 3, M_NF = 4, M_NCLS = 5, M_NCLONE = 6, M_FAIL = 7, M_NLEAF = 8, M_NPROG = 9, M_ND = 10, M_MOVED = 11, M_XAB = 12 };

// class list of dataset k: slot r -> leader particle / class value.  The first cls_lds slots
// live in LDS, the rest (burn-in only) in global memory.
// A table that lives in LDS when the configuration lets it fit and in the chain's arena otherwise.
// Which one is a launch constant: a uniform branch picks ds_* or global_* instructions (a generic
// pointer would make every access a flat instruction).
template <class Tp>
struct Dual {
    __attribute__((address_space(3))) Tp *l;
    PMDI_GLOBAL Tp *g;
    bool lds;
    struct Ref {
        const Dual &d;
        size_t i;
        __device__ __forceinline__ operator Tp() const { return d.lds ? d.l[i] : d.g[i]; }
        __device__ __forceinline__ Tp operator=(Tp v) const { if (d.lds) d.l[i] = v; else d.g[i] = v; return v; }
        __device__ __forceinline__ Tp operator+=(Tp v) const { const Tp w = Tp(*this) + v; *this = w; return w; }
    };
    __device__ __forceinline__ Ref operator[](size_t i) const { return Ref{*this, i}; }
    __device__ __forceinline__ Dual operator+(size_t off) const { return Dual{l + off, g + off, lds}; }
};
template <class Tp>
__device__ __forceinline__ Dual<Tp> dual(bool lds, __attribute__((address_space(3))) Tp *l, PMDI_GLOBAL Tp *g)
{
    return Dual<Tp>{l, g, lds};
}
// Dual whose global side is rebuilt from the argument block where it is used (pmdi_device.h, LazyArr)
template <class Tp, size_t DsetDev::*OFF>
struct DualL {
    __attribute__((address_space(3))) Tp *l;
    LazyArr<Tp, OFF> g;
    size_t off;
    bool lds;
    struct Ref {
        const DualL &d;
        size_t i;
        __device__ __forceinline__ operator Tp() const { return d.lds ? d.l[i] : d.g.p()[d.off + i]; }
        __device__ __forceinline__ Tp operator=(Tp v) const { if (d.lds) d.l[i] = v; else d.g.p()[d.off + i] = v; return v; }
        __device__ __forceinline__ Tp operator+=(Tp v) const { const Tp w = Tp(*this) + v; *this = w; return w; }
    };
    __device__ __forceinline__ Ref operator[](size_t i) const { return Ref{*this, i}; }
    __device__ __forceinline__ DualL operator+(size_t o) const { return DualL{l + o, g, off + o, lds}; }
};
template <class Tp, size_t DsetDev::*OFF>
__device__ __forceinline__ DualL<Tp, OFF> dual(bool lds, __attribute__((address_space(3))) Tp *l, const LazyArr<Tp, OFF> &g)
{
    return DualL<Tp, OFF>{l, g, 0, lds};
}

struct ClsList {
    lint l_lead, l_val;
    gint g_lead, g_val;
    int cap;
    __device__ __forceinline__ int lead(int r) const { return r < cap ? l_lead[r] : g_lead[r]; }
    __device__ __forceinline__ int val(int r) const { return r < cap ? l_val[r] : g_val[r]; }
    __device__ __forceinline__ void set(int r, int p, int v) const
    {
        if (r < cap) { l_lead[r] = p; l_val[r] = v; } else { g_lead[r] = p; g_val[r] = v; }
    }
};

// Rebuild the class list from pid[]: leader = lowest particle of each class (the particle
// whose CDF the reference caches in fprob_dict, src/pmdi.jl:225-248).  lead_of must be INF
// for every class value on entry; it is INF again on exit.  Returns the number of classes.
template <int T, class DualT>
__device__ __forceinline__ int rebuild_classes(const DualT &pidk, const ClsList &cl, const Sh &sh, int P)
{
    const int tid = threadIdx.x;
    for (int pb = 0; pb < P; pb += T) {
        const int p = pb + tid;
        const bool valid = p < P;
        const int cls = valid ? pidk[p] : 0;
        int cnt;
        if (wave_group(cls, valid, cnt)) atomicMin(gen(&sh.lead_of[cls]), p);
    }
    __syncthreads();
    unsigned long long carry = 0;
    for (int pb = 0; pb < P; pb += T) {
        const int p = pb + tid;
        const bool valid = p < P;
        const int cls = valid ? pidk[p] : 0;
        const bool isl = valid && sh.lead_of[cls] == p;
        unsigned long long tot;
        const unsigned long long ex = block_flag_scan<T>(isl, false, false, tot, gen(sh.scan)) + carry;
        if (isl) { cl.set((int)ex, p, cls); sh.slot_of[cls] = (int)ex; }
        carry += tot;
    }
    __syncthreads();
    for (int r = tid; r < (int)carry; r += T) sh.lead_of[cl.val(r)] = PMDI_INF_I;
    return (int)carry;
}

// position of this workgroup in the launch order.  In the settled-chain kernel's translation unit (PMDI_BSLOT_FROM_TICKET) this code
// runs in that kernel's workgroup, carrying a handed-over chain on: the position is the one the workgroup DREW (SweepArgs::ticket), kept
// under 1 + blockIdx.x.  The general kernel's own launches deal positions by blockIdx.x (a ticket there cost cfg2 6 %: one more
// memory access in every out-of-line function of the step; profiles/README.md r04).
#ifdef PMDI_BSLOT_FROM_TICKET
#define PMDI_BLOCK_SLOT() (a.ticket ? __hip_atomic_load(a.ticket + 1 + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (int)blockIdx.x)
#else
#define PMDI_BLOCK_SLOT() ((int)blockIdx.x)
#endif
#define PMDI_PREAMBLE PMDI_PREAMBLE_K(false)
#define PMDI_PREAMBLE_K(K1_)                                                                  \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                      \
    const SweepArgs &a = *ap;                                                                  \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                             \
    /* split mode: the K datasets of a chain are swept by K cooperating workgroups (blocks b, b+8, ... share a chain's \
       XCD under round-robin placement: speed only, the hand-off is placement-independent) */  \
    const int Kf = a.K;                                                                        \
    const int bslot = a.ksplit ? a.slot_base + (int)((blockIdx.x / (8u * (unsigned)Kf)) * 8u + (blockIdx.x & 7u)) : PMDI_BLOCK_SLOT(); \
    const int kd0 = a.ksplit ? (int)((blockIdx.x >> 3) % (unsigned)Kf) : 0;                    \
    const int chain = bslot < a.n_slots ? (a.chain_order ? a.chain_order[bslot] : bslot) : 0;  \
    const int K = ((K1_) || a.ksplit) ? 1 : a.K, N = a.N, P = a.P, cap = a.cap;                \
    const DsetDev *dsb = a.ds + kd0;                                                           \
    const long long n = a.n, n1 = a.n1;                                                        \
    const unsigned long long seed = a.seed + (unsigned long long)chain;                        \
    const unsigned iter = a.iter;                                                              \
    const int Dp = (a.Dmax + 15) & ~15;                                                        \
    const int H = PMDI_HT_SIZE;                                                                   \
    Sh sh;                                                                                     \
    build_sh(a, smem, sh);                                                                     \
    const gcint s_in = glob(a.s_in) + ((size_t)chain * Kf + kd0) * n;                          \
    const gcint order = glob(a.order) + (size_t)chain * n;                                     \
    const gcdbl Pi = glob(a.Pi) + ((size_t)chain * Kf + kd0) * N;                              \
    const gcdbl logphi = glob(a.logphi) + (size_t)chain * a.npairs;                            \
    const gcu8 flags = a.flags ? glob(a.flags) + (size_t)chain * a.sumD : (gcu8)nullptr;       \
    const gdbl usc = glob(a.uscratch) + ((size_t)chain * Kf + kd0) * P;                        \
    const gint pstar_raw = glob(a.partstar) + ((size_t)chain * Kf + kd0) * P;                  \
    (void)lane; (void)wave; (void)cap; (void)n1; (void)seed; (void)iter; (void)Dp; (void)H; (void)dsb; (void)Kf; (void)bslot; \
    (void)s_in; (void)order; (void)Pi; (void)logphi; (void)flags; (void)usc; (void)pstar_raw

__device__ __forceinline__ int opaque_vgpr(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void build_sh(const SweepArgs &a, unsigned char *smem, Sh &sh)
{
    const int H = PMDI_HT_SIZE;
        Carve c;
        carve_lds(a, c);
        sh.xs = (ldbl)(smem + c.xs); sh.pis = (ldbl)(smem + c.pis); sh.lw = (ldbl)(smem + c.lw);
        sh.term = (ldbl)(smem + c.term); sh.lpl = (ldbl)(smem + c.lpl); sh.cdf = (ldbl)(smem + c.cdf);
        sh.scan = (lu64)(smem + c.scan); sh.red = (ldbl)(smem + c.red);
        sh.pid = (lint)(smem + c.pid); sh.col = (lint)(smem + c.col); sh.sid = (lint)(smem + c.sid); sh.kv = (lint)(smem + c.kv);
        sh.lead_of = (lint)(smem + c.lead_of); sh.slot_of = (lint)(smem + c.slot_of);
        sh.cl_lead = (lint)(smem + c.cl_lead); sh.cl_val = (lint)(smem + c.cl_val);
        sh.need = (lint)(smem + c.need); sh.need_slot = (lint)(smem + c.need_slot);
        sh.item_id = (lint)(smem + c.item_id); sh.dl = (lint)(smem + c.dl); sh.dl_slot = (lint)(smem + c.dl_slot);
        sh.h1.key = (lint)(smem + c.h1k); sh.h1.a = (lint)(smem + c.h1a); sh.h1.b = (lint)smem; sh.h1.mask = (unsigned)H - 1;
        sh.h2.key = (lint)(smem + c.h2k); sh.h2.a = (lint)(smem + c.h2a); sh.h2.b = (lint)(smem + c.h2b); sh.h2.mask = (unsigned)H - 1;
        sh.ktab_minp = (lint)(smem + c.ktab_minp); sh.ktab_val = (lint)(smem + c.ktab_val);
        sh.klist = (lint)(smem + c.klist); sh.kl_v = (lint)(smem + c.kl_v); sh.kl_key = (lint)(smem + c.kl_key);
        sh.fl_p = (lint)(smem + c.fl_p); sh.fl_slot = (lint)(smem + c.fl_slot); sh.fl_nnew = (lint)(smem + c.fl_nnew);
        sh.fl_tgt = (lint)(smem + c.fl_tgt);
        sh.bm_fresh = (lu32)(smem + c.bm_fresh); sh.bm_clone = (lu32)(smem + c.bm_clone);
        sh.bm_keep = (lu32)(smem + c.bm_keep); sh.bm_reuse = (lu32)(smem + c.bm_reuse);
        sh.leaf_i1 = (lint)(smem + c.leaf_i1); sh.leaf_n = (lint)(smem + c.leaf_n);
        sh.leaf_tot = (ldbl)(smem + c.leaf_tot); sh.leaf_carry = (ldbl)(smem + c.leaf_carry); sh.leaf_prog = (lu8)(smem + c.leaf_prog);
        sh.kmaxid = (lint)(smem + c.kmaxid); sh.kncls = (lint)(smem + c.kncls); sh.kcur = (lint)(smem + c.kcur); sh.knflag = (lint)(smem + c.knflag); sh.khint = (lint)(smem + c.khint); sh.kncol = (lint)(smem + c.kncol);
        sh.lab = (lint)(smem + c.term); sh.misc = (lint)(smem + c.misc); sh.ph = (li64)(smem + c.ph); sh.stat = (li64)(smem + c.stat);
        sh.wk = (li64)(smem + c.wk);
        sh.fl = (lu8)(smem + c.fl); sh.news = (lu8)(smem + c.news);
}

// Cold paths live in __noinline__ functions (their loop-invariant values would otherwise be
// hoisted across the whole sweep loop and spill the hot path's registers).  Each one rebuilds
// its view of the arguments and of the LDS table.

// (the leaf decomposition of the resampling cumsum, as a macro: built once per sweep in the prefix)
#define PMDI_BUILD_LEAF_PROGRAM()                                                                       \
    do {                                                                                                \
            int nl = 0, np = 0, sp = 0;                                                                 \
            int st_i1[24], st_n[24], st_stage[24];                                                      \
            bool ok = P > 1;                                                                            \
            if (ok) { st_i1[0] = 1; st_n[0] = P - 1; st_stage[0] = 0; }                                 \
            while (ok && sp >= 0) {                                                                     \
                const int i1 = st_i1[sp], nn = st_n[sp];                                                \
                if (nn < 128) {                                                                         \
                    if (nl >= 64 || np >= 255) { ok = false; break; }                                   \
                    sh.leaf_i1[nl] = i1; sh.leaf_n[nl] = nn; ++nl;                                      \
                    sh.leaf_prog[np++] = 0; --sp;                                                       \
                } else if (st_stage[sp] == 0) {                                                         \
                    if (np >= 255) { ok = false; break; }                                               \
                    st_stage[sp] = 1; sh.leaf_prog[np++] = 3;                                           \
                    ++sp; st_i1[sp] = i1; st_n[sp] = nn >> 1; st_stage[sp] = 0;                         \
                } else if (st_stage[sp] == 1) {                                                         \
                    if (np >= 255) { ok = false; break; }                                               \
                    st_stage[sp] = 2; sh.leaf_prog[np++] = 1;                                           \
                    const int n2 = nn >> 1;                                                             \
                    ++sp; st_i1[sp] = i1 + n2; st_n[sp] = nn - n2; st_stage[sp] = 0;                    \
                } else {                                                                                \
                    if (np >= 255) { ok = false; break; }                                               \
                    sh.leaf_prog[np++] = 2; --sp;                                                       \
                }                                                                                       \
            }                                                                                           \
            sh.misc[M_NLEAF] = ok ? nl : 0;                                                             \
            sh.misc[M_NPROG] = ok ? np : 0;                                                             \
    } while (0)

// reset (src/pmdi.jl:165-171) and known prefix (src/pmdi.jl:188-207)
template <int T>
__device__ PMDI_COLD_PREFIX void sweep_prefix(const SweepArgs *__restrict__ ap)
{
    PMDI_PREAMBLE;
    if (tid == 0) {
        // leaf decomposition of Julia's accumulate_pairwise! over [1, P) and its recursion as a
        // post-order program (0 leaf, 3 descend left, 1 left done -> right, 2 node done), used by
        // the resampling cumsum.  More than 64 leaves (P > ~4096): the serial form is used.
        PMDI_BUILD_LEAF_PROGRAM();
    }
    __syncthreads();
    // ---- reset (src/pmdi.jl:165-171) and known prefix (src/pmdi.jl:188-207) ----
    for (int k = 0; k < K; ++k) {
        const DsetDev &d = dsb[k];
        const KS s = make_ks(d, chain);
        const int D = d.D;
        const auto pidk = dual(a.pid_lds != 0, sh.pid + (size_t)k * P, s.pid);
        const auto colk = dual(a.col_lds != 0, sh.col + (size_t)k * P, s.col);
        unsigned char *flk = gen(sh.fl + (size_t)k * Dp);
        for (int idx = tid; idx <= cap; idx += T) { s.counts[idx] = 0; s.ncop[idx] = 0; s.firstc[idx] = PMDI_INF_I; }
        for (int idx = tid; idx < N * P; idx += T) { s.newid[idx] = 0; s.cgrp[idx] = 0; }
        // every particle starts on the same column of particle[:, :, k] (:169-171): one column is stored
        for (int p = tid; p < P; p += T) { pidk[p] = 1; colk[p] = 0; }
        for (int u = tid; u < 256; u += T) { sh.lab[u] = PMDI_INF_I; sh.lab[256 + u] = 0; sh.lab[512 + u] = 0; }
        for (int q = tid; q < D; q += T) flk[q] = flags ? flags[d.flag_off + q] : (unsigned char)1;
        for (int nn = tid; nn < N; nn += T) sh.pis[k * N + nn] = Pi[(size_t)k * N + nn];
        __syncthreads();
        // unique(s[order_obs[1:n1-1], k]) in first-appearance order (:192)
        for (long long j = tid; j < n1 - 1; j += T) {
            const int u = s_in[(size_t)k * n + order[j]];
            atomicMin(gen(&sh.lab[u]), (int)j);
            atomicAdd(gen(&sh.lab[512 + u]), 1);
        }
        __syncthreads();
        if (tid < N) {
            const int fp = sh.lab[tid];
            if (fp != PMDI_INF_I) {
                int r = 0;
                for (int v = 0; v < N; ++v) r += (sh.lab[v] < fp) ? 1 : 0;
                sh.lab[256 + tid] = 2 + r;        // cluster id of label u (:197)
            }
        }
        __syncthreads();
        int nu = 0;
        for (int v = 0; v < N; ++v) nu += (sh.lab[v] != PMDI_INF_I) ? 1 : 0;
        // particle[u, :, k] .= id ; counts (:195-198)
        for (int nn = tid; nn < N; nn += T) {
            const int id = sh.lab[256 + nn];
            s.part[0][nn] = id ? id : 1;
        }
        if (tid < N && sh.lab[256 + tid]) s.counts[sh.lab[256 + tid]] = P;
        if (tid == 0) { s.counts[1] = P * N - nu * P; s.cn[1] = 0; }
        if (tid < N && sh.lab[256 + tid]) s.cn[sh.lab[256 + tid]] = sh.lab[512 + tid];
        // fresh clusters 1..nu+1 (:189,:194)
        if (d.kind == K_GAUSSIAN) {
            for (int it = tid; it < (nu + 1) * D; it += T) {
                st2(s.sb, D + it, make_double2(0.0, 0.5));
            }
        } else if (d.kind == K_CATEGORICAL) {
            for (int it = tid; it < (nu + 1) * D * d.L; it += T) s.cnt[(size_t)D * d.L + it] = 0;
        } else {
            for (int it = tid; it < (nu + 1) * D; it += T) s.nbs[D + it] = 0;
        }
        __syncthreads();
        // the first n1-1 shuffled observations join their previous cluster, sequentially in
        // shuffled order (:201-206); lanes = (label, feature)
        for (int it = tid; it < N * D; it += T) {
            const int u = it / D, q = it - u * D;
            const int id = sh.lab[256 + u];
            if (!id || !flk[q]) continue;
            if (d.kind == K_GAUSSIAN) {
                double2 sb = make_double2(0.0, 0.5);
                int c = 0;
                for (long long j = 0; j < n1 - 1; ++j) {
                    const int i = order[j];
                    if (s_in[(size_t)k * n + i] != u) continue;
                    ++c;
                    gauss_add_sb(glob(d.xf)[(size_t)i * D + q], c, sb);
                }
                st2(s.sb, (size_t)id * D + q, sb);
            } else if (d.kind == K_CATEGORICAL) {
                const gint cn_ = s.cnt + ((size_t)id * D + q) * d.L;
                for (long long j = 0; j < n1 - 1; ++j) {
                    const int i = order[j];
                    if (s_in[(size_t)k * n + i] != u) continue;
                    cn_[glob(d.xi)[(size_t)i * D + q] - 1] += 1;
                }
            } else {
                long long S = 0;
                for (long long j = 0; j < n1 - 1; ++j) {
                    const int i = order[j];
                    if (s_in[(size_t)k * n + i] != u) continue;
                    S += glob(d.xi)[(size_t)i * D + q];
                }
                s.nbs[(size_t)id * D + q] = S;
            }
        }
        if (tid == 0) {
            sh.kmaxid[k] = nu + 1;
            sh.kncls[k] = 1;
            sh.kcur[k] = 0;
            sh.kncol[k] = 1;
            int nf = 0;
            for (int q = 0; q < D; ++q) nf += flk[q];
            sh.knflag[k] = nf;
            const ClsList cl{sh.cl_lead + k * PMDI_CLS_LDS, sh.cl_val + k * PMDI_CLS_LDS, s.clslead, s.clsval, PMDI_CLS_LDS};
            cl.set(0, 0, 1);
        }
        __syncthreads();
    }

}

// A chain the settled-chain kernel (pmdi_sweep2_body.h, hand_over) gave up at some observation: instead of the reset and the known
// prefix, the state of the chain as that kernel left it in the arena -- log-weights, column and class of every particle, counters,
// per-dataset scalars -- and the class lists rebuilt from the particles' classes (leader = lowest particle: src/pmdi.jl:225-248).
template <int T>
__device__ __noinline__ void sweep_resume_load(const SweepArgs *__restrict__ ap)
{
    PMDI_PREAMBLE;
    if (tid == 0) { PMDI_BUILD_LEAF_PROGRAM(); }
    const int *rec = a.resume + (size_t)chain * 16;
    for (int p = tid; p < P; p += T) sh.lw[p] = usc[p];
    for (int k = 0; k < K; ++k) {
        const DsetDev &d = dsb[k];
        const KS s = make_ks(d, chain);
        const int D = d.D;
        const auto pidk = dual(a.pid_lds != 0, sh.pid + (size_t)k * P, s.pid);
        const auto colk = dual(a.col_lds != 0, sh.col + (size_t)k * P, s.col);
        unsigned char *flk = gen(sh.fl + (size_t)k * Dp);
        if (a.pid_lds) for (int p = tid; p < P; p += T) sh.pid[(size_t)k * P + p] = s.pid[p];
        if (a.col_lds) for (int p = tid; p < P; p += T) sh.col[(size_t)k * P + p] = s.col[p];
        // (the settled-chain kernel keeps its per-column masks where the copy-on-write split keeps its step-tagged scratch)
        for (int idx = tid; idx < N * P; idx += T) s.cgrp[idx] = 0;
        for (int q = tid; q < D; q += T) flk[q] = flags ? flags[d.flag_off + q] : (unsigned char)1;
        for (int nn = tid; nn < N; nn += T) sh.pis[k * N + nn] = Pi[(size_t)k * N + nn];
        __syncthreads();
        if (tid == 0) {
            sh.kmaxid[k] = a.kstate[((size_t)chain * PMDI_KMAX_I + k) * 2];
            sh.kcur[k] = 0;
            sh.kncol[k] = rec[2 + k];
            int nf = 0;
            for (int q = 0; q < D; ++q) nf += flk[q];
            sh.knflag[k] = nf;
        }
        const ClsList cl{sh.cl_lead + k * PMDI_CLS_LDS, sh.cl_val + k * PMDI_CLS_LDS, s.clslead, s.clsval, PMDI_CLS_LDS};
        const int ncls = rebuild_classes<T>(pidk, cl, sh, P);
        if (tid == 0) sh.kncls[k] = ncls;
        __syncthreads();
    }
    if (tid < 6) sh.stat[tid] = a.stats[(size_t)chain * 8 + tid];
    if (a.work && tid < K * 8) sh.wk[tid] = a.work[((size_t)chain * PMDI_KMAX_I) * 8 + tid];
    __syncthreads();
}

// The fallback step's list of distinct chosen clusters, (src id, updated id, new n) per entry.  The
// fast path's tables are idle during a fallback step, so the list lives in two LDS regions of theirs
// (need .. dl: 8 * item_cap + 3 * PMDI_DL_LDS ints, contiguous; fl_p .. fl_tgt: 4 * PMDI_HT_SIZE ints)
// and only what exceeds them (> 1 492 clusters at item_cap = 256) goes to the arena.  ktab_minp lies inside the first
// region: the fallback step re-arms it (INF) before it returns.
struct DList {
    lint a, b;
    gint g;
    int P, CA;                                                 // CA = (8 * item_cap + 3 * PMDI_DL_LDS) / 3
    static constexpr int CB = (4 * PMDI_HT_SIZE) / 3;
    __device__ __forceinline__ void set(int j, int src, int dst, int nnew) const
    {
        if (j < CA) { a[j] = src; a[CA + j] = dst; a[2 * CA + j] = nnew; }
        else if (j < CA + CB) { const int e = j - CA; b[e] = src; b[CB + e] = dst; b[2 * CB + e] = nnew; }
        else { g[j] = src; g[P + j] = dst; g[2 * P + j] = nnew; }
    }
    __device__ __forceinline__ int src(int j) const { return j < CA ? a[j] : (j < CA + CB ? b[j - CA] : g[j]); }
    __device__ __forceinline__ void get(int j, int &src_, int &dst, int &nnew) const
    {
        if (j < CA) { src_ = a[j]; dst = a[CA + j]; nnew = a[2 * CA + j]; }
        else if (j < CA + CB) { const int e = j - CA; src_ = b[e]; dst = b[CB + e]; nnew = b[2 * CB + e]; }
        else { src_ = g[j]; dst = g[P + j]; nnew = g[2 * P + j]; }
    }
};

// deepcopy + cluster_add! of every distinct chosen cluster (src/pmdi.jl:297,:300): lanes =
// (cluster, feature).  item(j, src, dst, nnew) names the j-th chosen cluster.  The clusters of a
// chain that still carries hundreds of private copies are spread over megabytes of pool, so
// every statistic is an HBM miss: a lane fetches U of them before it touches the first (U = 4; 8 in
// the 256-register wide build -- in the 128-register build 8 spills and is slower).
template <int T, int U, class Item>
__device__ __forceinline__ void stats_update_all(const DsetDev &d, const KS &s, const unsigned char *flk, const double *xs,
                                                 int nd, int D, int tid, Item item)
{
    const int total = nd * D;
    int it = tid;
    if (d.kind == K_GAUSSIAN) {
        for (; it + (U - 1) * T < total; it += U * T) {
            int src[U], dst[U], nnew[U], q[U];
            double2 sb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = it + u * T, j = t / D;
                q[u] = t - j * D;
                item(j, src[u], dst[u], nnew[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) sb[u] = ld2(s.sb, (size_t)src[u] * D + q[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool on = flk[q[u]] != 0;
                if (on) gauss_add_sb(xs[q[u]], nnew[u], sb[u]);
                if (on || dst[u] != src[u]) st2(s.sb, (size_t)dst[u] * D + q[u], sb[u]);
            }
        }
    }
    for (; it < total; it += T) {
        const int j = it / D, q = it - j * D;
        int src, dst, nnew;
        item(j, src, dst, nnew);
        stats_update_one(d, s, flk[q], xs, src, dst, nnew, D, q);
    }
}

// The particle -> cluster table particle[:, :, k] (N x P, src/pmdi.jl:131) is kept by DISTINCT COLUMN: tab[c*N + label] for the
// live columns c = 0..ncol-1 and a column index per particle.  A settled chain holds about ten distinct columns for its 1 024
// particles (scripts/column_stats.py), so what the reference does per particle -- the gather particle[:, partstar, k] of every
// resampling event (:322), the relabelling (:331-337), the remap of a cloned label (:301-308) -- is done per column.  The
// column numbering is internal: results do not depend on it.
//
// columns_apply = the copy-on-write remap `particle[s_id, part, k] = id` (:301-308) of one step.  wr(p, c, tgt) names particle p's
// chosen cluster c and the id it is updated under (tgt != c: it was cloned, the particle's table entry under its new label
// changes).  Particles of one column that chose the same label move together: such a group takes a copy of the column with
// that entry replaced -- or the column itself when nobody else stays on it (no particle that does not write, and the first
// group to ask).  Every live column keeps at least one particle, so there are never more than P of them.
template <int T, class DualT, class Wr>
__device__ __forceinline__ void columns_apply(const Sh &sh, const KS &s, gint tab, const DualT &colk, int k, int N, int P, int tid_,
                                              unsigned epoch, Wr wr)
{
    const int tid = tid_;
    const int ncol0 = sh.kncol[k];
    const int nbw = 2 * ((P >> 6) + 1);                       // 32-bit words of a bitmap
    int anyw = 0;
    for (int pb = 0; pb < P; pb += T) {                       // which columns keep a particle that does not write?
        const int p = pb + tid;
        const bool valid = p < P;
        int c = 0, tgt = 0, cl = 0;
        if (valid) { wr(p, c, tgt); cl = colk[p]; }
        const bool w = valid && tgt != c;
        anyw |= w ? 1 : 0;
        int cnt;
        if (wave_group_capped(cl, valid && !w, cnt, 8)) atomicOr(gen(&sh.bm_keep[cl >> 5]), 1u << (cl & 31));
    }
    if (!__syncthreads_or(anyw)) {                             // nothing was cloned: the table stands
        for (int e = tid; e < nbw; e += T) sh.bm_keep[e] = 0;
        return;
    }
    // One owner per (column, label) group makes the group's column.  The scratch entry of a key carries the step it was
    // claimed in (its epoch, unique within the sweep and the dataset), so nothing has to be cleared between steps.
    const unsigned long long etag = (unsigned long long)epoch << 32;
    for (int pb = 0; pb < P; pb += T) {
        const int p = pb + tid;
        const bool valid = p < P;
        int c = 0, tgt = 0, cl = 0, ns = 0;
        if (valid) { wr(p, c, tgt); cl = colk[p]; ns = sh.news[k * P + p]; }
        const bool w = valid && tgt != c;
        const int key = cl * N + ns;
        int cnt;
        if (wave_group_capped(key, w, cnt, 8)) {
            unsigned long long *e = gen(s.cgrp + key);
            const unsigned long long old = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old >> 32) != epoch && atomicCAS(e, old, etag) == old) {
                const unsigned bit = 1u << (cl & 31);
                const bool keep = (sh.bm_keep[cl >> 5] & bit) != 0;
                const bool inplace = !keep && !(atomicOr(gen(&sh.bm_reuse[cl >> 5]), bit) & bit);
                int newc = cl;
                if (!inplace) {                                // the originals are read here; in-place entries are written after the barrier
                    newc = atomicAdd(gen(&sh.kncol[k]), 1);
                    if (newc < P) {
                        for (int nn = 0; nn < N; nn += 4) {
                            int v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = (nn + u < N) ? tab[(size_t)cl * N + nn + u] : 0;
#pragma unroll
                            for (int u = 0; u < 4; ++u) if (nn + u < N) tab[(size_t)newc * N + nn + u] = (nn + u == ns) ? tgt : v[u];
                        }
                    }
                }
                __hip_atomic_store(e, etag | (unsigned long long)(unsigned)(inplace ? ((cl + 1) | 0x40000000) : (newc + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    for (int pb = 0; pb < P; pb += T) {
        const int p = pb + tid;
        if (p < P) {
            int c, tgt;
            wr(p, c, tgt);
            if (tgt != c) {
                const int cl = colk[p], ns = sh.news[k * P + p];
                const unsigned v = (unsigned)__hip_atomic_load(gen(s.cgrp + (cl * N + ns)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (sc1: from L2, where the owner's CAS and store went)
                if (v & 0x40000000u) tab[(size_t)cl * N + ns] = tgt;      // the group kept its column: every member writes the same value
                colk[p] = (int)(v & 0x3fffffffu) - 1;
            }
        }
    }
    for (int e = tid; e < nbw; e += T) { sh.bm_keep[e] = 0; sh.bm_reuse[e] = 0; }
    if (tid == 0) sh.wk[k * 8 + WK_SPLITS] += sh.kncol[k] - ncol0;
    // (the caller's next barrier orders these writes before the next step's reads)
}

// Burn-in (the class x label items outgrow the LDS tables): every live cluster is evaluated, as the reference does
// (src/pmdi.jl:218-220).  There are hundreds to thousands of them, so the lanes are the CLUSTERS: each lane walks its
// cluster's features in order (the same terms added in the same order as the staged form of the fast path), four pool
// reads in flight.  (Staging the terms in LDS allowed 10 clusters per round at D = 200: the ordered sums of 2 600 clusters
// took 3.3 M cycles per step, 63 % of cfg5's first sweep.)  Out of line: its registers are not the step loop's.
template <int T>
__device__ __noinline__ void sweep_logprob_all(const SweepArgs *__restrict__ ap, int k, int maxid)
{
    PMDI_PREAMBLE;
    const DsetDev &d = dsb[k];
    const KS s = make_ks(d, chain);
    const int D = d.D;
    const unsigned char *flk = gen(sh.fl + (size_t)k * Dp);
    const int nflag = sh.knflag[k];
    for (int j0 = 0; j0 < maxid; j0 += T) {
        const int id = 1 + j0 + tid;
        if (id <= maxid) {
            const int cn = s.cn[id];
            double out;
            if (d.kind == K_GAUSSIAN) {
                out = (double)nflag * glob(d.gtab)[cn];
                for (int q0 = 0; q0 < D; q0 += 4) {
                    double2 sb4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) sb4[u] = ld2(s.sb, (size_t)id * D + min(q0 + u, D - 1));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = q0 + u;
                        if (q < D && flk[q]) {
                            double ta, tb;
                            gauss_terms(sh.xs[q], (double)cn, gauss_ml(cn, sb4[u]), ta, tb);
                            out += ta; out -= tb;
                        }
                    }
                }
            } else if (d.kind == K_CATEGORICAL) {
                double acc = 0.0;                                  // categorical_cluster.jl:30
                for (int q = 0; q < D; ++q) if (flk[q]) acc += glob(d.lhtab)[glob(d.maxcol)[q] + 2 * cn];
                out = -acc;
                for (int q = 0; q < D; ++q)
                    if (flk[q]) {
                        const int x = ((const int *)sh.xs)[q];
                        const int c = s.cnt[((size_t)id * D + q) * d.L + (x - 1)];
                        out += (cn == 0) ? glob(d.lhtab)[1] : glob(d.lhtab)[2 * c + 1];
                    }
            } else {
                out = 0.0;                                         // negbinom_cluster.jl:25
                for (int q = 0; q < D; ++q)
                    if (flk[q]) out += negbin_term(glob(d.lgtab), cn, ((const int *)sh.xs)[q], s.nbs[(size_t)id * D + q]);
            }
            s.lp[id] = out;
        }
    }
    __syncthreads();
}

// One (observation, dataset) step on the fallback path: per-particle class keys, ballot scans,
// per-id tables in global memory.  `converted`: the fast path already drew the allocations but
// its LDS census overflowed.  Results (clones, classes, pool overflow) go back through sh.misc.
template <int T, int WPS>
__device__ PMDI_COLD_SLOW void sweep_slow(const SweepArgs *__restrict__ ap, int k, int i, long long pos, bool small,
                                        bool converted, int maxid, int ncls, long long &ph_last, int &ph_cur)
{
    PMDI_PREAMBLE;
#define PHS(i_)                                                                 \
    do {                                                                        \
        if (a.phase && tid == 0) {                                              \
            const long long t_ = clock64();                                     \
            sh.ph[ph_cur] += t_ - ph_last; ph_last = t_; ph_cur = (i_);         \
        }                                                                       \
    } while (0)
    PHS(12);
    const DsetDev &d = dsb[k];
    const KS s = make_ks(d, chain);
    const int D = d.D;
    const gint part = s.part[sh.kcur[k]];
    const auto pidk = dual(a.pid_lds != 0, sh.pid + (size_t)k * P, s.pid);
    const auto colk = dual(a.col_lds != 0, sh.col + (size_t)k * P, s.col);
    const auto sidp = dual(a.pp_lds != 0, sh.sid, s.sid);
    const auto kvp = dual(a.pp_lds != 0, sh.kv, s.kv);
    const unsigned char *flk = gen(sh.fl + (size_t)k * Dp);
    const ClsList cl{sh.cl_lead + k * PMDI_CLS_LDS, sh.cl_val + k * PMDI_CLS_LDS, s.clslead, s.clsval, PMDI_CLS_LDS};
    const int items = ncls * N;
    const auto cdfp = dual(small, sh.cdf, s.cdf);
    const DList dl{sh.need, sh.fl_p, s.dl, P, (8 * a.item_cap + 3 * PMDI_DL_LDS) / 3};
    int nd = 0, nclone = 0, new_ncls = 0, failed = 0;
    (void)items;
    if (converted) {
                    for (int w = tid; w < items && w < a.item_cap; w += T) sh.ktab_minp[w] = PMDI_INF_I;
                    for (int e = tid; e < H; e += T) { sh.h2.key[e] = 0; sh.h2.a[e] = 0; sh.h2.b[e] = PMDI_INF_I; }
                    for (int pb = 0; pb < P; pb += T) {
                        const int p = pb + tid;
                        const bool valid = p < P;
                        int c = 0, key = 0;
                        bool fresh = false;
                        if (valid) {
                            const int ns = sh.news[k * P + p];
                            key = (pidk[p] - 1) * N + ns;
                            c = part[(size_t)colk[p] * N + ns];
                            const int v = s.newid[key];
                            sidp[p] = c;
                            kvp[p] = v;
                            fresh = v <= 0;
                        }
                        int cnt;
                        if (wave_group(key, fresh, cnt)) atomicMin(gen(&s.newid[key]), p - P);
                        if (wave_group(c, valid, cnt)) { atomicAdd(gen(&s.ncop[c]), cnt); atomicMin(gen(&s.firstc[c]), p); }
                    }
        __syncthreads();
    }
    bool gcensus = converted;
    {
                const bool direct = !converted && sh.khint[k] != 0;   // skip the LDS census attempt
                if (!converted) {
                    for (int pb = 0; pb < P; pb += T) {
                        const int p = pb + tid;
                        const bool valid = p < P;
                        int ns = 0, c = 0, key = 0;
                        bool fresh = false;
                        if (valid) {
                            const int cls = pidk[p];
                            const auto row = cdfp + (size_t)sh.slot_of[cls] * (N + 2);
                            if (p != 0) {
                                const double u = uniform01(seed, iter, (unsigned)pos, (unsigned)(kd0 + k), (unsigned)p, SITE_DRAW);
                                for (int t = 0; t < N - 1; ++t) {
                                    if (row[ns] > u) break;
                                    ++ns;
                                }
                            } else {
                                ns = s_in[(size_t)k * n + i];            // reference trajectory (:262)
                            }
                            if (!a.ksplit) {
                                sh.lw[p] += row[N];
                            } else {
                                const size_t xo = (((size_t)chain * 2 + (size_t)((pos - (n1 - 1)) & 1)) * Kf + kd0) * P + p;
                                __hip_atomic_store((unsigned long long *)a.xinc + xo, (unsigned long long)__double_as_longlong((double)row[N]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(a.xlab + xo, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            key = (cls - 1) * N + ns;
                            c = part[(size_t)colk[p] * N + ns];          // sstar_id (:264)
                            const int v = s.newid[key];
                            sidp[p] = c;
                            kvp[p] = v;
                            sh.news[k * P + p] = (unsigned char)ns;
                            s.sstar[(size_t)pos * P + p] = (unsigned char)ns;   // (:265)
                            fresh = v <= 0;
                        }
                        int cnt;
                        if (wave_group(key, fresh, cnt)) atomicMin(gen(&s.newid[key]), p - P);
                        if (wave_group(c, valid, cnt)) {
                            if (direct) {
                                atomicAdd(gen(&s.ncop[c]), cnt); atomicMin(gen(&s.firstc[c]), p);
                            } else {
                                bool won;
                                const int slot = ht_insert(sh.h2, c, won, 48);
                                if (slot < 0) sh.misc[M_OVF] = 1;
                                else { atomicAdd(gen(&sh.h2.a[slot]), cnt); atomicMin(gen(&sh.h2.b[slot]), p); }
                            }
                        }
                    }
                    __syncthreads();
                    gcensus = direct || sh.misc[M_OVF] != 0;
                    if (gcensus && !direct) {   // too many distinct clusters for the LDS table: per-id tables in global memory
                        for (int e = tid; e < H; e += T) { sh.h2.key[e] = 0; sh.h2.a[e] = 0; sh.h2.b[e] = PMDI_INF_I; }
                        for (int pb = 0; pb < P; pb += T) {
                            const int p = pb + tid;
                            const bool valid = p < P;
                            const int c = valid ? sidp[p] : 0;
                            int cnt;
                            if (wave_group(c, valid, cnt)) { atomicAdd(gen(&s.ncop[c]), cnt); atomicMin(gen(&s.firstc[c]), p); }
                        }
                        __syncthreads();
                    }
                }

                PHS(6);
                // -- D: ranks in particle order: fresh class keys (:266-269) and distinct chosen
                // clusters, clone-or-in-place (:276-299)
                unsigned long long carry = 0;
                for (int pb = 0; pb < P; pb += T) {
                    const int p = pb + tid;
                    const bool valid = p < P;
                    int key = 0, c = 0, slot = 0, ncp = 0;
                    bool fk = false, fc = false, nc = false;
                    if (valid) {
                        c = sidp[p];
                        if (kvp[p] <= 0) {
                            key = (pidk[p] - 1) * N + sh.news[k * P + p];
                            fk = s.newid[key] == p - P;
                        }
                        if (gcensus) { fc = s.firstc[c] == p; if (fc) ncp = s.ncop[c]; }
                        else { slot = ht_find(sh.h2, c); fc = sh.h2.b[slot] == p; ncp = sh.h2.a[slot]; }
                        nc = fc && (ncp != s.counts[c]);
                    }
                    unsigned long long tot;
                    const unsigned long long ex = block_flag_scan<T>(fk, fc, nc, tot, gen(sh.scan)) + carry;
                    if (fk) s.newid[key] = (int)(ex & 0xfffffull) + 1;
                    if (fc) {
                        const int rc = (int)((ex >> 20) & 0xfffffull);
                        const int tgt = nc ? maxid + (int)(ex >> 40) + 1 : c;
                        if (tgt <= cap) {
                            const int nnew = s.cn[c] + 1;
                            if (nc) { s.counts[c] -= ncp; s.counts[tgt] = ncp; }   // (:293-294)
                            s.cn[tgt] = nnew;
                            dl.set(rc, c, tgt, nnew);
                            if (rc < PMDI_DL_LDS) sh.dl_slot[rc] = slot;
                            if (gcensus) s.ncop[c] = tgt; else sh.h2.a[slot] = tgt;   // chosen id -> updated id
                        }
                    }
                    carry += tot;
                }
                nd = (int)((carry >> 20) & 0xfffffull);
                nclone = (int)(carry >> 40);
                if (maxid + nclone > cap) failed = 1;
                __syncthreads();
                if (failed) { if (tid == 0) sh.misc[M_FAIL] = 1; __syncthreads(); return; }

                PHS(7);
                // -- E: apply: new class ids, remap cloned labels (:301-308)
                columns_apply<T>(sh, s, part, colk, k, N, P, tid, (unsigned)(pos - (n1 - 1)) + 1u, [&](int p, int &c, int &tgt) {
                    c = sidp[p];
                    tgt = gcensus ? s.ncop[c] : sh.h2.a[ht_find(sh.h2, c)];
                });
                for (int pb = 0; pb < P; pb += T) {
                    const int p = pb + tid;
                    const bool valid = p < P;
                    int newcls = 0;
                    if (valid) {
                        const int ns = sh.news[k * P + p];
                        const int key = (pidk[p] - 1) * N + ns;
                        const int v = kvp[p];
                        newcls = (v <= 0) ? s.newid[key] : v;
                        pidk[p] = newcls;
                        kvp[p] = key;
                    }
                    int cnt;
                    if (wave_group(newcls, valid, cnt)) atomicMin(gen(&sh.lead_of[newcls]), p);
                }
                __syncthreads();

                // -- F: class list for the next step; scratch clean-up; sufficient-statistic update
                // of every distinct chosen cluster (deepcopy + cluster_add!, :297,:300):
                // lanes = (cluster, feature)
                PHS(8);
                {
                    unsigned long long ccarry = 0;
                    for (int pb = 0; pb < P; pb += T) {
                        const int p = pb + tid;
                        const bool valid = p < P;
                        const int cls = valid ? pidk[p] : 0;
                        const bool isl = valid && sh.lead_of[cls] == p;
                        unsigned long long tot;
                        const unsigned long long ex = block_flag_scan<T>(isl, false, false, tot, gen(sh.scan)) + ccarry;
                        if (isl) { cl.set((int)ex, p, cls); sh.slot_of[cls] = (int)ex; }
                        if (valid && a.q1 == 1) s.newid[kvp[p]] = 0;   // corrected mode: new_id per step
                        ccarry += tot;
                    }
                    if (gcensus) {
                        for (int j = tid; j < nd; j += T) {
                            const int c = dl.src(j);
                            s.ncop[c] = 0; s.firstc[c] = PMDI_INF_I;
                        }
                    } else if (nd <= PMDI_DL_LDS) {
                        for (int j = tid; j < nd; j += T) { const int sl = sh.dl_slot[j]; sh.h2.key[sl] = 0; sh.h2.a[sl] = 0; sh.h2.b[sl] = PMDI_INF_I; }
                    } else {
                        for (int e = tid; e < H; e += T) { sh.h2.key[e] = 0; sh.h2.a[e] = 0; sh.h2.b[e] = PMDI_INF_I; }
                    }
                    PHS(13);
                    stats_update_all<T, (T >= 512 && WPS <= 2) ? PMDI_VH_U : 4>(d, s, flk, gen(sh.xs), nd, D, tid, [&](int j, int &src, int &dst, int &nnew) {
                        dl.get(j, src, dst, nnew);
                    });
                    new_ncls = (int)ccarry;
                    __syncthreads();
                    for (int r = tid; r < new_ncls; r += T) sh.lead_of[cl.val(r)] = PMDI_INF_I;
                }
    }
    for (int e = tid; e < a.item_cap; e += T) sh.ktab_minp[e] = PMDI_INF_I;        // it lies inside the list's LDS region
    if (tid == 0) {
        sh.misc[M_NCLONE] = nclone; sh.misc[M_NCLS] = new_ncls; sh.misc[M_ND] = nd;
        sh.khint[k] = (gcensus && nd > PMDI_HT_SIZE / 4) ? 1 : 0;   // stay on the global census while it is needed
    }
    __syncthreads();
#undef PHS
}

// draw_partstar (src/misc.jl:27-47), gather and compact renumbering (src/pmdi.jl:318-340)
template <int T>
__device__ PMDI_COLD_RESAMPLE void sweep_resample(const SweepArgs *__restrict__ ap, long long pos, double mx)
{
    // (no phase-timer references in here: a __noinline__ function that takes the address of the step
    // loop's timer variables pins them in memory and costs the 128-register build ~50 more spill slots)
    PMDI_PREAMBLE;
#ifdef PMDI_RESAMPLE_TIMERS      // A/B builds only: where does a resampling event spend its time?  (slots 0..7 of sh.stat are
                                 // the counters; the timer state sits in slots of sh.ph that the sweep does not use here)
#define PHR(i_) do { if (a.phase && tid == 0) { const long long t_ = clock64(); sh.stat[8 + (i_)] += t_ - rs_last; rs_last = t_; } } while (0)
    long long rs_last = clock64();
#else
#define PHR(i_) do { } while (0)
#endif
            // draw_partstar (src/misc.jl:27-47)
            const double u01 = uniform01(seed, iter, (unsigned)pos, 0, 0, SITE_RESAMPLE_U);
            const double usl = uniform01(seed, iter, (unsigned)pos, 0, 0, SITE_RESAMPLE_SLOT);
            double *wb = gen(sh.term);
            for (int p = tid; p < P; p += T) wb[p] = exp(sh.lw[p] - mx);
            __syncthreads();
            PHR(0);   // weights
            // cumsum (:29) in Julia's accumulate_pairwise! order, bit-exactly but in parallel: the
            // recursion splits [1, P) into leaves of < 128 elements; a leaf's running sums s_ do not
            // depend on its carry, so (1) one lane per leaf forms them, (2) one lane walks the tree
            // to get every leaf's carry s = op(s, s_left), (3) c[i] = s + s_[i] for all i.
            const int nleaf = sh.misc[M_NLEAF];
            if (nleaf == 0 && tid == 0) jl_cumsum_inplace(wb, P);
            if (tid < nleaf) {
                const int i1 = sh.leaf_i1[tid], nn = sh.leaf_n[tid];
                double s_ = wb[i1];
                int i = i1 + 1;
                for (; i + 4 <= i1 + nn; i += 4) {
                    const double v0 = wb[i], v1 = wb[i + 1], v2 = wb[i + 2], v3 = wb[i + 3];
                    s_ = s_ + v0; wb[i] = s_;
                    s_ = s_ + v1; wb[i + 1] = s_;
                    s_ = s_ + v2; wb[i + 2] = s_;
                    s_ = s_ + v3; wb[i + 3] = s_;
                }
                for (; i < i1 + nn; ++i) { s_ = s_ + wb[i]; wb[i] = s_; }
                sh.leaf_tot[tid] = s_;
            }
            const bool p_pow2 = (P & (P - 1)) == 0;
            if (p_pow2) {
                // u += 1/particles by repeated addition (:34), in parallel and still bit-exact: h = 1/P is a
                // power of two, so inside a binade [2^e, 2^(e+1)) every u + h is exact (h is a multiple of
                // ulp(u)); only the additions that cross into the next binade round.  Each lane replays the
                // crossings (about ten of them) and jumps over the exact runs in between.
                const double h = 1.0 / (double)P;
                for (int j = tid; j < P; j += T) {
                    double u = u01 / (double)P;
                    int done = 0;
                    while (done < j) {
                        int e;
                        (void)frexp(u, &e);                          // u in [2^(e-1), 2^e)
                        const double B = ldexp(1.0, e);
                        const double x = (B - u) * (double)P;        // exact: B - u (same binade) and the power-of-two scale
                        double m = floor(x);
                        if (m == x) m -= 1.0;                        // largest m with u + m*h < B
                        if (m > (double)(j - done)) m = (double)(j - done);
                        if (m >= 1.0) { u = u + m * h; done += (int)m; }      // exact run inside the binade
                        if (done < j) { u = u + h; done += 1; }               // the crossing step rounds like the loop's
                    }
                    usc[j] = u;
                }
            } else if (tid == T - 64) {                           // general P: the additions round, one lane replays them
                double u = u01 / (double)P;
                const double h = 1.0 / (double)P;
                usc[0] = u;
                int j = 1;
                for (; j + 8 <= P; j += 8) {
                    const double u1 = u + h, u2 = u1 + h, u3 = u2 + h, u4 = u3 + h, u5 = u4 + h, u6 = u5 + h, u7 = u6 + h, u8 = u7 + h;
                    usc[j] = u1; usc[j + 1] = u2; usc[j + 2] = u3; usc[j + 3] = u4;
                    usc[j + 4] = u5; usc[j + 5] = u6; usc[j + 6] = u7; usc[j + 7] = u8;
                    u = u8;
                }
                for (; j < P; ++j) { u += h; usc[j] = u; }
            }
            __syncthreads();
            if (tid == 0 && nleaf > 0) {
                // carries: replay the recursion over the leaf totals.  The decomposition (kernel
                // start) stored the recursion as a post-order program: op 0 = leaf, op 1 = "right
                // child starts: carry = carry_of_node + left total", op 2 = "node done: total =
                // left + right".
                double cs[24], lt[24];       // carry / left-total stacks
                int sp = 0;
                cs[0] = wb[0];
                double ret = 0.0;
                const int nprog = sh.misc[M_NPROG];
                int leaf = 0;
                for (int pc = 0; pc < nprog; ++pc) {
                    const int op = sh.leaf_prog[pc];
                    if (op == 0) { sh.leaf_carry[leaf] = cs[sp]; ret = sh.leaf_tot[leaf]; ++leaf; }
                    else if (op == 3) { cs[sp + 1] = cs[sp]; ++sp; }                 // descend into a left child
                    else if (op == 1) { lt[sp - 1] = ret; cs[sp] = cs[sp - 1] + ret; } // left done: right child's carry
                    else { --sp; ret = lt[sp] + ret; }                                 // node done (s_ += right)
                }
            }
            __syncthreads();
            for (int lf = wave; lf < nleaf; lf += T / 64) {       // c[i] = op(s, s_)
                const int i1 = sh.leaf_i1[lf], nn = sh.leaf_n[lf];
                const double sc = sh.leaf_carry[lf];
                for (int i = i1 + lane; i < i1 + nn; i += 64) wb[i] = sc + wb[i];
            }
            __syncthreads();
            PHR(1);   // cumsum + u sequence
            const double last = wb[P - 1];
            for (int j = tid; j < P; j += T) {
                const double uj = usc[j];
                int lo = 0, hi = P - 1;           // smallest p with pprob[p]/last >= u_j
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (wb[mid] / last >= uj) hi = mid; else lo = mid + 1;
                }
                pstar_raw[j] = lo;
            }
            int js = (int)(usl * (double)P);      // shuffle!, partstar[1]=1, sort! (:43-45)
            if (js >= P) js = P - 1;
            for (int p = tid; p < P; p += T) sh.lw[p] = 1.0;     // src/pmdi.jl:319
            __syncthreads();
            // ancestor of every slot after the conditional-SMC fix-up (slot 0 keeps particle 0)
            const auto ancp = dual(a.pp_lds != 0, sh.kv, make_ks(dsb[0], chain).kv);
            for (int p = tid; p < P; p += T) ancp[p] = (p == 0) ? 0 : (p <= js ? pstar_raw[p - 1] : pstar_raw[p]);
            __syncthreads();
            if (a.q2) {
                // __pmdi() permutes the allocation history on every resampling (src/__pmdi.jl:285: sstar[:,:,k] = sstar[partstar,:,k]);
                // here the ancestor table of the event is logged and the selected particle's lineage is traced back once,
                // at the end of the sweep (sweep_final): same trajectory, n*P bytes moved per event less
                const long long ev = sh.stat[1] - 1;
                const gint lg = glob(a.anclog) + ((size_t)chain * (size_t)(n - n1 + 1) + (size_t)ev) * P;
                for (int p = tid; p < P; p += T) lg[p] = ancp[p];
                if (tid == 0) glob(a.evpos)[(size_t)chain * 2 * (size_t)(n - n1 + 1) + ev] = (int)pos;
            }
            PHR(2);   // search + ancestors
            for (int k = 0; k < K; ++k) {                         // src/pmdi.jl:320-340
                const DsetDev &d = dsb[k];
                const KS s = make_ks(d, chain);
                const int D = d.D;
                const int cur = sh.kcur[k];
                const int oldmax = sh.kmaxid[k];
                const auto pidk = dual(a.pid_lds != 0, sh.pid + (size_t)k * P, s.pid);
                const auto sidp = dual(a.pp_lds != 0, sh.sid, s.sid);
                const ClsList cl{sh.cl_lead + k * PMDI_CLS_LDS, sh.cl_val + k * PMDI_CLS_LDS, s.clslead, s.clsval, PMDI_CLS_LDS};
                const gcint src = s.part[cur];
                const gint dst = s.part[cur ^ 1];
                const auto colk = dual(a.col_lds != 0, sh.col + (size_t)k * P, s.col);
                const int ncol_old = sh.kncol[k];
                // The gather particle[:, partstar, k] (:322) by column: a particle takes its ancestor's column INDEX; the columns
                // that still have a particle are compacted into the other buffer, relabelled on the way (:331-337).  Occupancy
                // of every old id (= its new count, :338) = sum over the live columns of (particles on the column) x (entries
                // holding the id), in LDS (the term buffer and the hash / list tables are idle here) when the ids fit; otherwise
                // the per-id scratch tables in global memory.
                lint hist = sh.h1.key;                            // 9 * PMDI_HT_SIZE contiguous ints
                const bool lm = oldmax + 1 <= 2 * a.terms_cap && oldmax + 1 <= 9 * PMDI_HT_SIZE;
                lint lmap = (lint)sh.term;
                lint mult = sh.slot_of;                           // particles per old column (slot_of is rebuilt at the top of every step)
                lint cmap = sh.lead_of;                           // old column -> new column (INF again on exit)
                const gint tmpc = pstar_raw;                      // (the ancestors were derived from it above: free)
                PHR(3);   // (dataset loop top)
                if (lm) for (int e = tid; e <= oldmax; e += T) { lmap[e] = 0; hist[e] = 0; }
                for (int id = 1 + tid; id <= oldmax; id += T) s.counts[id] = 0;   // (:326)
                for (int c = tid; c < ncol_old; c += T) mult[c] = 0;
                __syncthreads();
                for (int pb = 0; pb < P; pb += T) {               // particle[:, partstar, k], particle_id[partstar, k] (:322-323)
                    const int p = pb + tid;
                    const bool valid = p < P;
                    int nc = 0;
                    if (valid) {
                        const int an = ancp[p];
                        sidp[p] = (int)pidk[an];
                        nc = colk[an];
                        tmpc[p] = nc;
                    }
                    int cnt;
                    if (wave_group_capped(nc, valid, cnt, 8)) atomicAdd(gen(&mult[nc]), cnt);
                }
                __syncthreads();
                PHR(4);   // gather
                unsigned long long ccarry = 0;                    // new index of every column that kept a particle
                for (int b = 0; b < ncol_old; b += T) {
                    const int c = b + tid;
                    const bool live = c < ncol_old && mult[c] != 0;
                    unsigned long long tot;
                    const unsigned long long ex = block_flag_scan<T>(live, false, false, tot, gen(sh.scan)) + ccarry;
                    if (live) cmap[c] = (int)ex;
                    ccarry += tot;
                }
                const int ncol_new = (int)ccarry;
                for (int idx = tid; idx < ncol_old * N; idx += T) {
                    const int m = mult[idx / N];
                    if (m) {
                        const int v = src[idx];
                        if (lm) atomicAdd(gen(&hist[v]), m); else s.ncop[v] = 1;
                    }
                }
                __syncthreads();
                // sort(unique(particle)) ascending -> 1..U' (:329): scan of the live ids
                unsigned long long carry = 0;
                for (int b = 0; b < oldmax; b += T) {
                    const int id = 1 + b + tid;
                    const int occ = (lm && id <= oldmax) ? hist[id] : 0;
                    const bool live = (id <= oldmax) && (lm ? occ != 0 : s.ncop[id] != 0);
                    unsigned long long tot;
                    const unsigned long long ex = block_flag_scan<T>(live, false, false, tot, gen(sh.scan)) + carry;
                    if (live) {
                        if (lm) { lmap[id] = (int)ex + 1; s.counts[(int)ex + 1] = occ; }     // (:338)
                        else s.firstc[id] = (int)ex + 1;
                    }
                    carry += tot;
                }
                const int newmax = (int)carry;
                PHR(5);   // id scan
                __syncthreads();
#define PMDI_NEWID(id_) (lm ? lmap[(id_)] : (s.ncop[(id_)] ? s.firstc[(id_)] : 0))
                const bool moves = newmax > 0 && PMDI_NEWID(newmax) != newmax;   // else ids 1..newmax stay put
                // the live columns, compacted and relabelled (:331-337) -- and, on the global-table path, the recount (:338)
                for (int idx = tid; idx < ncol_old * N; idx += T) {
                    const int c = idx / N;
                    const int m = mult[c];
                    if (m) {
                        const int v = src[idx];
                        const int nv = moves ? PMDI_NEWID(v) : v;
                        dst[(size_t)cmap[c] * N + (idx - c * N)] = nv;
                        if (!lm) atomicAdd(gen(&s.counts[nv]), m);
                    }
                }
                for (int p = tid; p < P; p += T) { colk[p] = cmap[tmpc[p]]; pidk[p] = (int)sidp[p]; }
                __syncthreads();
                for (int c = tid; c < ncol_old; c += T) cmap[c] = PMDI_INF_I;
                if (tid == 0) { sh.kncol[k] = ncol_new; sh.wk[k * 8 + WK_COLS] += ncol_old; }
                PHR(6);   // relabel + recount
                if (lm) {
                    for (int e = tid; e <= oldmax && e < 9 * PMDI_HT_SIZE; e += T)       // tables back to empty (h2.b = INF)
                        hist[e] = (e >= 4 * PMDI_HT_SIZE && e < 5 * PMDI_HT_SIZE) ? PMDI_INF_I : 0;
                }
                PHR(7);   // table reset
                if (moves) {
                    // clusters[k][i] = deepcopy(clusters[k][id]) for id > i, ascending (:336):
                    // batches in ascending order, load -> barrier -> store
                    for (int b = 0; b < oldmax; b += T) {
                        const int id = 1 + b + tid;
                        const int nid = (id <= oldmax) ? PMDI_NEWID(id) : 0;
                        const bool mv = nid != 0 && nid != id;
                        const int v = mv ? s.cn[id] : 0;
                        const unsigned long long bmv = __ballot(mv);
                        if (lane == 0 && bmv) atomicAdd(gen(&sh.misc[M_MOVED]), __popcll(bmv));
                        __syncthreads();
                        if (mv) s.cn[nid] = v;
                    }
                    const long long nitems = (long long)oldmax * D;
                    if (d.kind == K_GAUSSIAN) {
                        for (long long b = 0; b < nitems; b += T) {
                            const long long it = b + tid;
                            const int id = 1 + (int)(it / D), q = (int)(it - (long long)(id - 1) * D);
                            const int nid = (it < nitems) ? PMDI_NEWID(id) : 0;
                            const bool mv = nid != 0 && nid != id;
                            double2 sb = make_double2(0, 0);
                            if (mv) sb = ld2(s.sb, (size_t)id * D + q);
                            __syncthreads();
                            if (mv) st2(s.sb, (size_t)nid * D + q, sb);
                        }
                    } else if (d.kind == K_CATEGORICAL) {
                        const long long itemsL = nitems * d.L;
                        const int DL = D * d.L;
                        for (long long b = 0; b < itemsL; b += T) {
                            const long long it = b + tid;
                            const int id = 1 + (int)(it / DL), r = (int)(it - (long long)(id - 1) * DL);
                            const int nid = (it < itemsL) ? PMDI_NEWID(id) : 0;
                            const bool mv = nid != 0 && nid != id;
                            const int v = mv ? s.cnt[(size_t)id * DL + r] : 0;
                            __syncthreads();
                            if (mv) s.cnt[(size_t)nid * DL + r] = v;
                        }
                    } else {
                        for (long long b = 0; b < nitems; b += T) {
                            const long long it = b + tid;
                            const int id = 1 + (int)(it / D), q = (int)(it - (long long)(id - 1) * D);
                            const int nid = (it < nitems) ? PMDI_NEWID(id) : 0;
                            const bool mv = nid != 0 && nid != id;
                            const long long v = mv ? s.nbs[(size_t)id * D + q] : 0;
                            __syncthreads();
                            if (mv) s.nbs[(size_t)nid * D + q] = v;
                        }
                    }
                }
#undef PMDI_NEWID
                __syncthreads();
                if (!lm) for (int id = 1 + tid; id <= oldmax; id += T) { s.ncop[id] = 0; s.firstc[id] = PMDI_INF_I; }
                __syncthreads();
                PHR(8);   // moves
                // one class before the resampling = one class after it (same value, particle 0 still its lowest member: slot 0 keeps
                // ancestor 0): the class list stands as it is
                const int nc2 = (sh.kncls[k] == 1) ? 1 : rebuild_classes<T>(pidk, cl, sh, P);
                if (tid == 0) {
                    sh.kmaxid[k] = newmax; sh.kncls[k] = nc2; sh.kcur[k] = cur ^ 1;
                    sh.wk[k * 8 + WK_MOVED] += sh.misc[M_MOVED]; sh.wk[k * 8 + WK_MOVE_EVENTS] += moves ? 1 : 0;
                    sh.misc[M_MOVED] = 0;
                }
                __syncthreads();
                PHR(9);   // classes
            }
#undef PHR
}

// particle pick (src/pmdi.jl:345-350), s = sstar[p_star,:,:] (:373), counters
template <int T>
__device__ PMDI_COLD_FINAL void sweep_final(const SweepArgs *__restrict__ ap)
{
    PMDI_PREAMBLE;
    // ---- particle pick (src/pmdi.jl:345-350) + s = sstar[p_star,:,:] (:373) ----
    {
        double mx = -INFINITY;
        for (int p = tid; p < P; p += T) { const double v = sh.lw[p]; mx = (v > mx) ? v : mx; }
        mx = block_max<T>(mx, gen(sh.red));
        double *wb = gen(sh.term);
        for (int p = tid; p < P; p += T) wb[p] = exp(sh.lw[p] - mx);
        __syncthreads();
        if (tid == 0) {   // StatsBase.sample(::Weights): sequential sum and scan, as the oracle
            double sum = 0.0;
            for (int p = 0; p < P; ++p) sum += wb[p];
            const double t = uniform01(seed, iter, 0, 0, 0, SITE_PSTAR) * sum;
            int ip = 0;
            double cw = wb[0];
            while (cw < t && ip < P - 1) { ++ip; cw += wb[ip]; }
            sh.misc[M_PSTAR] = ip;
        }
        __syncthreads();
        const int pstar = sh.misc[M_PSTAR];
        // q2_mode 1 (__pmdi): the history was permuted by every resampling event at or after a position, i.e. the value
        // that ends up in slot p_star was written into the slot its lineage occupied then.  lin[e] = that slot for the
        // positions in (evpos[e-1], evpos[e]]; positions after the last event read slot p_star itself.
        const int nev = a.q2 ? (int)sh.stat[1] : 0;
        const gint evp = a.q2 ? glob(a.evpos) + (size_t)chain * 2 * (size_t)(n - n1 + 1) : (gint)nullptr;
        const gint lin = evp + (n - n1 + 1);
        const gint anc = a.q2 ? glob(a.anclog) + (size_t)chain * (size_t)(n - n1 + 1) * P : (gint)nullptr;
        if (nev > 0) {
            __syncthreads();
            if (tid == 0) {
                int cur = pstar;
                for (int e = nev - 1; e >= 0; --e) { cur = anc[(size_t)e * P + cur]; lin[e] = cur; }
            }
            __syncthreads();
        }
        for (long long pp = tid; pp < n; pp += T) {
            const int i = order[pp];
            int slot = pstar;
            if (nev > 0 && pp >= n1 - 1 && (int)pp <= evp[nev - 1]) {
                int lo = 0, hi = nev - 1;              // first event whose position is >= pp
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (evp[mid] >= (int)pp) hi = mid; else lo = mid + 1; }
                slot = lin[lo];
            }
            for (int k = 0; k < K; ++k) {
                int v;
                if (pp < n1 - 1) v = s_in[(size_t)k * n + i];   // sstar[:, i, k] .= s[i, k] (:204)
                else {
                    const unsigned char *ss = (const unsigned char *)(dsb[k].arena + (size_t)chain * dsb[k].stride + dsb[k].o_sstar);
                    v = ss[(size_t)pp * P + slot];
                }
                a.s_out[((size_t)chain * Kf + kd0 + k) * n + i] = v;
            }
        }
        if (a.lw_out && kd0 == 0) for (int p = tid; p < P; p += T) a.lw_out[(size_t)chain * P + p] = sh.lw[p];
        if (a.pid_lds)   // debug export reads the class ids from global memory
            for (int k = 0; k < K; ++k) {
                int *gp = (int *)(dsb[k].arena + (size_t)chain * dsb[k].stride + dsb[k].o_pid);
                for (int p = tid; p < P; p += T) gp[p] = sh.pid[(size_t)k * P + p];
            }
        if (a.col_lds)   // ... and the column indices
            for (int k = 0; k < K; ++k) {
                int *gc = (int *)(dsb[k].arena + (size_t)chain * dsb[k].stride + dsb[k].o_col);
                for (int p = tid; p < P; p += T) gc[p] = (int)sh.col[(size_t)k * P + p];
            }
        if (tid < K) {
            a.kstate[((size_t)chain * PMDI_KMAX_I + kd0 + tid) * 2] = sh.kmaxid[tid];
            a.kstate[((size_t)chain * PMDI_KMAX_I + kd0 + tid) * 2 + 1] = sh.kcur[tid];
        }
        if (tid == 0) {
            long long *st = a.stats + (size_t)chain * 8;
            if (!a.ksplit) {
                a.pstar[chain] = pstar;
                st[ST_NOPS] = sh.stat[0]; st[ST_NRESAMPLE] = sh.stat[1]; st[ST_NCLONES] = sh.stat[2];
                st[ST_MAXID] = sh.stat[3]; st[ST_SUMCLASSES] = sh.stat[4];
                st[5] = sh.stat[5]; st[6] = sh.stat[6]; st[7] = sh.stat[7];
                if (!a.err_keep) a.err[chain] = 0;
                if (a.requeue) a.requeue[chain] = 0;
            } else {
                // the K workgroups of the chain add their datasets' counters (the host zeroed stats and err before the launch)
                if (kd0 == 0) { a.pstar[chain] = pstar; st[ST_NRESAMPLE] = sh.stat[1]; if (a.requeue_only && a.requeue) a.requeue[chain] = 0; }
                atomicAdd((unsigned long long *)&st[ST_NOPS], (unsigned long long)sh.stat[0]);
                atomicAdd((unsigned long long *)&st[ST_NCLONES], (unsigned long long)sh.stat[2]);
                atomicMax((unsigned long long *)&st[ST_MAXID], (unsigned long long)sh.stat[3]);
                atomicAdd((unsigned long long *)&st[ST_SUMCLASSES], (unsigned long long)sh.stat[4]);
                atomicAdd((unsigned long long *)&st[5], (unsigned long long)sh.stat[5]);
                atomicAdd((unsigned long long *)&st[6], (unsigned long long)sh.stat[6]);
                atomicAdd((unsigned long long *)&st[7], (unsigned long long)sh.stat[7]);
            }
        }
        if (a.work && tid < K * 8) a.work[((size_t)chain * PMDI_KMAX_I + kd0) * 8 + tid] = sh.wk[tid];
    }
}

// The first observation of a chain the settled-chain kernel handed over: its labels are drawn and recorded (:265), the increments and
// the Phi term are in the log-weights.  A dataset whose step that kernel finished is done (`done` mask); the others still need the
// step's bookkeeping -- class ids (:266-272), copy-on-write (:275-310) -- which is what the fallback step does with draws it did
// not make itself (`converted`).  Out of line: nothing of this may cost the step loop a register.
template <int T, int WPS>
__device__ __noinline__ void sweep_replay(const SweepArgs *__restrict__ ap, long long pos, int i, int done)
{
    PMDI_PREAMBLE;
    long long ph_last = 0;
    int ph_cur = 0;
    for (int k = 0; k < K; ++k) {
        if ((done >> k) & 1) continue;
        const DsetDev &d = dsb[k];
        const KS s = make_ks(d, chain);
        const int D = d.D;
        const int maxid = sh.kmaxid[k], ncls = sh.kncls[k];
        const ClsList cl{sh.cl_lead + k * PMDI_CLS_LDS, sh.cl_val + k * PMDI_CLS_LDS, s.clslead, s.clsval, PMDI_CLS_LDS};
        for (int q = tid; q < D; q += T) {
            if (d.kind == K_GAUSSIAN) sh.xs[q] = glob(d.xf)[(size_t)i * D + q]; else ((int *)sh.xs)[q] = glob(d.xi)[(size_t)i * D + q];
        }
        for (int p = tid; p < P; p += T) sh.news[k * P + p] = s.sstar[(size_t)pos * P + p];
        for (int r = tid; r < ncls; r += T) sh.slot_of[cl.val(r)] = r;
        if (tid == 0) sh.misc[M_OVF] = 0;
        __syncthreads();
        sweep_slow<T, WPS>(ap, k, i, pos, false, true, maxid, ncls, ph_last, ph_cur);
        if (sh.misc[M_FAIL]) return;
        if (tid == 0) {
            const int nclone_r = sh.misc[M_NCLONE];
            sh.stat[7] += 1;
            sh.stat[0] += maxid;                  // src/__pmdi.jl:187
            sh.stat[4] += ncls;
            sh.stat[2] += nclone_r;
            if (maxid + nclone_r > sh.stat[3]) sh.stat[3] = maxid + nclone_r;
            sh.kmaxid[k] = maxid + nclone_r; sh.kncls[k] = sh.misc[M_NCLS];
            sh.wk[k * 8 + WK_UPD] += sh.misc[M_ND]; sh.wk[k * 8 + WK_CLONE] += nclone_r;
        }
        __syncthreads();
    }
    // calc_ESS (src/misc.jl:15-25) and the resampling decision (src/pmdi.jl:317) of this observation, as the step loop does them
    double mx = -INFINITY;
    for (int p = tid; p < P; p += T) { const double v = sh.lw[p]; mx = (v > mx) ? v : mx; }
    mx = block_max<T>(mx, gen(sh.red));
    double sa = 0.0, sb2 = 0.0;
    for (int p = tid; p < P; p += T) { const double w = exp(sh.lw[p] - mx); sa += w; sb2 += w * w; }
    block_sum2<T>(sa, sb2, gen(sh.red));
    double ess = (sa * sa) / sb2;
    if (fabs(ess - 0.5 * (double)P) <= 1e-9 * (double)P) {         // (the reference's order of additions when the comparison is that close)
        __syncthreads();
        if (tid == 0) {
            double na = 0.0, nb = 0.0;
            for (int p = 0; p < P; ++p) { const double w = exp(sh.lw[p] - mx); na += w; nb += w * w; }
            sh.red[40] = (na * na) / nb;
        }
        __syncthreads();
        ess = sh.red[40];
    }
    const bool resample = ess <= 0.5 * (double)P;
    if (resample) {
        if (tid == 0) sh.stat[1] += 1;
        sweep_resample<T>(ap, pos, mx);
    }
    if (tid == 0) sh.misc[M_XAB] = resample ? 1 : 0;              // (every log-weight equal again: what the caller's lw_uniform means)
    if (a.trace_on && tid == 0) {
        double *tr = a.trace + ((size_t)chain * (n - n1 + 1) + (pos - (n1 - 1))) * (2 + 2 * Kf);
        tr[0] = ess; tr[1] = resample ? 1.0 : 0.0;
        for (int k = 0; k < K; ++k) { tr[2 + k] = (double)sh.kmaxid[k]; tr[2 + Kf + k] = (double)sh.kncls[k]; }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// WPS = minimum waves per SIMD the register allocation must allow (2 co-resident chains per CU
// at T = 512 need 4)
// K1: single-dataset models (K == 1) get a specialisation in which the dataset index is a
// compile-time 0, so the per-dataset views are loop invariants of the sweep loop.
// MANY: the build for more than 64 labels (its CDF stage holds four labels per lane: registers every other build would pay for)
template <int T, int WPS, bool K1, bool MANY, bool RESUME>
__device__ __forceinline__ void pmdi_sweep_body(const SweepArgs *__restrict__ ap)
{
    PMDI_PREAMBLE_K(K1);
    if (!RESUME) {         // (RESUME: the workgroup is the settled-chain kernel's, its chain is already chosen)
    if (a.start_sig && tid == 0 && (!a.ksplit || bslot < a.n_slots))
        __hip_atomic_fetch_add(a.start_sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (bslot >= a.n_slots) return;            // padding of a split launch (chain slots are dealt in groups of eight)
    // two launches share the chains of a sweep (heavy: wide workgroups, light: narrow ones)
    if (a.group_flag && ((int)a.group_flag[chain] != a.group_sel || bslot < a.rank_lo || bslot >= a.rank_hi)) return;
    // ... or this launch sweeps exactly the chains the settled-chain kernel (pmdi_sweep2.hip) gave back, from the start (PMDI_CONTINUE=0)
    if (a.requeue_only && !a.requeue[chain]) return;
    }
    // hand-off area of the split mode: per (chain, parity of the swept observation, dataset): the log-weight increment and
    // the chosen label of every particle; one arrival counter per chain
    // (the addresses are rebuilt from the argument block where they are used: nothing of this stays live across the step loop)
#define XSPLIT (a.ksplit != 0)
#define XCNT (glob(a.xcnt) + (size_t)chain * 32)
#define XOFF(par_, kd_) ((((size_t)chain * 2 + (size_t)(par_)) * (size_t)a.K + (size_t)(kd_)) * (size_t)a.P)

    const long long t_start = clock64();
#ifdef PMDI_RESAMPLE_TIMERS
    if (tid < 24) sh.stat[tid] = 0;
#else
    if (tid < 8) sh.stat[tid] = 0;
#endif
       // the sweep's counters live in LDS, kept by lane 0: eight 64-bit values less across
                                      // the step loop (the 128-register build: 158 -> 124 spill slots)
    long long ph_last = 0;
    int ph_cur = 0;
#define PH(i_)                                                                  \
    do {                                                                        \
        if (a.phase && tid == 0) {                                              \
            const long long t_ = clock64();                                     \
            sh.ph[ph_cur] += t_ - ph_last; ph_last = t_; ph_cur = (i_);         \
        }                                                                       \
    } while (0)
    if (tid < 16) { sh.ph[tid] = 0; sh.misc[tid] = 0; }
    if (tid < PMDI_KMAX_I * 8) sh.wk[tid] = 0;
    if (tid < PMDI_KMAX_I) sh.khint[tid] = 0;
    long long ph_t0 = 0, ph_r0 = 0;
    if (a.phase && tid == 0) { ph_last = clock64(); ph_t0 = ph_last; ph_r0 = wall_clock64(); }

    for (int p = tid; p < P; p += T) sh.lw[p] = a.lw_init;
    for (int c = tid; c <= P; c += T) { sh.lead_of[c] = PMDI_INF_I; sh.slot_of[c] = 0; }
    for (int e = tid; e < H; e += T) { sh.h1.key[e] = 0; sh.h1.a[e] = 0; sh.h2.key[e] = 0; sh.h2.a[e] = 0; sh.h2.b[e] = PMDI_INF_I; }
    for (int e = tid; e < a.item_cap; e += T) sh.ktab_minp[e] = PMDI_INF_I;
    for (int e = tid; e < 2 * ((P >> 6) + 1); e += T) { sh.bm_fresh[e] = 0; sh.bm_clone[e] = 0; sh.bm_keep[e] = 0; sh.bm_reuse[e] = 0; }

    // a chain the settled-chain kernel handed over carries on at the observation where that kernel stopped: the bookkeeping of the
    // datasets it left undone is replayed from the recorded draws, then calc_ESS of that observation and everything after it
    long long pos0 = n1 - 1;
    int failed = 0;
    bool lw_uniform = true;     // every particle holds the same log-weight (then ESS == P exactly)
    if (RESUME) {
        __syncthreads();
        sweep_resume_load<T>(ap);
        const long long posr = a.resume[(size_t)chain * 16];
        sweep_replay<T, WPS>(ap, posr, order[posr], a.resume[(size_t)chain * 16 + 1]);      // the whole of that observation
        if (sh.misc[M_FAIL]) failed = 1;
        lw_uniform = sh.misc[M_XAB] != 0;
        __syncthreads();
        if (tid == 0) sh.misc[M_XAB] = 0;
        pos0 = posr + 1;
    } else
    sweep_prefix<T>(ap);
    // ---- the sweep: src/pmdi.jl:209-342 ----
    PH(1);
    int i_next = pos0 < n ? order[pos0] : 0;
    double nx = 0.0;          // register-staged observation row of the upcoming step
    int nxi = 0;
    int ns0_next = s_in[i_next];   // ... and the reference trajectory's label there (dataset 0)
    {
        const DsetDev &d0 = dsb[0];
        if (tid < d0.D) {
            if (d0.kind == K_GAUSSIAN) nx = glob(d0.xf)[(size_t)i_next * d0.D + tid]; else nxi = glob(d0.xi)[(size_t)i_next * d0.D + tid];
        }
    }
    for (long long pos = pos0; pos < n && !failed; ++pos) {
        int xhdr = 0;          // split mode: this step's hand-off is its header alone (bit 0), its one-hot label (bits 8..15)
        const int i = i_next;
        if (pos + 1 < n) i_next = order[pos + 1];
        for (int k = 0; k < K && !failed; ++k) {
            // a fresh copy of the lane's index per step: addresses derived from it are recomputed
            // each step instead of being hoisted out of the sweep loop and held (spilled) across it
            const int tid_outer_ = tid, lane_outer_ = lane;
            {
            int tid = opaque_vgpr(tid_outer_), lane = opaque_vgpr(lane_outer_);
#define FRESH_LANE_IDS() asm volatile("" : "+v"(tid), "+v"(lane))
            const DsetDev &d = dsb[k];
            const KS s = make_ks(d, chain);
            const int D = d.D;
            const int maxid = sh.kmaxid[k];
            const int ncls = sh.kncls[k];
            const int cur = sh.kcur[k];
            const gint part = s.part[cur];
            const auto pidk = dual(a.pid_lds != 0, sh.pid + (size_t)k * P, s.pid);
            const auto colk = dual(a.col_lds != 0, sh.col + (size_t)k * P, s.col);
            const auto sidp = dual(a.pp_lds != 0, sh.sid, s.sid);
            const auto kvp = dual(a.pp_lds != 0, sh.kv, s.kv);
            const unsigned char *flk = gen(sh.fl + (size_t)k * Dp);
            const double *pik = gen(sh.pis + k * N);
            const ClsList cl{sh.cl_lead + k * PMDI_CLS_LDS, sh.cl_val + k * PMDI_CLS_LDS, s.clslead, s.clsval, PMDI_CLS_LDS};
            const int items = ncls * N;
            const bool small = items <= a.item_cap;
            if (ncls != 1) lw_uniform = false;

            PH(1);
            const int ns0_cur = ns0_next;
            // the observation row was fetched into registers during the previous step
            if (tid < D) {
                if (d.kind == K_GAUSSIAN) sh.xs[tid] = nx; else ((int *)sh.xs)[tid] = nxi;
            }
            // slot_of is shared by the K datasets: rebuild it from this dataset's class list
            for (int r = tid; r < ncls; r += T) sh.slot_of[cl.val(r)] = r;
            if (tid == 0) sh.misc[M_OVF] = 0;

            // -- A1: which clusters can a class leader reach?  (the reference evaluates every
            // id 1..max at :218-220, but only these entries are ever read at :232)
            if (small) {
                for (int w = tid; w < items; w += T) {
                    const int r = w / N, nn = w - r * N;
                    const int id = part[(size_t)colk[cl.lead(r)] * N + nn];
                    sh.item_id[w] = id;
                    bool won;
                    const int slot = ht_insert(sh.h1, id, won, H);   // cannot fail: items <= ht_size/2
                    if (won) {
                        const int ps = atomicAdd(gen(&sh.misc[M_NEED]), 1);
                        sh.need[ps] = id; sh.need_slot[ps] = slot; sh.h1.a[slot] = ps;
                    }
                }
            }
            for (int q = T + tid; q < D; q += T) {
                if (d.kind == K_GAUSSIAN) sh.xs[q] = glob(d.xf)[(size_t)i * D + q]; else ((int *)sh.xs)[q] = glob(d.xi)[(size_t)i * D + q];
            }
            if (small) lds_barrier(); else __syncthreads();
            const int nneed = small ? sh.misc[M_NEED] : maxid;
            const int nflag = sh.knflag[k];

            // -- A2/A3: log-predictive of the needed clusters.  Lanes = (cluster, feature) for
            // the per-feature terms, then one lane per cluster adds them in feature order
            // (bit-identical to the sequential loops of calc_logprob).
            if (!small) {
                PH(2); FRESH_LANE_IDS();
                sweep_logprob_all<T>(ap, k, maxid);
            }
            else {
                const int RS = 2 * D + 1, D1 = D + 1;
                int CH = a.terms_cap / RS;
                if (CH < 1) CH = 1;
                for (int j0 = 0; j0 < nneed; j0 += CH) {
                    const int nid = min(CH, nneed - j0);
                    PH(2); FRESH_LANE_IDS();
                    for (int it = tid; it < nid * D1; it += T) {
                        const int il = it / D1, q = it - il * D1;
                        const int id = small ? sh.need[j0 + il] : 1 + j0 + il;
                        const int cn = s.cn[id];
                        if (q == D) {   // the per-cluster prefix: gaussian_cluster.jl:38-40
                            if (d.kind == K_GAUSSIAN) sh.term[il * RS + 2 * D] = (double)nflag * glob(d.gtab)[cn];
                            continue;
                        }
                        if (!flk[q]) continue;
                        double ta = 0.0, tb = 0.0;
                        if (d.kind == K_GAUSSIAN) {
                            gauss_terms(sh.xs[q], (double)cn, gauss_ml(cn, ld2(s.sb, (size_t)id * D + q)), ta, tb);
                        } else if (d.kind == K_CATEGORICAL) {
                            const int x = ((const int *)sh.xs)[q];
                            ta = glob(d.lhtab)[glob(d.maxcol)[q] + 2 * cn];                 // log(nlevels_q + n)
                            const int c = s.cnt[((size_t)id * D + q) * d.L + (x - 1)];
                            tb = (cn == 0) ? glob(d.lhtab)[1] : glob(d.lhtab)[2 * c + 1];   // log(0.5 + counts)
                        } else {
                            const int x = ((const int *)sh.xs)[q];
                            ta = negbin_term(glob(d.lgtab), cn, x, s.nbs[(size_t)id * D + q]);
                        }
                        sh.term[il * RS + 2 * q] = ta;
                        sh.term[il * RS + 2 * q + 1] = tb;
                    }
                    if (small) lds_barrier(); else __syncthreads();
                    PH(3); FRESH_LANE_IDS();
                    for (int il = tid; il < nid; il += T) {
                        const double *t = gen(sh.term + il * RS);
                        double out;
                        if (d.kind == K_GAUSSIAN) {
                            out = t[2 * D];
                            if (nflag == D) {   // all features on: fetch 8 features' terms, then add in order
                                for (int q0 = 0; q0 < D; q0 += 8) {
                                    double ra[8], rb[8];
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        const int q = min(q0 + u, D - 1);
                                        ra[u] = t[2 * q]; rb[u] = t[2 * q + 1];
                                    }
#pragma unroll
                                    for (int u = 0; u < 8; ++u)
                                        if (q0 + u < D) { out += ra[u]; out -= rb[u]; }
                                }
                            } else {
                                for (int q = 0; q < D; ++q)
                                    if (flk[q]) { out += t[2 * q]; out -= t[2 * q + 1]; }
                            }
                        } else if (d.kind == K_CATEGORICAL) {
                            double acc = 0.0;                                  // categorical_cluster.jl:30
                            for (int q = 0; q < D; ++q) if (flk[q]) acc += t[2 * q];
                            out = -acc;
                            for (int q = 0; q < D; ++q) if (flk[q]) out += t[2 * q + 1];
                        } else {
                            out = 0.0;                                         // negbinom_cluster.jl:25
                            for (int q = 0; q < D; ++q) if (flk[q]) out += t[2 * q];
                        }
                        if (small) sh.lpl[j0 + il] = out; else s.lp[1 + j0 + il] = out;
                    }
                    if (small) lds_barrier(); else __syncthreads();
                }
            }

            // -- B: mutation CDF per particle class (:231-248): lanes = (class, label) inside a
            // wave; max / cumsum / normalise by shuffles.  The cumsum follows Julia's
            // accumulate_pairwise!: c[n] = e[0] + (e[1] + ... + e[n]).
            PH(4); FRESH_LANE_IDS();
            const auto cdfp = dual(small, sh.cdf, s.cdf);
            if (MANY && N > 64) {
                // 64 < N <= 255: one class per wave, the labels in up to four chunks of 64 lanes (same arithmetic, same order as Julia's
                // accumulate_pairwise!: c[0] = e[0]; fewer than 128 further elements: c[n] = e[0] + (e[1] + ... + e[n]); else the rest
                // splits once into two leaves (both shorter than 128 up to N = 255) and the right leaf's carry is e[0] + total(left))
                constexpr int NC = 4;
                double *wv = gen(sh.term + wave * 512);      // v[0..255], e[0..255]
                const int nrest = N - 1;
                const int n2 = (nrest >= 128) ? (nrest >> 1) : nrest;       // elements 1 .. n2 form the left (or only) leaf
                for (int r = wave; r < ncls; r += T / 64) {
                    double v2[NC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int nn = lane + 64 * c;
                        if (nn < N) {
                            if (small) v2[c] = sh.lpl[sh.h1.a[ht_find(sh.h1, sh.item_id[r * N + nn])]];
                            else v2[c] = s.lp[part[(size_t)colk[cl.lead(r)] * N + nn]];
                            wv[nn] = v2[c];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double m = wv[0];
                    for (int j = 1; j < N; ++j) { const double t = wv[j]; m = (t > m) ? t : m; }
                    double e2[NC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int nn = lane + 64 * c;
                        if (nn < N) {
                            double e = v2[c] - m;
                            e = exp(e);
                            e = e * pik[nn];
                            e2[c] = e;
                            wv[256 + nn] = e;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const double e0 = wv[256];
                    double carry_r = e0;                       // carry of the right leaf (only when the rest was split)
                    if (n2 < nrest) {
                        double tl = wv[256 + 1];
                        for (int j = 2; j <= n2; ++j) tl = tl + wv[256 + j];
                        carry_r = e0 + tl;
                    }
                    double c2[NC] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int nn = lane + 64 * c;
                        if (nn < N) {
                            const int j0 = (nn <= n2) ? 1 : n2 + 1;            // first element of the leaf that holds nn
                            double s_ = 0.0;
                            for (int j = j0; j <= nn; ++j) { const double t = wv[256 + j]; s_ = (j == j0) ? t : s_ + t; }
                            c2[c] = (nn == 0) ? e2[c] : ((nn <= n2) ? e0 + s_ : carry_r + s_);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int c = 0; c < NC; ++c) { const int nn = lane + 64 * c; if (nn < N) wv[nn] = c2[c]; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const double fN = wv[N - 1];
                    unsigned long long mo[NC], mt[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int nn = lane + 64 * c;
                        const double cd = c2[c] / fN;
                        mo[c] = __ballot(nn < N && (cd == 1.0 || nn == N - 1));
                        mt[c] = __ballot(nn < N && cd < 0x1p-53);
                        if (nn < N) cdfp[(size_t)r * (N + 2) + nn] = cd;
                    }
                    if (lane == 0) {
                        cdfp[(size_t)r * (N + 2) + N] = log(fN) + m;
                        // the first label whose CDF is 1 (chunk cs, bit ns1) and whether every label before it is negligible
                        int cs = NC - 1;
                        unsigned long long mo_s = mo[NC - 1], mt_s = mt[NC - 1];
                        bool full_before = true;             // every label of the chunks before cs is negligible
#pragma unroll
                        for (int c = NC - 2; c >= 0; --c) if (mo[c] != 0) { cs = c; mo_s = mo[c]; mt_s = mt[c]; }
#pragma unroll
                        for (int c = 0; c < NC - 1; ++c) if (c < cs) full_before = full_before && mt[c] == ~0ull;
                        const int ns1 = __ffsll((long long)mo_s) - 1;
                        const unsigned long long below = (ns1 == 0) ? 0ull : ((1ull << ns1) - 1ull);
                        const bool onehot = full_before && (mt_s & below) == below;
                        cdfp[(size_t)r * (N + 2) + N + 1] = onehot ? (double)(64 * cs + ns1) : -1.0;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                const int G = 64 / N;
                const int g = lane / N, nn = lane - g * N;
                const int gbase = (g < G) ? g * N : 0;
                double *wv = gen(sh.term + wave * 128);      // this wave's exchange area (terms are dead here)
                for (int r0 = 0; r0 < ncls; r0 += (T / 64) * G) {
                    if (r0 + wave * G >= ncls) break;          // wave-uniform: nothing left for this wave
                    const int r = r0 + wave * G + g;
                    const bool valid = (g < G) && (r < ncls);
                    double v = 0.0;
                    if (valid) {
                        if (small) v = sh.lpl[sh.h1.a[ht_find(sh.h1, sh.item_id[r * N + nn])]];
                        else v = s.lp[part[(size_t)colk[cl.lead(r)] * N + nn]];
                    }
                    wv[lane] = v;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double m = v;
                    {
                        int j = 0;
                        for (; j + 4 <= N; j += 4) {      // four LDS reads in flight
                            const double t0 = wv[gbase + j], t1 = wv[gbase + j + 1], t2 = wv[gbase + j + 2], t3 = wv[gbase + j + 3];
                            m = (t0 > m) ? t0 : m; m = (t1 > m) ? t1 : m; m = (t2 > m) ? t2 : m; m = (t3 > m) ? t3 : m;
                        }
                        for (; j < N; ++j) { const double t = wv[gbase + j]; m = (t > m) ? t : m; }
                    }
                    double e = v - m;
                    e = exp(e);
                    e = e * pik[nn];
                    wv[64 + lane] = e;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const double e0 = wv[64 + gbase];
                    double s_ = 0.0;
                    {
                        int j = 1;
                        for (; j + 4 <= N; j += 4) {      // loads first, then the ordered adds
                            const double t0 = wv[64 + gbase + j], t1 = wv[64 + gbase + j + 1],
                                         t2 = wv[64 + gbase + j + 2], t3 = wv[64 + gbase + j + 3];
                            if (j <= nn) s_ = (j == 1) ? t0 : s_ + t0;
                            if (j + 1 <= nn) s_ = s_ + t1;
                            if (j + 2 <= nn) s_ = s_ + t2;
                            if (j + 3 <= nn) s_ = s_ + t3;
                        }
                        for (; j < N; ++j) { const double t = wv[64 + gbase + j]; if (j <= nn) s_ = (j == 1) ? t : s_ + t; }
                    }
                    const double c = (nn == 0) ? e : e0 + s_;
                    __builtin_amdgcn_wave_barrier();
                    wv[lane] = c;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const double fN = wv[gbase + N - 1];
                    const double cd = c / fN;
                    // one-hot to working precision?  every uniform is an odd multiple of 2^-53, so a
                    // label whose CDF is < 2^-53 is never chosen and one whose CDF is 1.0 always
                    // stops the search: the draw (:252-260) is then the same label for every u
                    const unsigned long long m_one = __ballot(valid && (cd == 1.0 || nn == N - 1));
                    const unsigned long long m_tiny = __ballot(valid && cd < 0x1p-53);
                    if (valid) {
                        cdfp[(size_t)r * (N + 2) + nn] = cd;
                        if (nn == N - 1) {
                            cdfp[(size_t)r * (N + 2) + N] = log(fN) + m;
                            const unsigned long long grp = (N == 64) ? ~0ull : (((1ull << N) - 1ull) << gbase);
                            const int nstar = __ffsll((long long)((m_one & grp) >> gbase)) - 1;
                            const unsigned long long below = (nstar == 0) ? 0ull : ((1ull << nstar) - 1ull);
                            const bool onehot = (((m_tiny & grp) >> gbase) & below) == below;
                            cdfp[(size_t)r * (N + 2) + N + 1] = onehot ? (double)nstar : -1.0;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (small) lds_barrier(); else __syncthreads();

            // -- C..F: allocation draw (:251-265), class ids (:266-272), copy-on-write update of the
            // chosen clusters (:275-310).  Fast path (the step's tables fit LDS): classes come from a
            // (class, label) key table, ranks "in particle order" from LDS bitmaps + popcounts, no
            // block-wide scans.  Fallback (burn-in): per-particle keys, ballot scans, global tables.
            PH(5); FRESH_LANE_IDS();
            {   // prefetch the next step's observation row (dataset k+1 of this observation, or
                // dataset 0 of the next one) into registers; it is consumed a whole step later
                int kn = k + 1, in_ = i;
                bool have = true;
                if (kn == K) { kn = 0; in_ = i_next; have = pos + 1 < n; }
                const DsetDev &dn = dsb[kn];
                if (have && tid < dn.D) {
                    if (dn.kind == K_GAUSSIAN) nx = glob(dn.xf)[(size_t)in_ * dn.D + tid]; else nxi = glob(dn.xi)[(size_t)in_ * dn.D + tid];
                }
                if (have) ns0_next = s_in[(size_t)kn * n + in_];
            }
            if (small) {
                for (int j = tid; j < nneed; j += T) { const int sl = sh.need_slot[j]; sh.h1.key[sl] = 0; sh.h1.a[sl] = 0; }
                if (tid == 0) sh.misc[M_NEED] = 0;
            }
            bool fast = small && sh.khint[k] == 0;   // hint: the last steps chose too many distinct clusters for the LDS census
            bool converted = false;
            int nd = 0, nclone = 0, new_ncls = 0;
            // -- C1: allocation draw (:251-265).  While drawing, the particles vote on whether the step is
            // unanimous: one class, every particle draws the reference trajectory's label and holds the
            // same cluster under it.  Then :266-310 has one outcome for all particles (one key, one
            // chosen cluster with ncopies = P, cloned iff counts != P): no census, no ranks.
            bool ustep = false;
            if (fast) {
                const int ns0 = ns0_cur;                                     // reference trajectory (:262), fetched a step ago
                const int c0 = part[(size_t)colk[0] * N + ns0];
                int same = (ncls == 1) ? 1 : 0;
                const bool one = ncls == 1;                                  // every particle reads CDF row 0
                // split mode, one class with a one-hot CDF row: every particle gets the same increment, particle 0 the reference
                // label and all others the one-hot label -- the hand-off is a 16-byte header, no per-particle records
                const bool hdr_only = XSPLIT && one && (int)sh.cdf[N + 1] >= 0;
                if (hdr_only) xhdr = 1 | ((int)sh.cdf[N + 1] << 8);
                for (int pb0 = 0; pb0 < P; pb0 += 4 * T) {       // four particles per lane, stage by stage: their
                    int ns_[4], c_[4], r_[4];                    // LDS chains and pool reads overlap
                    double inc_[4], lw_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = pb0 + u * T + tid;
                        r_[u] = (!one && p < P) ? sh.slot_of[pidk[p]] : 0;
                        c_[u] = (p < P) ? (int)colk[p] : 0;      // the particle's column, then its entry under the drawn label
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = pb0 + u * T + tid;
                        const double *row = gen(sh.cdf + (size_t)r_[u] * (N + 2));
                        const int hot = (int)row[N + 1];
                        inc_[u] = row[N];
                        lw_[u] = (p < P) ? sh.lw[p] : 0.0;
                        int ns = 0;
                        if (p == 0) {
                            ns = ns0;
                        } else if (hot >= 0) {
                            ns = hot;                                // one-hot CDF: no random number needed
                        } else if (p < P) {
                            const double u01 = uniform01(seed, iter, (unsigned)pos, (unsigned)(kd0 + k), (unsigned)p, SITE_DRAW);
                            // first label whose CDF exceeds u (:252-260); the CDF is non-decreasing, so
                            // that is the number of leading entries that do not exceed u
                            int t = 0;
                            for (; t + 4 <= N - 1; t += 4) {          // four LDS reads in flight
                                const double a0 = row[t], a1 = row[t + 1], a2 = row[t + 2], a3 = row[t + 3];
                                ns += ((a0 > u01) ? 0 : 1) + ((a1 > u01) ? 0 : 1) + ((a2 > u01) ? 0 : 1) + ((a3 > u01) ? 0 : 1);
                            }
                            for (; t < N - 1; ++t) ns += (row[t] > u01) ? 0 : 1;
                        }
                        ns_[u] = ns;
                        c_[u] = (p < P) ? part[(size_t)c_[u] * N + ns] : 0;     // sstar_id (:264)
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = pb0 + u * T + tid;
                        if (p < P) {
                            if (!XSPLIT) {
                                sh.lw[p] = lw_[u] + inc_[u];
                            } else if (!hdr_only) {        // the increments of the K datasets are added in dataset order after the hand-off
                                const size_t xo = XOFF((pos - (n1 - 1)) & 1, kd0) + p;
                                __hip_atomic_store((unsigned long long *)a.xinc + xo, (unsigned long long)__double_as_longlong(inc_[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(a.xlab + xo, ns_[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            sh.news[k * P + p] = (unsigned char)ns_[u];
                            s.sstar[(size_t)pos * P + p] = (unsigned char)ns_[u];   // (:265)
                            sidp[p] = r_[u] * N + ns_[u];
                            kvp[p] = c_[u];
                            same &= (ns_[u] == ns0 && c_[u] == c0) ? 1 : 0;
                        }
                    }
                }
                // what a unanimous step needs from the pool, fetched before the vote's barrier
                const int key = (cl.val(0) - 1) * N + ns0;
                int v = (a.q1 == 1 || ncls != 1) ? 0 : s.newid[key];         // (:266)
                const bool freshk = v <= 0;
                if (freshk) v = 1;                                           // curr_id += 1 (:267-269)
                const bool needs = s.counts[c0] != P;                        // ncopies == counts ? (:286)
                const int nnew = s.cn[c0] + 1;
                ustep = __syncthreads_and(same) != 0;
                if (ustep) {
                    PH(13);
                    const int tgt = needs ? maxid + 1 : c0;                  // (:290-292)
                    nclone = needs ? 1 : 0;
                    new_ncls = 1;
                    if (maxid + nclone > cap) { failed = 1; break; }
                    for (int p = tid; p < P; p += T) pidk[p] = v;
                    if (needs) {                                             // (:301-308): every live column holds c0 under ns0
                        const int ncol = sh.kncol[k];
                        for (int c = tid; c < ncol; c += T) part[(size_t)c * N + ns0] = tgt;
                    }
                    if (tid == 0) {
                        if (freshk && a.q1 == 0) s.newid[key] = v;
                        if (needs) { s.counts[c0] -= P; s.counts[tgt] = P; } // (:293-294)
                        s.cn[tgt] = nnew;
                        cl.set(0, 0, v);
                    }
                    stats_update_all<T, (T >= 512 && WPS <= 2) ? PMDI_VH_U : 4>(d, s, flk, gen(sh.xs), 1, D, tid, [&](int, int &src, int &dst, int &nn) {
                        src = c0; dst = tgt; nn = nnew;
                    });
                }
            }
            if (!ustep) {
            if (fast) {
                // -- C2: census of the chosen clusters and of the touched (class, label) keys
                for (int pb = 0; pb < P; pb += T) {
                    const int p = pb + tid;
                    const bool valid = p < P;
                    int c = 0, kidx = 0;
                    if (valid) { c = kvp[p]; kidx = sidp[p]; }
                    const unsigned long long vmask = __ballot(valid);
                    const int k0 = __builtin_amdgcn_readlane(kidx, 0), c0w = __builtin_amdgcn_readlane(c, 0);
                    int slot = -1;
                    if (__all(!valid || (kidx == k0 && c == c0w))) {     // the whole wave agrees: one lane speaks
                        if (lane == 0 && valid) {
                            atomicMin(gen(&sh.ktab_minp[k0]), p);
                            bool won;
                            slot = ht_insert(sh.h2, c0w, won, 48);
                            if (slot < 0) sh.misc[M_OVF] = 1;
                            else { atomicAdd(gen(&sh.h2.a[slot]), __popcll(vmask)); atomicMin(gen(&sh.h2.b[slot]), p); }
                        }
                        slot = __builtin_amdgcn_readlane(slot, 0);
                    } else {
                        if (valid) atomicMin(gen(&sh.ktab_minp[kidx]), p);
                        int cnt;
                        const int lead = wave_group_lead(c, valid, cnt);
                        if (valid && lead == lane) {
                            bool won;
                            slot = ht_insert(sh.h2, c, won, 48);
                            if (slot < 0) sh.misc[M_OVF] = 1;
                            else { atomicAdd(gen(&sh.h2.a[slot]), cnt); atomicMin(gen(&sh.h2.b[slot]), p); }
                        }
                        slot = __shfl(slot, lead);
                    }
                    if (valid) kvp[p] = slot;
                }
                lds_barrier();
                if (sh.misc[M_OVF]) {
                    // too many distinct chosen clusters for the LDS census: hand this step to the
                    // fallback (per-particle keys, per-id tables in global memory)
                    fast = false;
                    converted = true;
                    __syncthreads();
                }
            }
            if (fast) {
                // -- D'1: the touched (class, label) keys and the first particle of every chosen cluster
                PH(6); FRESH_LANE_IDS();
                for (int w = tid; w < items; w += T) {
                    const int mp = sh.ktab_minp[w];
                    if (mp != PMDI_INF_I) {
                        const int r = w / N, ns = w - r * N;
                        const int key = (cl.val(r) - 1) * N + ns;
                        const int v = (a.q1 == 1) ? 0 : s.newid[key];          // (:266)
                        const int j = atomicAdd(gen(&sh.misc[M_NK]), 1);
                        sh.klist[j] = w; sh.kl_v[j] = v; sh.kl_key[j] = key;
                        if (v <= 0) atomicOr(gen(&sh.bm_fresh[mp >> 5]), 1u << (mp & 31));
                    }
                }
                for (int pb = 0; pb < P; pb += T) {
                    const int p = pb + tid;
                    if (p < P) {
                        const int slot = kvp[p];
                        if (sh.h2.b[slot] == p) {
                            const int c = sh.h2.key[slot];
                            const bool needs = sh.h2.a[slot] != s.counts[c];       // ncopies == counts ? (:286)
                            const int j = atomicAdd(gen(&sh.misc[M_NF]), 1);
                            sh.fl_p[j] = needs ? (p | 0x40000000) : p;
                            sh.fl_slot[j] = slot;
                            sh.fl_nnew[j] = s.cn[c] + 1;
                            if (needs) atomicOr(gen(&sh.bm_clone[p >> 5]), 1u << (p & 31));
                        }
                    }
                }
                lds_barrier();
                // -- D'2: ranks in particle order by popcounts below the particle's bit
                PH(7); FRESH_LANE_IDS();
                const int nk = sh.misc[M_NK], nf = sh.misc[M_NF];
                if (wave == 0) {
                    for (int j0 = 0; j0 < nk; j0 += 64) {
                        const int j = j0 + lane;
                        if (j < nk) {
                            const int w = sh.klist[j];
                            int v = sh.kl_v[j];
                            if (v <= 0) {                                           // curr_id += 1 (:267-269)
                                v = 1 + popc_below(gen(sh.bm_fresh), sh.ktab_minp[w]);
                                if (a.q1 == 0) s.newid[sh.kl_key[j]] = v;
                            }
                            sh.kl_v[j] = v;
                            sh.ktab_val[w] = v;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // classes of the next step: one per distinct value, leader = lowest first particle
                    for (int j0 = 0; j0 < nk; j0 += 64) {
                        const int j = j0 + lane;
                        if (j < nk) {
                            const int v = sh.kl_v[j], mp = sh.ktab_minp[sh.klist[j]];
                            int rep = 1;
                            for (int j2 = 0; j2 < nk; ++j2)
                                if (sh.kl_v[j2] == v && sh.ktab_minp[sh.klist[j2]] < mp) rep = 0;
                            sh.kl_key[j] = rep;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    int nrep = 0;
                    for (int j0 = 0; j0 < nk; j0 += 64) {
                        const int j = j0 + lane;
                        const bool rep = (j < nk) && sh.kl_key[j];
                        if (rep) {
                            const int mp = sh.ktab_minp[sh.klist[j]];
                            int slot = 0;
                            for (int j2 = 0; j2 < nk; ++j2)
                                if (sh.kl_key[j2] && sh.ktab_minp[sh.klist[j2]] < mp) ++slot;
                            cl.set(slot, mp, sh.kl_v[j]);
                        }
                        nrep += __popcll(__ballot(rep));
                    }
                    if (lane == 0) { sh.misc[M_NCLS] = nrep; sh.misc[M_NCLONE] = popc_below(gen(sh.bm_clone), P); }
                }
                for (int j = tid; j < nf; j += T) {
                    const int pp = sh.fl_p[j];
                    const bool needs = (pp & 0x40000000) != 0;
                    const int p = pp & 0x3fffffff;
                    const int slot = sh.fl_slot[j];
                    const int c = sh.h2.key[slot], ncp = sh.h2.a[slot];
                    const int tgt = needs ? maxid + 1 + popc_below(gen(sh.bm_clone), p) : c;    // (:290-292)
                    sh.fl_p[j] = c;
                    sh.fl_tgt[j] = tgt;
                    if (tgt <= cap) {
                        if (needs) { s.counts[c] -= ncp; s.counts[tgt] = ncp; }             // (:293-294)
                        s.cn[tgt] = sh.fl_nnew[j];
                        sh.h2.a[slot] = tgt;                                                // chosen id -> updated id
                    }
                }
                lds_barrier();
                nd = nf;
                nclone = sh.misc[M_NCLONE];
                new_ncls = sh.misc[M_NCLS];
                if (maxid + nclone > cap) { failed = 1; break; }
                // -- E': apply (:301-308), sufficient statistics (:297,:300), table clean-up
                PH(8); FRESH_LANE_IDS();
                columns_apply<T>(sh, s, part, colk, k, N, P, tid, (unsigned)(pos - (n1 - 1)) + 1u, [&](int p, int &c, int &tgt) {
                    const int slot = kvp[p];
                    c = sh.h2.key[slot]; tgt = sh.h2.a[slot];
                });
                for (int p = tid; p < P; p += T) pidk[p] = sh.ktab_val[sidp[p]];
                stats_update_all<T, (T >= 512 && WPS <= 2) ? PMDI_VH_U : 4>(d, s, flk, gen(sh.xs), nd, D, tid, [&](int j, int &src, int &dst, int &nnew) {
                    src = sh.fl_p[j]; dst = sh.fl_tgt[j]; nnew = sh.fl_nnew[j];
                });
                for (int j = tid; j < nk; j += T) sh.ktab_minp[sh.klist[j]] = PMDI_INF_I;
                for (int w = tid; w < 2 * ((P >> 6) + 1); w += T) { sh.bm_fresh[w] = 0; sh.bm_clone[w] = 0; }
                lds_barrier();
                for (int j = tid; j < nf; j += T) { const int sl = sh.fl_slot[j]; sh.h2.key[sl] = 0; sh.h2.a[sl] = 0; sh.h2.b[sl] = PMDI_INF_I; }
                if (tid == 0) { sh.misc[M_NK] = 0; sh.misc[M_NF] = 0; }
            } else {
                sweep_slow<T, WPS>(ap, k, i, pos, small, converted, maxid, ncls, ph_last, ph_cur);
                if (sh.misc[M_FAIL]) { failed = 1; break; }
                nclone = sh.misc[M_NCLONE];
                new_ncls = sh.misc[M_NCLS];
                nd = sh.misc[M_ND];
            }
            }
            if (tid == 0) {
                sh.stat[fast ? 5 : (converted ? 6 : 7)] += 1;
                sh.stat[0] += maxid;                  // src/__pmdi.jl:187
                sh.stat[4] += ncls;
                sh.stat[2] += nclone;
                if (maxid + nclone > sh.stat[3]) sh.stat[3] = maxid + nclone;
                sh.kmaxid[k] = maxid + nclone; sh.kncls[k] = new_ncls;
                sh.wk[k * 8 + WK_EVAL] += nneed; sh.wk[k * 8 + WK_UPD] += ustep ? 1 : nd; sh.wk[k * 8 + WK_CLONE] += nclone;
            }
            __syncthreads();
#undef FRESH_LANE_IDS
            }
        }
        if (failed) break;

        // -- Phi_upweight! (src/misc.jl:50-59)
        PH(9);
        if (XSPLIT) {
            // Hand-off between the K workgroups of the chain (one per swept observation): every workgroup has stored its dataset's
            // records with agent-scope (sc1, write-through) stores; each storing wave drains them, the workgroup meets, one lane
            // adds to the chain's arrival counter and polls it with sc1 loads; after the workgroup barrier every lane reads the
            // K records of its particles with sc1 loads (they bypass the CU's L1, which another CU's stores never refresh).
            // MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": placement-independent.
            if (tid == 0) {        // header: {header-only flag | one-hot label << 8 | reference label << 16, increment of class slot 0}
                unsigned long long *hd = (unsigned long long *)a.xhdr + ((((size_t)chain * 2 + (size_t)((pos - (n1 - 1)) & 1)) * (size_t)a.K + (size_t)kd0) * 2);
                const unsigned long long w0 = (unsigned long long)(unsigned)(xhdr | ((int)sh.news[0] << 16));
                __hip_atomic_store(hd, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(hd + 1, (unsigned long long)__double_as_longlong((double)sh.cdf[N]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(gen(XCNT), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int target = Kf * (int)(pos - (n1 - 1) + 1);
                const long long w0 = wall_clock64();
                int ab = 0;
                for (;;) {
                    const int v = __hip_atomic_load(gen(XCNT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v >= (1 << 28)) { ab = 2; break; }                    // a partner failed (pool capacity): stop with it
                    if (v >= target) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - w0 > 2000000000LL) { ab = 3; break; } // 20 s at 100 MHz: a partner never arrived
                }
                sh.misc[M_XAB] = ab;
            }
            __syncthreads();
            if (sh.misc[M_XAB]) { failed = sh.misc[M_XAB]; break; }
            const unsigned long long *xi = (const unsigned long long *)a.xinc + XOFF((pos - (n1 - 1)) & 1, 0);
            const int *xl = a.xlab + XOFF((pos - (n1 - 1)) & 1, 0);
            const unsigned long long *xh = (const unsigned long long *)a.xhdr + (((size_t)chain * 2 + (size_t)((pos - (n1 - 1)) & 1)) * (size_t)a.K) * 2;
            PH(12);
            // the K headers first, all loads in flight together (they are the same addresses for every lane)
            unsigned hw[PMDI_KMAX_I];
            unsigned long long hinc[PMDI_KMAX_I];
#pragma unroll
            for (int kk = 0; kk < PMDI_KMAX_I; ++kk) {
                hw[kk] = 0; hinc[kk] = 0;
                if (kk < Kf) {
                    hw[kk] = (unsigned)__hip_atomic_load(xh + 2 * kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hinc[kk] = __hip_atomic_load(xh + 2 * kk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            for (int p = tid; p < P; p += T) {
                unsigned long long labs = 0;                                   // the K chosen labels, a byte each (N <= 64)
                double w = sh.lw[p];
                unsigned long long rinc[PMDI_KMAX_I];
                int rlab[PMDI_KMAX_I];
#pragma unroll
                for (int kk = 0; kk < PMDI_KMAX_I; ++kk) {                     // per-particle records of the datasets that sent them
                    rinc[kk] = hinc[kk];
                    rlab[kk] = (p == 0) ? (int)((hw[kk] >> 16) & 0xffu) : (int)((hw[kk] >> 8) & 0xffu);
                    if (kk < Kf && !(hw[kk] & 1u)) {
                        rinc[kk] = __hip_atomic_load(xi + (size_t)kk * P + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        rlab[kk] = __hip_atomic_load(xl + (size_t)kk * P + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < PMDI_KMAX_I; ++kk) {                     // logweight[p] += increment, in dataset order (:227,:245)
                    if (kk < Kf) {
                        labs |= (unsigned long long)(rlab[kk] & 0xff) << (8 * kk);
                        w = w + __longlong_as_double((long long)rinc[kk]);
                    }
                }
                int pr = 0;
                for (int k1 = 0; k1 < Kf - 1; ++k1)
                    for (int k2 = k1 + 1; k2 < Kf; ++k2) {
                        w += (((labs >> (8 * k1)) & 0xff) == ((labs >> (8 * k2)) & 0xff)) ? logphi[pr] : 0.0;
                        ++pr;
                    }
                sh.lw[p] = w;
            }
            lw_uniform = false;
            __syncthreads();
        } else if (K > 1) {
            for (int p = tid; p < P; p += T) {
                int pr = 0;
                double w = sh.lw[p];
                for (int k1 = 0; k1 < K - 1; ++k1)
                    for (int k2 = k1 + 1; k2 < K; ++k2) {
                        w += (sh.news[k1 * P + p] == sh.news[k2 * P + p]) ? logphi[pr] : 0.0;
                        ++pr;
                    }
                sh.lw[p] = w;
            }
            lw_uniform = false;
            __syncthreads();
        }

        // -- calc_ESS (src/misc.jl:15-25).  If every log-weight is the same number the sums are
        // exact (P ones): ESS == P, no resampling; skip the exps.
        // (once the log-weights differ they stay different until a resampling resets them)
        double ess = (double)P;
        bool resample = false;
        double mx = 0.0;
        if (!lw_uniform) {
            mx = -INFINITY;
            for (int p = tid; p < P; p += T) { const double v = sh.lw[p]; mx = (v > mx) ? v : mx; }
            mx = block_max<T>(mx, gen(sh.red));
            double sa = 0.0, sb2 = 0.0;
            for (int p = tid; p < P; p += T) { const double w = exp(sh.lw[p] - mx); sa += w; sb2 += w * w; }
            block_sum2<T>(sa, sb2, gen(sh.red));
            ess = (sa * sa) / sb2;
            // The tree-ordered sums agree with calc_ESS's sequential loop (src/misc.jl:19-23) to ~1e-13 relative; the decision
            // below is a comparison, so when ESS lands that close to P/2 -- k equal weights and the rest negligible give exactly k in
            // the reference's order, and k = P/2 does happen -- the sums are redone in the reference's order by one lane.
            if (fabs(ess - 0.5 * (double)P) <= 1e-9 * (double)P) {
                __syncthreads();
                if (tid == 0) {
                    double na = 0.0, nb = 0.0;
                    for (int p = 0; p < P; ++p) { const double w = exp(sh.lw[p] - mx); na += w; nb += w * w; }
                    sh.red[40] = (na * na) / nb;
                }
                __syncthreads();
                ess = sh.red[40];
            }
            resample = ess <= 0.5 * (double)P;            // src/pmdi.jl:317
        }

        if (resample) {
            PH(10);
            if (tid == 0) sh.stat[1] += 1;
            sweep_resample<T>(ap, pos, mx);
            lw_uniform = true;
        }

        if (a.trace_on && tid == 0) {
            double *tr = a.trace + ((size_t)chain * (n - n1 + 1) + (pos - (n1 - 1))) * (2 + 2 * Kf);
            if (kd0 == 0) { tr[0] = ess; tr[1] = resample ? 1.0 : 0.0; }
            for (int k = 0; k < K; ++k) { tr[2 + kd0 + k] = (double)sh.kmaxid[k]; tr[2 + Kf + kd0 + k] = (double)sh.kncls[k]; }
        }
    }

    if (failed) {
        if (tid == 0) {
            if (failed == 1) a.err[chain] = -4;                                     // PMDI_E_POOL
            else if (failed == 3) a.err[chain] = -6;                                // a partner workgroup never arrived
            else if (failed == 2 && a.err[chain] == 0) a.err[chain] = -4;           // stopped with a partner that ran out of pool
            a.cost[chain] = clock64() - t_start;
            if (XSPLIT) __hip_atomic_fetch_add(gen(XCNT), 1 << 28, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // partners stop waiting
        }
        // the chain keeps its allocations: a caller that goes on (device-resident chains) never reads an unwritten s_out
        for (int k = 0; k < K; ++k)
            for (long long pp = tid; pp < n; pp += T) a.s_out[((size_t)chain * Kf + kd0 + k) * n + pp] = s_in[(size_t)k * n + pp];
        return;
    }

    PH(11);
    sweep_final<T>(ap);
    PH(12);
    if (tid == 0 && kd0 == 0) {
        a.cost[chain] = (RESUME ? a.cost[chain] : 0) + (clock64() - t_start);
        if (a.swept_by) a.swept_by[chain] = (RESUME || a.requeue_only) ? 2 : 0;       // (2: the settled-chain kernel had the chain first)
    }
    if (a.phase && tid == 0) {
        sh.ph[14] = clock64() - ph_t0; sh.ph[15] = wall_clock64() - ph_r0;
    }
    __syncthreads();
    if (a.phase && tid < 16 && kd0 == 0) a.phase[(size_t)chain * 16 + tid] = sh.ph[tid];
#ifdef PMDI_RESAMPLE_TIMERS
    __syncthreads();
    if (a.phase && tid < 10) a.phase[(size_t)chain * 16 + tid] = sh.stat[8 + tid];   // A/B build: resampling sub-phases instead
#endif
}

}  // namespace
