// pmdi_api.cpp -- host side of the C ABI declared in include/pmdi_hip.h.
// Converts the reference's conventions (column-major, 1-based Int64) at the
// boundary, owns device memory, builds the exact-arithmetic lookup tables and
// launches the kernels of pmdi_kernels.hip.  No CPU fallback exists: every
// compute entry point runs on the gfx950 device or fails.
#include "pmdi_internal.h"
#include "../../include/pmdi_hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess)                                                         \
            return fail(e__ == hipErrorOutOfMemory ? PMDI_E_MEMORY : PMDI_E_DEVICE,    \
                        "%s: %s", #expr, hipGetErrorString(e__));                      \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the device generator (pmdi_device.h uniform01) on the host: Philox4x32-10, ctr = (p, pos, site<<16|k, iter)
double host_uniform01(unsigned long long seed, unsigned iter, unsigned pos, unsigned k, unsigned p, unsigned site)
{
    unsigned c0 = p, c1 = pos, c2 = (site << 16) | k, c3 = iter;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const unsigned long long m0 = (unsigned long long)0xD2511F53u * c0, m1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned hi0 = (unsigned)(m0 >> 32), lo0 = (unsigned)m0, hi1 = (unsigned)(m1 >> 32), lo1 = (unsigned)m1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned long long m = ((unsigned long long)(c0 >> 6) << 26) | (unsigned long long)(c1 >> 6);
    return (double)(2 * m + 1) * (1.0 / 9007199254740992.0);
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t b)
    {
        if (b <= bytes && p) return 0;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (b == 0) b = 16;
        hipError_t e = hipMalloc(&p, b);
        if (e != hipSuccess) { p = nullptr; return fail(PMDI_E_MEMORY, "hipMalloc(%zu): %s", b, hipGetErrorString(e)); }
        bytes = b;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// fills the arena offsets of one dataset; returns the per-chain stride
size_t layout_arena(DsetDev &d, int N, int P, long long cap, long long n_rows_sstar, bool sweep_state)
{
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t ids = (size_t)cap + 1;
    if (sweep_state) {
        d.o_particle[0] = take((size_t)N * P * 4);
        d.o_particle[1] = take((size_t)N * P * 4);
        d.o_col = take((size_t)P * 4);
        d.o_cgrp = take((size_t)(N > 3 ? N : 3) * P * 8);      // (the settled-chain kernel keeps 20 bytes per column there: N = 2 would not hold them)
        d.o_pid = take((size_t)P * 4);
        d.o_sid = take((size_t)P * 4);
        d.o_kv = take((size_t)P * 4);
        d.o_newid = take((size_t)N * P * 4);
        d.o_counts = take(ids * 4);
        d.o_ncop = take(ids * 4);
        d.o_firstc = take(ids * 4);
        d.o_lp = take(ids * 8);
        d.o_sstar = take((size_t)n_rows_sstar * P);
        d.o_clslead = take((size_t)P * 4);
        d.o_clsval = take((size_t)P * 4);
        d.o_cdf = take((size_t)P * (N + 2) * 8);
        d.o_dl = take((size_t)3 * P * 4);
        d.o_s2x = take((size_t)2048 * 12);
    }
    d.o_cn = take(ids * 4);
    if (d.kind == K_GAUSSIAN) {
        if (!sweep_state) d.o_ml = take(ids * d.D * 16);   // stand-alone batches keep (mu, lambda) as stored values
        d.o_sb = take(ids * d.D * 16);
    } else if (d.kind == K_CATEGORICAL) {
        d.o_cnt = take(ids * d.D * d.L * 4);
    } else {
        d.o_nbs = take(ids * d.D * 8);
    }
    return o;
}

}  // namespace

// error reporting for the other translation units of the library (pmdi_csv.cpp, pmdi_comm.cpp)
int pmdi_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct pmdi_handle {
    pmdi_config cfg{};
    pmdi_tuning tun{};           // the creator's knobs (a copy: cfg.tuning is not kept), -1 = automatic
    int T = 0;
    long long cap = 0;
    int Dmax = 0, sumD = 0, npairs = 1;
    int terms_cap = 0, pid_lds = 0, pp_lds = 0, col_lds = 0, two_per_cu = 0;
    // light group (block_threads == 0 only): chains whose last sweep met few live clusters per step are
    // swept by 256-thread workgroups on a second stream, concurrently with the wide workgroups of the rest
    bool split = false;
    int l_terms_cap = 0, l_pid_lds = 0, l_pp_lds = 0, l_col_lds = 0;
    int r_terms_cap = 0, r_pid_lds = 0, r_pp_lds = 0, r_col_lds = 0;     // the general kernel's LDS layout at the settled-chain kernel's workgroup width (hand-over in place)
    long long light_ids = 0;
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    unsigned *start_sig = nullptr;   // signal memory: workgroups of the heaviest-chains launch that have started (hipStreamWaitValue32)
    int very_heavy = 0;          // the first `very_heavy` heavy chains of the launch order get a CU each (256-register build)
    bool phase_on = false;
    hipStream_t stream = nullptr;
    DsetDev ds[PMDI_KMAX_I]{};
    std::vector<void *> owned;          // device allocations freed in destroy
    // per-call staging (device)
    DevBuf d_s_in, d_order, d_Pi, d_logphi, d_flags, d_s_out, d_lw, d_pstar, d_stats, d_err, d_trace;
    DevBuf d_swept_by, d_resume;
    bool s2_continue = true;     // a chain the settled-chain kernel gives back is carried on by the general kernel at that observation
                                 // (false: swept again from the start -- the round-3 behaviour, kept for A/B runs)
    DevBuf d_usc, d_partstar, d_kstate, d_phase, d_args, d_args2, d_args3, d_args4, d_args5, d_requeue, d_requeue_total, d_handed, d_group, d_cost, d_lorder, d_ticket, d_work, d_anclog, d_evpos, d_xcnt, d_xinc, d_xlab, d_xhdr;
    int ksplit = 0;
    int ksplit_batch = 0;        // split mode: chain slots per launch when n_chains * K workgroups are not resident at once (0 = one launch)
    bool have_order = false;
    // argument blocks travel through a ring of pinned host slots: the stream-ordered copy is then asynchronous for the host too
    static constexpr int RING = 64;
    SweepArgs *ring = nullptr;
    hipEvent_t ring_ev[RING] = {};
    bool ring_used[RING] = {};
    int ring_head = 0;
    // settled-chain kernel (pmdi_sweep2.hip): takes the light group of a sweep when the configuration is one it is built for
    bool s2_ok = false;
    S2Layout s2{};
    int sweep_no = 0;
    int sticky = 3;              // sweeps a chain stays with the general kernel after the settled-chain kernel gave it back (PMDI_STICKY)
    int err_keep = 0;            // set by the device-resident driver around its sweeps (pmdi_gibbs_step)
    int children = 0;            // live pmdi_gibbs / cluster-batch objects: pmdi_destroy refuses while > 0
    // feature selection
    DevBuf d_traj, d_lm, d_firstpos, d_fnull, d_fflags, d_fprob;
    bool swept = false;
    long long last_n1 = 0;
};

struct pmdi_cluster_batch {
    pmdi_handle *h = nullptr;
    int k = 0, B = 0;
    DsetDev d{};
    void *arena = nullptr;
    DevBuf d_rows, d_flags, d_out;
};

// Device-resident Gibbs state of every chain of a handle (pmdi_gibbs_* entry points)
struct pmdi_gibbs {
    pmdi_handle *h = nullptr;
    int device = 0;                      // (cached: destroy must not have to look at the handle)
    GibbsArgs ga{};
    int *s_next = nullptr;               // the sweep's output; exchanged with ga.s after every sweep
    unsigned char *flags = nullptr;      // [chain][sumD] featureFlag
    double *fprob = nullptr;             // [chain][sumD] featureProb of the last feature selection
    double *lw = nullptr;                // [chain][P]
    int *pstar = nullptr;                // [chain]
    long long *stats = nullptr;          // [chain][8]
    int *err = nullptr;                  // [chain]
    long long n1 = 0;
    int feature_select = 0;
    int64_t iter = 0;                    // Gibbs iterations done
    std::vector<void *> owned;
};

namespace {

int dev_alloc(pmdi_handle *h, void **p, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { *p = nullptr; return fail(PMDI_E_MEMORY, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    h->owned.push_back(*p);
    return 0;
}

template <class Tt>
int upload(pmdi_handle *h, const std::vector<Tt> &v, const Tt **out)
{
    void *p = nullptr;
    int rc = dev_alloc(h, &p, v.size() * sizeof(Tt));
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(Tt), hipMemcpyHostToDevice));
    *out = (const Tt *)p;
    return 0;
}

void fill_sweep_common(const pmdi_handle *h, SweepArgs &a)
{
    memset(&a, 0, sizeof(a));
    a.K = h->cfg.K; a.N = h->cfg.N; a.P = h->cfg.P; a.cap = (int)h->cap;
    a.Dmax = h->Dmax; a.sumD = h->sumD; a.npairs = h->npairs;
    a.q1 = h->cfg.q1_mode; a.q2 = h->cfg.q2_mode;
    a.terms_cap = h->terms_cap;
    a.item_cap = h->cfg.N > 32 ? PMDI_ITEM_CAP_BIGN : PMDI_ITEM_CAP;
    a.pid_lds = h->pid_lds; a.pp_lds = h->pp_lds; a.col_lds = h->col_lds; a.two_per_cu = h->two_per_cu;
    a.phase = h->phase_on ? (long long *)h->d_phase.p : nullptr;
    a.cost = (long long *)h->d_cost.p;
    a.work = (long long *)h->d_work.p;
    a.anclog = (int *)h->d_anclog.p; a.evpos = (int *)h->d_evpos.p;
    a.ksplit = h->ksplit; a.xcnt = (int *)h->d_xcnt.p; a.xinc = (double *)h->d_xinc.p; a.xlab = (int *)h->d_xlab.p; a.xhdr = (unsigned long long *)h->d_xhdr.p;
    a.chain_order = h->have_order ? (const int *)h->d_lorder.p : nullptr;
    a.s2 = h->s2;
    a.requeue = h->s2_ok ? (int *)h->d_requeue.p : nullptr;
    a.requeue_total = h->s2_ok ? (long long *)h->d_requeue_total.p : nullptr;
    a.handed = h->s2_ok ? (int *)h->d_handed.p : nullptr;
    a.sweep_no = h->sweep_no;
    a.swept_by = (int *)h->d_swept_by.p;
    a.resume = (h->s2_ok && h->s2_continue) ? (int *)h->d_resume.p : nullptr;
    a.n = h->cfg.n;
    a.seed = h->cfg.seed;
    for (int k = 0; k < h->cfg.K; ++k) a.ds[k] = h->ds[k];
}

// next pinned staging slot of the handle's ring (waits -- normally not at all -- for the copy that last used it)
SweepArgs *next_slot(pmdi_handle *h, hipStream_t st_of_copy, int *idx)
{
    (void)st_of_copy;
    if (!h->ring) { *idx = -1; return nullptr; }
    const int i = h->ring_head;
    h->ring_head = (i + 1) % pmdi_handle::RING;
    if (h->ring_used[i]) (void)hipEventSynchronize(h->ring_ev[i]);
    *idx = i;
    return h->ring + i;
}

hipError_t launch_one(pmdi_handle *h, const SweepArgs &a, SweepArgs *d_args, int n_chains, int T, hipStream_t st)
{
    int idx;
    SweepArgs *slot = next_slot(h, st, &idx);
    hipError_t e = pmdi_launch_sweep(a, d_args, n_chains, T, st, slot);
    if (e == hipSuccess && idx >= 0) { e = hipEventRecord(h->ring_ev[idx], st); h->ring_used[idx] = true; }
    return e;
}

// split mode: all chain slots in one launch when their K workgroups each are resident at once, else in residency-sized batches
// one after the other (a chain whose partners straddle the residency limit would spin until some other chain's whole sweep ends)
hipError_t launch_maybe_batched(pmdi_handle *h, const SweepArgs &a, SweepArgs *d_args, int n_chains, int T, hipStream_t st)
{
    if (!a.ksplit || h->ksplit_batch <= 0 || n_chains <= h->ksplit_batch) return launch_one(h, a, d_args, n_chains, T, st);
    for (int b0 = 0; b0 < n_chains; b0 += h->ksplit_batch) {
        SweepArgs ab = a;
        ab.n_slots = n_chains; ab.slot_base = b0;
        const int nb = (n_chains - b0 < h->ksplit_batch) ? n_chains - b0 : h->ksplit_batch;
        hipError_t e = launch_one(h, ab, d_args, nb, T, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// One sweep of all chains on stream `st`: a single launch, or (automatic width) the heavy chains in
// wide workgroups on `st` and the light ones in 256-thread workgroups on the handle's second stream,
// forked from and joined back into `st`.  Then the launch order and groups of the next sweep.
int launch_sweep_groups(pmdi_handle *h, SweepArgs &a, hipStream_t st)
{
    const int C = h->cfg.n_chains;
    hipError_t e;
    a.err_keep = h->err_keep;
    a.sweep_no = ++h->sweep_no;        // (counts the sweeps of this handle: how long ago was a chain given back?)
    // nobody is on the list of given-back chains when a sweep starts (a chain whose re-run failed must not be swept by two launches
    // of the next sweep at once)
    if (h->s2_ok) HIP_TRY(hipMemsetAsync(h->d_requeue.p, 0, (size_t)C * 4, st));
    if (h->ksplit) {       // arrival counters of the hand-offs; the K workgroups of a chain ADD their counters into stats
        HIP_TRY(hipMemsetAsync(h->d_xcnt.p, 0, (size_t)C * 32 * 4, st));
        HIP_TRY(hipMemsetAsync(a.stats, 0, (size_t)C * 64, st));
        if (!h->err_keep) HIP_TRY(hipMemsetAsync(a.err, 0, (size_t)C * 4, st));
    }
    if (!h->split) {
        e = launch_maybe_batched(h, a, (SweepArgs *)h->d_args.p, C, h->T, st);
        if (e != hipSuccess) return fail(PMDI_E_DEVICE, "sweep launch: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(hipEventRecord(h->ev_fork, st));
        a.group_flag = (const unsigned char *)h->d_group.p;
        a.group_sel = 1;
        const int vh = h->very_heavy < C ? h->very_heavy : C;
        if (vh > 0) {
            // the heaviest chains bound the launch: they get the 256-register build, one CU each
            SweepArgs av = a;
            av.two_per_cu = 0; av.rank_lo = 0; av.rank_hi = vh;
            HIP_TRY(hipStreamWaitEvent(h->stream3, h->ev_fork, 0));
            // A workgroup of that build needs a CU to itself.  If the other two launches were released at the same moment their
            // smaller workgroups would take a slot on every CU first and the heaviest chains -- the ones that bound the sweep --
            // would start late, one by one, whenever a CU happens to drain (measured at cfg2: 1 020 ms for a launch whose chains
            // take 550 ms).  So its workgroups count themselves in on entry and the other launches wait for that count.
            const bool gate = h->start_sig != nullptr && !h->ksplit;
            if (gate) av.start_sig = h->start_sig;      // (zero here: reset on `st` behind the previous sweep's join, below)
            e = launch_one(h, av, (SweepArgs *)h->d_args3.p, vh, h->T, h->stream3);
            if (e != hipSuccess) return fail(PMDI_E_DEVICE, "sweep launch (heaviest chains): %s", hipGetErrorString(e));
            HIP_TRY(hipEventRecord(h->ev_join3, h->stream3));
            if (gate) {
                const unsigned n_wg = (unsigned)vh;
                HIP_TRY(hipStreamWaitValue32(st, h->start_sig, n_wg, hipStreamWaitValueGte, 0xFFFFFFFFu));
                HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
                HIP_TRY(hipStreamWaitValue32(h->stream2, h->start_sig, n_wg, hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
        }
        a.rank_lo = vh; a.rank_hi = C;
        e = launch_maybe_batched(h, a, (SweepArgs *)h->d_args.p, C, h->T, st);
        if (e != hipSuccess) return fail(PMDI_E_DEVICE, "sweep launch (heavy group): %s", hipGetErrorString(e));
        if (vh > 0) HIP_TRY(hipStreamWaitEvent(st, h->ev_join3, 0));
        SweepArgs al = a;
        al.group_sel = 0; al.rank_lo = 0; al.rank_hi = C;
        al.terms_cap = h->l_terms_cap; al.pid_lds = h->l_pid_lds; al.pp_lds = h->l_pp_lds; al.col_lds = h->l_col_lds;
        HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        if (h->s2_ok) {
            // the light chains go to the settled-chain kernel.  A chain whose step does not fit its tables is carried on from that
            // observation by the general kernel's code in the same workgroup: the argument block holds the general kernel's LDS
            // layout for that workgroup width
            al.terms_cap = h->r_terms_cap; al.pid_lds = h->r_pid_lds; al.pp_lds = h->r_pp_lds; al.col_lds = h->r_col_lds;
            int idx;
            SweepArgs *slot = next_slot(h, h->stream2, &idx);
            SweepArgs as = al;
            // its workgroups draw their positions in the launch order instead of taking blockIdx.x (SweepArgs::ticket; PMDI_TICKET=0: dealt)
            as.ticket = (h->tun.ticket != 0) ? (int *)h->d_ticket.p : nullptr;
            e = pmdi_launch_sweep2(as, (SweepArgs *)h->d_args4.p, C, h->stream2, slot);
            if (e == hipSuccess && idx >= 0) { e = hipEventRecord(h->ring_ev[idx], h->stream2); h->ring_used[idx] = true; }
            if (e != hipSuccess) return fail(PMDI_E_DEVICE, "sweep launch (settled chains): %s", hipGetErrorString(e));
            if (!h->s2_continue) {
                // (PMDI_CONTINUE=0, the round-3 behaviour kept for A/B runs: no hand-over record; the chains that kernel gives back are
                // swept again FROM THE START by the general kernel right behind it -- with one workgroup per dataset when the model has several)
                SweepArgs ar = al;
                ar.group_flag = nullptr; ar.requeue_only = 1;
                ar.terms_cap = h->l_terms_cap; ar.pid_lds = h->l_pid_lds; ar.pp_lds = h->l_pp_lds; ar.col_lds = h->l_col_lds;
                if (h->cfg.K > 1 && h->d_xcnt.p && h->tun.requeue_ksplit != 0) {
                    ar.ksplit = 1;
                    HIP_TRY(hipMemsetAsync(h->d_xcnt.p, 0, (size_t)C * 32 * 4, h->stream2));
                }
                e = launch_one(h, ar, (SweepArgs *)h->d_args5.p, C, 256, h->stream2);
                if (e != hipSuccess) return fail(PMDI_E_DEVICE, "sweep launch (chains given back by the settled-chain kernel): %s", hipGetErrorString(e));
            }
        } else {
            e = launch_maybe_batched(h, al, (SweepArgs *)h->d_args2.p, C, 256, h->stream2);
            if (e != hipSuccess) return fail(PMDI_E_DEVICE, "sweep launch (light group): %s", hipGetErrorString(e));
        }
        HIP_TRY(hipEventRecord(h->ev_join, h->stream2));
        HIP_TRY(hipStreamWaitEvent(st, h->ev_join, 0));
        // re-arm the start gate only here: `st` has now joined BOTH gated consumers (its own wait and stream2's), so neither can
        // still be waiting for the value that is being reset
        if (vh > 0 && h->start_sig && !h->ksplit) HIP_TRY(hipStreamWriteValue32(st, h->start_sig, 0, 0));
    }
    if (C > 1 || h->split) {
        const long long steps = (a.n - a.n1 + 1) * (long long)a.K;
        e = pmdi_launch_chain_order((const long long *)h->d_cost.p, (int *)h->d_lorder.p, a.stats,
                                    h->split ? (unsigned char *)h->d_group.p : nullptr, h->light_ids * steps, C, st,
                                    h->s2_ok ? (const int *)h->d_handed.p : nullptr, h->sweep_no + 3 - h->sticky);
        if (e != hipSuccess) return fail(PMDI_E_DEVICE, "chain-order launch: %s", hipGetErrorString(e));
        h->have_order = true;
    }
    return PMDI_OK;
}

}  // namespace

extern "C" {

const char *pmdi_last_error(void) { return g_err; }

void pmdi_tuning_default(pmdi_tuning *t)
{
    if (!t) return;
    int32_t *f = (int32_t *)t;
    for (size_t i = 0; i < sizeof(pmdi_tuning) / sizeof(int32_t); ++i) f[i] = -1;
}

// The one place of the library that reads the environment -- and only when a caller asks for it.
void pmdi_tuning_from_env(pmdi_tuning *t)
{
    if (!t) return;
    pmdi_tuning_default(t);
    auto env = [](const char *name, int32_t &field) { const char *v = getenv(name); if (v && *v) field = (int32_t)atoi(v); };
    env("PMDI_SETTLED", t->settled); env("PMDI_CONTINUE", t->continue_inplace); env("PMDI_STICKY", t->sticky);
    env("PMDI_LIGHT_IDS", t->light_ids); env("PMDI_S2_COLS", t->s2_cols); env("PMDI_S2_IDCAP", t->s2_idcap); env("PMDI_S2_CLS", t->s2_cls);
    env("PMDI_KSPLIT", t->ksplit); env("PMDI_REQUEUE_KSPLIT", t->requeue_ksplit); env("PMDI_SPLIT", t->split);
    env("PMDI_HEAVY_T", t->heavy_threads); env("PMDI_TWO_PER_CU", t->two_per_cu); env("PMDI_VERY_HEAVY", t->very_heavy);
    env("PMDI_START_GATE", t->start_gate); env("PMDI_TICKET", t->ticket); env("PMDI_TERMS_CAP", t->terms_cap); env("PMDI_LDS_TARGET", t->lds_target);
    if (getenv("PMDI_PHASE_TIMERS")) t->phase_timers = 1;
    auto has = [](const char *name, const char *what) { const char *v = getenv(name); return v && strstr(v, what) != nullptr; };
    t->profiled = (has("LD_PRELOAD", "rocprof") || has("ROCP_TOOL_LIBRARIES", "rocprof") || has("HSA_TOOLS_LIB", "rocprof")) ? 1 : 0;
}
int pmdi_abi_version(void) { return PMDI_ABI_VERSION; }

int pmdi_destroy(pmdi_handle *h)
{
    if (!h) return PMDI_OK;
    if (h->children > 0)
        return fail(PMDI_E_STATE, "pmdi_destroy: %d object(s) created from this handle (pmdi_gibbs_create / pmdi_clusters_new) are still alive", h->children);
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < pmdi_handle::RING; ++i) if (h->ring_ev[i]) (void)hipEventDestroy(h->ring_ev[i]);
    if (h->ring) (void)hipHostFree(h->ring);
    for (void *p : h->owned) (void)hipFree(p);
    DevBuf *bufs[] = {&h->d_s_in, &h->d_order, &h->d_Pi, &h->d_logphi, &h->d_flags, &h->d_s_out, &h->d_lw,
                      &h->d_pstar, &h->d_stats, &h->d_err, &h->d_trace, &h->d_usc, &h->d_partstar, &h->d_kstate, &h->d_phase, &h->d_args, &h->d_args2, &h->d_args3, &h->d_args4, &h->d_args5, &h->d_requeue, &h->d_requeue_total, &h->d_handed, &h->d_group, &h->d_cost, &h->d_lorder, &h->d_ticket, &h->d_work, &h->d_anclog, &h->d_evpos, &h->d_xcnt, &h->d_xinc, &h->d_xlab, &h->d_xhdr,
                      &h->d_swept_by, &h->d_resume, &h->d_traj, &h->d_lm, &h->d_firstpos, &h->d_fnull, &h->d_fflags, &h->d_fprob};
    for (DevBuf *b : bufs) b->release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream3) { (void)hipStreamSynchronize(h->stream3); (void)hipStreamDestroy(h->stream3); }
    if (h->ev_join3) (void)hipEventDestroy(h->ev_join3);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
    return PMDI_OK;
}

int pmdi_create(const pmdi_config *cfg, const pmdi_dataset *datasets, pmdi_handle **out)
{
    if (!cfg || !datasets || !out) return fail(PMDI_E_ARG, "null argument");
    *out = nullptr;
    if (cfg->abi_version != PMDI_ABI_VERSION) return fail(PMDI_E_ARG, "abi_version %d != %d", cfg->abi_version, PMDI_ABI_VERSION);
    const int K = cfg->K, N = cfg->N, P = cfg->P;
    const long long n = cfg->n;
    // the @asserts of src/pmdi.jl:50-55 that concern this path
    if (K < 1 || K > PMDI_KMAX_I) return fail(PMDI_E_ARG, "K=%d outside 1..%d", K, PMDI_KMAX_I);
    if (n < 2 || n > 0x7fffffffLL / 4) return fail(PMDI_E_ARG, "n=%lld out of range", n);
    if (!(N <= n && N > 1)) return fail(PMDI_E_ARG, "Number of clusters must be greater than 1 and not greater than the number of observations");
    if (N > 255) return fail(PMDI_E_ARG, "N=%d: this build supports N <= 255 (labels travel as bytes; the mutation CDF of a particle class is formed by one wave, up to four labels per lane, pairwise beyond 128 like Base.cumsum)", N);
    if (P < 2) return fail(PMDI_E_ARG, "Conditional particle filter requires 2 or more particles");
    if (P > 1048575) return fail(PMDI_E_ARG, "P=%d too large", P);
    if (cfg->n_chains < 1) return fail(PMDI_E_ARG, "n_chains must be >= 1");
    if (cfg->q1_mode < 0 || cfg->q1_mode > 1) return fail(PMDI_E_ARG, "q1_mode must be 0 or 1");
    if (cfg->q2_mode < 0 || cfg->q2_mode > 1) return fail(PMDI_E_ARG, "q2_mode must be 0 or 1");
    long long cap = cfg->pool_cap > 0 ? cfg->pool_cap : (long long)N * P + 1;
    if (cap > (long long)N * P + 1) cap = (long long)N * P + 1;
    if (cap < N + 2) return fail(PMDI_E_ARG, "pool_cap too small");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(PMDI_E_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(PMDI_E_DEVICE, "device %d not in 0..%d", cfg->device, ndev - 1);
    HIP_TRY(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PMDI_E_DEVICE, "device %d is %s; the kernels are built for gfx950 only", cfg->device, prop.gcnArchName);

    pmdi_handle *h = new (std::nothrow) pmdi_handle();
    if (!h) return fail(PMDI_E_MEMORY, "out of host memory");
    h->cfg = *cfg;
    if (cfg->tuning) h->tun = *cfg->tuning; else pmdi_tuning_default(&h->tun);
    h->cfg.tuning = nullptr;
    const pmdi_tuning &tn = h->tun;
    h->cap = cap;
    h->npairs = K > 1 ? K * (K - 1) / 2 : 1;
    int rc = 0;
    auto bail = [&](int code) { pmdi_destroy(h); return code; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "hipStreamCreate failed"));

    int flag_off = 0;
    for (int k = 0; k < K; ++k) {
        const pmdi_dataset &src = datasets[k];
        DsetDev &d = h->ds[k];
        if (src.D < 1) return bail(fail(PMDI_E_ARG, "dataset %d: D=%d", k, src.D));
        if (src.ld < n) return bail(fail(PMDI_E_ARG, "dataset %d: ld=%lld < n. Datasets don't have same number of observations.", k, (long long)src.ld));
        d.kind = src.kind; d.D = src.D; d.L = 0; d.flag_off = flag_off;
        flag_off += src.D;
        if (src.D > h->Dmax) h->Dmax = src.D;
        const int D = src.D;
        if (src.kind == PMDI_GAUSSIAN) {
            if (!src.xf) return bail(fail(PMDI_E_ARG, "dataset %d: xf is null", k));
            std::vector<double> x((size_t)n * D);
            for (long long i = 0; i < n; ++i)
                for (int q = 0; q < D; ++q) x[(size_t)i * D + q] = src.xf[(size_t)q * src.ld + i];
            if ((rc = upload(h, x, &d.xf))) return bail(rc);
            // gaussian_cluster.jl:38-40 prefix and :76-79 constant, per cluster size
            std::vector<double> g((size_t)n + 1), lm((size_t)n + 1);
            for (long long m = 0; m <= n; ++m) {
                const double nn = (double)m;
                g[m] = (log(1.0 / sqrt(M_PI)) + lgamma(0.5 * nn + 1.0)) - lgamma(0.5 * nn + 0.5);
                const double a_n = (nn / 2.0 + 0.5), a_0 = 0.5, b_0 = 0.5, k_0 = 0.001, k_n = nn + k_0;
                lm[m] = (a_0 * log(b_0)) + lgamma(a_n) - lgamma(a_0) + 0.5 * (log(k_0) - log(k_n)) -
                        (nn * 0.5) * log(2.0 * M_PI);
            }
            if ((rc = upload(h, g, &d.gtab))) return bail(rc);
            if ((rc = upload(h, lm, &d.lmtab))) return bail(rc);
        } else if (src.kind == PMDI_CATEGORICAL || src.kind == PMDI_NEGBINOM) {
            if (!src.xi) return bail(fail(PMDI_E_ARG, "dataset %d: xi is null", k));
            std::vector<int> x((size_t)n * D);
            std::vector<int> maxcol(D, 0);
            std::vector<long long> colsum(D, 0);
            long long gmax = 0;
            for (long long i = 0; i < n; ++i)
                for (int q = 0; q < D; ++q) {
                    const long long v = src.xi[(size_t)q * src.ld + i];
                    if (src.kind == PMDI_CATEGORICAL && v < 1) return bail(fail(PMDI_E_DATA, "dataset %d: categorical level %lld < 1", k, v));
                    if (src.kind == PMDI_NEGBINOM && v < 0) return bail(fail(PMDI_E_DATA, "dataset %d: negative count %lld", k, v));
                    if (v > 0x3fffffff) return bail(fail(PMDI_E_DATA, "dataset %d: value %lld too large", k, v));
                    x[(size_t)i * D + q] = (int)v;
                    if (v > maxcol[q]) maxcol[q] = (int)v;
                    if (v > gmax) gmax = v;
                    colsum[q] += v;
                }
            if ((rc = upload(h, x, &d.xi))) return bail(rc);
            if (src.kind == PMDI_CATEGORICAL) {
                if (gmax > 4096) return bail(fail(PMDI_E_DATA, "dataset %d: %lld categorical levels (max 4096)", k, gmax));
                d.L = (int)gmax;                                  // categorical_cluster.jl:8
                if ((rc = upload(h, maxcol, &d.maxcol))) return bail(rc);
                std::vector<double> lh((size_t)(2 * n + gmax + 3)), lgh((size_t)(2 * (n + gmax) + 3));
                for (size_t j = 0; j < lh.size(); ++j) lh[j] = log(0.5 * (double)j);
                for (size_t j = 0; j < lgh.size(); ++j) lgh[j] = lgamma(0.5 * (double)j);
                if ((rc = upload(h, lh, &d.lhtab))) return bail(rc);
                if ((rc = upload(h, lgh, &d.lghtab))) return bail(rc);
            } else {
                long long smax = 0;
                for (int q = 0; q < D; ++q) if (colsum[q] > smax) smax = colsum[q];
                const long long len = n + gmax + smax + 8;
                if (len > (1LL << 27)) return bail(fail(PMDI_E_DATA, "dataset %d: NegBinom lgamma table of %lld entries is too large", k, len));
                std::vector<double> lg((size_t)len);
                for (long long m = 0; m < len; ++m) lg[m] = lgamma((double)m);
                if ((rc = upload(h, lg, &d.lgtab))) return bail(rc);
                d.lgtab_len = len;
            }
        } else {
            return bail(fail(PMDI_E_ARG, "dataset %d: unknown kind %d", k, src.kind));
        }
        d.stride = layout_arena(d, N, P, cap, n, true);
        void *arena = nullptr;
        if ((rc = dev_alloc(h, &arena, d.stride * (size_t)cfg->n_chains))) return bail(rc);
        d.arena = (char *)arena;
        if (hipMemset(arena, 0, d.stride * (size_t)cfg->n_chains) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "hipMemset failed"));
    }
    h->sumD = flag_off;
    h->T = cfg->block_threads ? cfg->block_threads : (P >= 2048 ? 1024 : (P > 256 ? 512 : 256));
    if (!cfg->block_threads && h->T > 256 && tn.heavy_threads > 0) h->T = tn.heavy_threads;   // tuning knob: width of the heavy group
    if (h->T != 128 && h->T != 256 && h->T != 512 && h->T != 1024) return bail(fail(PMDI_E_ARG, "block_threads must be 128, 256, 512 or 1024"));
    {
        auto knob = [](int32_t v, int dflt) { return v >= 0 ? (int)v : dflt; };      // (-1 = automatic)
        // (a step whose particle classes' (class, label) items outgrow the LDS tables takes the general route through global memory;
        // one class always fits: N <= 255 < 384)
        if (N > (N > 32 ? PMDI_ITEM_CAP_BIGN : PMDI_ITEM_CAP)) return bail(fail(PMDI_E_ARG, "N=%d too large for the LDS tables", N));
        // K > 1: one workgroup per (chain, dataset), meeting once per swept observation (pmdi_sweep.hip): 2.2x shorter sweeps per
        // chain, but the partners repeat the per-observation serial work (weights, ESS, resampling indices), so when the chains alone
        // can fill the GPU the single-workgroup form has the higher aggregate throughput.  Default: split while every workgroup of
        // the split launch is RESIDENT at once -- decided below, after the LDS layout and the register-capped / uncapped build are
        // known, from the occupancy of the kernel that will actually run (a chain whose K workgroups straddle the residency limit
        // would spin in its hand-off until some other chain's whole sweep has ended).  PMDI_KSPLIT=0/1 forces either form (a forced
        // split launch that is not resident at once goes out in residency-sized batches).  The history-permuting __pmdi mode
        // (q2_mode = 1) always keeps the single-workgroup form: its ancestor log is per chain.
        const bool ks_ok = K > 1 && cfg->q2_mode == 0 && (long long)K * n < (1LL << 27);
        const int ks_env = tn.ksplit;
        // ... unless the handle can have the settled-chain kernel (round 4): ONE workgroup of it sweeps a settled chain faster than the K
        // cooperating workgroups of the general kernel do (HL, one chain alone on the GPU: 3.18 against 2.98-3.00 iterations/s,
        // bench.py `one_chain_alone`), on a quarter of the workgroup slots
        const bool s2_possible = knob(tn.settled, 1) != 0 && K >= 2 && cfg->block_threads == 0 && cfg->q1_mode == 0 && cfg->q2_mode == 0 &&
                                 pmdi_sweep2_supports(K, N, P, h->Dmax, cap);
        h->ksplit = (ks_ok && ks_env != 0 && !(ks_env < 0 && s2_possible)) ? 1 : 0;      // tentative: the layout below is the split form's
        h->phase_on = tn.phase_timers > 0;
        // per workgroup width: LDS term buffer (at least P doubles for the resampling weights, the
        // per-wave CDF exchange areas, a few rows of 2*D+1) and which per-particle tables fit LDS
        auto configure = [&](int T, int &terms_cap, int &pid_lds, int &pp_lds, int &col_lds) -> int {
            int tc = knob(tn.terms_cap, 1024);
            if (tc < P) tc = P;
            if (tc < (T / 64) * (N > 64 ? 512 : 128)) tc = (T / 64) * (N > 64 ? 512 : 128);   // per-wave exchange areas of the CDF stage
            if (tc < 4 * (2 * h->Dmax + 1)) tc = 4 * (2 * h->Dmax + 1);
            if (tc < 384) tc = 384;                                   // the known-prefix label table (3 x 256 ints) lives there
            terms_cap = tc;
            SweepArgs a;
            fill_sweep_common(h, a);
            a.terms_cap = tc; a.pid_lds = 1; a.pp_lds = 1;
            const int col_ok = 1;
            a.col_lds = col_ok;
            // Two chains per CU (<= 80 KiB each) hide each other's dependent latencies: +40 % aggregate throughput on the
            // headline workload even with the per-particle tables in global memory.  So: the largest set of per-particle
            // tables that still fits 80 KiB; if none does, one chain per CU with everything that fits 150 KiB in LDS.
            const bool tgt = tn.lds_target >= 0;
            const size_t half = 80 * 1024 - 256;      // half a CU's 160 KiB, less the kernel's static LDS (256 bytes of reduction scratch)
            bool fits_half = false;
            if (!tgt) {
                // (what a step reads of them, most often first: the column indices -- every step --, the step scratch, the class ids
                // -- only in steps with more than one particle class)
                for (int opt = 0; opt < 4 && !fits_half; ++opt) {
                    a.pid_lds = opt < 1; a.pp_lds = opt < 3; a.col_lds = (opt < 2) ? col_ok : 0;
                    fits_half = pmdi_sweep_lds_bytes(a, T) <= half;
                }
            }
            if (!fits_half) {
                a.pid_lds = 1; a.pp_lds = 1; a.col_lds = col_ok;
                const size_t lds_target = tgt ? (size_t)tn.lds_target : (size_t)150 * 1024;
                if (pmdi_sweep_lds_bytes(a, T) > lds_target) a.pid_lds = 0;   // class ids of K*P particles do not fit: global memory
                if (pmdi_sweep_lds_bytes(a, T) > lds_target) a.pp_lds = 0;    // nor does the per-particle step scratch
                if (pmdi_sweep_lds_bytes(a, T) > lds_target) a.col_lds = 0;   // nor do the column indices
            }
            pid_lds = a.pid_lds; pp_lds = a.pp_lds; col_lds = a.col_lds;
            if (pmdi_sweep_lds_bytes(a, T) > 160 * 1024)
                return fail(PMDI_E_ARG, "configuration needs %zu bytes of LDS (> 160 KiB)", pmdi_sweep_lds_bytes(a, T));
            return 0;
        };
        auto configure_wide = [&]() -> int {
            h->two_per_cu = knob(tn.two_per_cu, 1);
            const int r = configure(h->T, h->terms_cap, h->pid_lds, h->pp_lds, h->col_lds);
            if (r) return r;
            // one chain per CU anyway: the 256-register build (no spills) instead of the register-capped one
            SweepArgs a;
            fill_sweep_common(h, a);
            if (tn.two_per_cu < 0 && pmdi_sweep_lds_bytes(a, h->T) > 80 * 1024 - 256) h->two_per_cu = 0;
            return 0;
        };
        if ((rc = configure_wide())) return bail(rc);
        if (h->ksplit) {
            SweepArgs a;
            fill_sweep_common(h, a);
            int blocks = 0;
            if (pmdi_sweep_blocks_per_cu(a, h->T, &blocks) != hipSuccess || blocks < 1) blocks = 1;
            const long long capacity = (long long)blocks * prop.multiProcessorCount;
            const long long grid = ((long long)(cfg->n_chains + 7) / 8) * 8 * K;       // chain slots are dealt in groups of eight
            if (grid > capacity) {
                if (ks_env < 0) {                        // default: the single-workgroup form (and its own LDS layout)
                    h->ksplit = 0;
                    if ((rc = configure_wide())) return bail(rc);
                } else {                                 // forced: whole groups of eight chain slots that are resident together
                    long long b = capacity / (8LL * K) * 8;
                    h->ksplit_batch = (int)(b < 8 ? 8 : b);
                }
            }
        }
        // the settled-chain kernel: configurations of the shapes it is built for (any mix of the three cluster types), the reference's
        // default quirk modes, one workgroup per chain; its LDS tables sized so that two 256-thread chains share a CU (fewer LDS columns /
        // ids if need be); a 512-thread chain (P = 2 048) has a CU's registers to itself and may take its LDS too
        {
            // (K = 1: three of its four waves would idle through the cluster phases -- cfg2 runs 2 812 it/s on the general kernel's
            // one-dataset build and 2 086 with this kernel; PMDI_SETTLED=2 forces it for the tests of its K = 1 instantiations)
            bool ok = knob(tn.settled, 1) != 0 && (K >= 2 || knob(tn.settled, 1) == 2) && cfg->block_threads == 0 && !h->ksplit && cfg->q1_mode == 0 && cfg->q2_mode == 0 &&
                      pmdi_sweep2_supports(K, N, P, h->Dmax, cap);
            if (ok) {
                const size_t budget = pmdi_sweep2_threads(K, P) > 256 ? (size_t)159 * 1024 : (size_t)80 * 1024;
                int cols_l = knob(tn.s2_cols, 64), idcap = knob(tn.s2_idcap, 128);
                if (cols_l < 1) cols_l = 1;
                if (cols_l > P) cols_l = P;
                if (idcap < 8) idcap = 8;
                if (idcap > 4096) idcap = 4096;
                // particle classes per dataset the LDS tables hold: as many as the budget allows, 32 at most (a step with more hands the
                // chain over to the general kernel: with N labels a single ambiguous observation fans one class out into up to N)
                int cls = knob(tn.s2_cls, 32);
                if (cls < 16) cls = 16;
                if (cls > pmdi_sweep2_max_classes(K, P)) cls = pmdi_sweep2_max_classes(K, P);
                cls &= ~3;
                // (the mutation-CDF rows of the class slots beyond the first 16 go to the chain's arena before the class count shrinks: a
                // step with that many classes is rare, a hand-over costs the rest of the chain's sweep at the general kernel's speed)
                int cdfl = cls;
                pmdi_sweep2_layout(K, N, P, h->Dmax, cols_l, idcap, cls, cdfl, &h->s2);
                if ((size_t)h->s2.total > budget && cls > 16 && pmdi_sweep2_cdf_arena(K, P)) { cdfl = 16; pmdi_sweep2_layout(K, N, P, h->Dmax, cols_l, idcap, cls, cdfl, &h->s2); }
                while ((size_t)h->s2.total > budget && cls > 16) { cls -= 4; if (cdfl > cls) cdfl = cls; pmdi_sweep2_layout(K, N, P, h->Dmax, cols_l, idcap, cls, cdfl, &h->s2); }
                while ((size_t)h->s2.total > budget && (cols_l > 16 || idcap > 64)) {
                    // (a settled chain holds 6-40 columns and ids below ~40 at the 99th percentile of its steps: the id tables go first)
                    if (idcap > 96) idcap -= 16; else if (cols_l > 32) cols_l -= 8; else if (idcap > 64) idcap -= 16; else cols_l -= 8;
                    pmdi_sweep2_layout(K, N, P, h->Dmax, cols_l, idcap, cls, cdfl, &h->s2);
                }
                ok = (size_t)h->s2.total <= budget;
            }
            h->s2_ok = ok;
        }
        // automatic width: split the chains of a sweep into a heavy and a light launch
        h->split = cfg->block_threads == 0 && (h->T > 256 || h->s2_ok) && knob(tn.split, 1) != 0;
        if (!h->split) h->s2_ok = false;
        // A chain is "light" when its last sweep met few live clusters per step.  With the settled-chain kernel: as many as that kernel's
        // LDS id tables hold (it evaluates any number in place -- HL's busiest chains, 40-90 ids per step, run 20 % faster there than
        // in the general kernel -- but a K = 1 chain that never resamples keeps thousands of private clusters, and those steps belong to
        // the general kernel's hash tables: cfg2 with every chain light took 3.4 s per sweep instead of 0.8).  Without it: 40.
        h->light_ids = knob(tn.light_ids, h->s2_ok ? h->s2.idcap : 40);
        if (h->s2_ok && knob(tn.settled, 1) == 2) h->light_ids = 1LL << 40;      // (tests: every chain starts every sweep on the settled-chain kernel)
        h->sticky = knob(tn.sticky, 3);
        h->s2_continue = knob(tn.continue_inplace, 1) != 0;
        if (h->split) {
            if (configure(256, h->l_terms_cap, h->l_pid_lds, h->l_pp_lds, h->l_col_lds)) h->split = false;
        }
        if (h->split && h->s2_ok) {
            const int thr = pmdi_sweep2_threads(K, P);
            if (thr == 256) { h->r_terms_cap = h->l_terms_cap; h->r_pid_lds = h->l_pid_lds; h->r_pp_lds = h->l_pp_lds; h->r_col_lds = h->l_col_lds; }
            else if (configure(thr, h->r_terms_cap, h->r_pid_lds, h->r_pp_lds, h->r_col_lds)) h->s2_continue = false;
        }
        if (!h->split) h->s2_ok = false;
        if (h->split) {
            // the slow chains bound the launch, so their workgroups must not queue behind the many
            // light ones: light launch on a low-priority stream, heaviest chains on a high-priority one
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // (least, greatest); greatest is the smaller number
            if (hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_lo) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
                hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_join3, hipEventDisableTiming) != hipSuccess)
                return bail(fail(PMDI_E_DEVICE, "stream/event creation failed"));
            // (with the settled-chain kernel only the few chains it gave back lately are heavy at all: no CU-each launch for them --
            // HL 1 160 it/s without against 1 146 with it -- and so no start gate either, which a profiler would switch off)
            h->very_heavy = (h->T == 512 && h->two_per_cu) ? knob(tn.very_heavy, h->s2_ok ? 0 : 128) : 0;
            int can_wait = 0;
            (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, cfg->device);
            // (not under a profiler that collects counters: rocprofv3 --pmc runs the kernels of all queues one at a time, and a launch
            // that waits for another launch's workgroups then never starts -- observed as a hang of the FETCH_SIZE pass)
            const bool profiled = tn.profiled > 0;
            if (h->very_heavy > 0 && can_wait && !profiled && knob(tn.start_gate, 1) != 0) {
                void *sig = nullptr;
                if (hipExtMallocWithFlags(&sig, 8, hipMallocSignalMemory) == hipSuccess) {
                    h->start_sig = (unsigned *)sig; h->owned.push_back(sig);
                    (void)hipMemset(sig, 0, 8);
                }
            }
        }
    }
    const int C = cfg->n_chains;
    if ((rc = h->d_usc.ensure((size_t)C * K * P * 8)) || (rc = h->d_partstar.ensure((size_t)C * K * P * 4)) ||
        (rc = h->d_kstate.ensure((size_t)C * PMDI_KMAX_I * 2 * 4)) || (rc = h->d_err.ensure((size_t)C * 4)) ||
        (rc = h->d_stats.ensure((size_t)C * 8 * 8)) || (rc = h->d_pstar.ensure((size_t)C * 4)) ||
        (rc = h->d_phase.ensure((size_t)C * 16 * 8)) || (rc = h->d_args.ensure(sizeof(SweepArgs))) ||
        (rc = h->d_cost.ensure((size_t)C * 8)) || (rc = h->d_work.ensure((size_t)C * PMDI_KMAX_I * 8 * 8)) || (rc = h->d_lorder.ensure((size_t)C * 4)) || (rc = h->d_ticket.ensure(((size_t)C + 1) * 4)) ||
        (rc = h->d_args2.ensure(sizeof(SweepArgs))) || (rc = h->d_args3.ensure(sizeof(SweepArgs))) || (rc = h->d_group.ensure((size_t)C)) ||
        (rc = h->d_args4.ensure(sizeof(SweepArgs))) || (rc = h->d_args5.ensure(sizeof(SweepArgs))) || (rc = h->d_requeue.ensure((size_t)C * 4)) ||
        (rc = h->d_requeue_total.ensure(4 * 8)) || (rc = h->d_handed.ensure((size_t)C * 4)) || (rc = h->d_swept_by.ensure((size_t)C * 4)) ||
        (rc = h->d_resume.ensure((size_t)C * 16 * 4)))
        return bail(rc);
    if (hipMemset(h->d_swept_by.p, 0, (size_t)C * 4) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "hipMemset failed"));
    {
        std::vector<int> never((size_t)C, -1000);
        if (hipMemcpy(h->d_handed.p, never.data(), (size_t)C * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "memcpy failed"));
    }
    if (hipMemset(h->d_requeue.p, 0, (size_t)C * 4) != hipSuccess || hipMemset(h->d_requeue_total.p, 0, 32) != hipSuccess)
        return bail(fail(PMDI_E_DEVICE, "hipMemset failed"));
    {   // pinned staging ring of the argument blocks (asynchronous launches); without it the copies fall back to pageable memory
        void *ring = nullptr;
        if (hipHostMalloc(&ring, sizeof(SweepArgs) * pmdi_handle::RING, hipHostMallocDefault) == hipSuccess) {
            h->ring = (SweepArgs *)ring;
            for (int i = 0; i < pmdi_handle::RING; ++i)
                if (hipEventCreateWithFlags(&h->ring_ev[i], hipEventDisableTiming) != hipSuccess) {
                    for (int j = 0; j < i; ++j) { (void)hipEventDestroy(h->ring_ev[j]); h->ring_ev[j] = nullptr; }
                    (void)hipHostFree(ring); h->ring = nullptr;
                    break;
                }
        }
    }
    if ((h->ksplit || (h->s2_ok && K > 1 && (long long)K * n < (1LL << 27))) && ((rc = h->d_xcnt.ensure((size_t)C * 32 * 4)) || (rc = h->d_xinc.ensure((size_t)C * 2 * K * P * 8)) ||
                      (rc = h->d_xlab.ensure((size_t)C * 2 * K * P * 4)) || (rc = h->d_xhdr.ensure((size_t)C * 2 * K * 16))))
        return bail(rc);
    if (cfg->q2_mode == 1 &&       // ancestor log of the resampling events: up to one per swept observation
        ((rc = h->d_anclog.ensure((size_t)C * (size_t)n * P * 4)) || (rc = h->d_evpos.ensure((size_t)C * 2 * (size_t)n * 4))))
        return bail(rc);
    // first sweep: every chain is heavy (PMDI_SETTLED=2: every chain starts on the settled-chain kernel -- tests of its hand-back path)
    if (hipMemset(h->d_group.p, (h->s2_ok && tn.settled == 2) ? 0 : 1, (size_t)C) != hipSuccess)
        return bail(fail(PMDI_E_DEVICE, "hipMemset failed"));

    // null-cluster marginal (src/pmdi.jl:120-128): all rows in one cluster, all features on
    {
        std::vector<int> traj((size_t)K * n, 0);
        if ((rc = h->d_traj.ensure((size_t)C * K * n * 4)) || (rc = h->d_lm.ensure((size_t)C * h->sumD * N * 8)) ||
            (rc = h->d_firstpos.ensure((size_t)C * K * N * 4)) || (rc = h->d_fnull.ensure((size_t)h->sumD * 8)) ||
            (rc = h->d_fflags.ensure((size_t)C * h->sumD)) || (rc = h->d_fprob.ensure((size_t)C * h->sumD * 8)))
            return bail(rc);
        if (hipMemcpy(h->d_traj.p, traj.data(), traj.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "memcpy failed"));
        std::vector<double> zero((size_t)h->sumD, 0.0);
        if (hipMemcpy(h->d_fnull.p, zero.data(), zero.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "memcpy failed"));
        FeatSelArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.K = K; fa.N = 1; fa.sumD = h->sumD; fa.n = n; fa.iter = 0; fa.seed = cfg->seed;
        for (int k = 0; k < K; ++k) fa.ds[k] = h->ds[k];
        fa.traj = (const int *)h->d_traj.p; fa.lm = (double *)h->d_lm.p; fa.firstpos = (int *)h->d_firstpos.p;
        fa.fnull = (const double *)h->d_fnull.p; fa.flags_out = (unsigned char *)h->d_fflags.p; fa.prob_out = (double *)h->d_fprob.p;
        hipError_t e = pmdi_launch_featsel(fa, 1, h->stream);
        if (e != hipSuccess) return bail(fail(PMDI_E_DEVICE, "null-marginal launch: %s", hipGetErrorString(e)));
        if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "null-marginal kernel failed"));
        std::vector<double> lm((size_t)h->sumD);
        if (hipMemcpy(lm.data(), h->d_lm.p, lm.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "memcpy failed"));
        for (double &v : lm) v = -v;                               // featureNull = -calc_logmarginal (:127)
        if (hipMemcpy(h->d_fnull.p, lm.data(), lm.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(PMDI_E_DEVICE, "memcpy failed"));
    }
    *out = h;
    return PMDI_OK;
}

int pmdi_phase_timers(pmdi_handle *h, int32_t chain, int64_t *out16)
{
    if (!h || !out16 || chain < 0 || chain >= h->cfg.n_chains) return fail(PMDI_E_ARG, "bad argument");
    if (!h->phase_on) return fail(PMDI_E_STATE, "set PMDI_PHASE_TIMERS=1 before pmdi_create");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, (char *)h->d_phase.p + (size_t)chain * 16 * 8, 16 * 8, hipMemcpyDeviceToHost));
    return PMDI_OK;
}

int pmdi_chain_costs(pmdi_handle *h, int64_t *out)
{
    if (!h || !out) return fail(PMDI_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_cost.p, (size_t)h->cfg.n_chains * 8, hipMemcpyDeviceToHost));
    return PMDI_OK;
}
int pmdi_work_counters(pmdi_handle *h, int64_t *out)
{
    if (!h || !out) return fail(PMDI_E_ARG, "null argument");
    const int C = h->cfg.n_chains, K = h->cfg.K;
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<long long> w((size_t)C * PMDI_KMAX_I * 8);
    HIP_TRY(hipMemcpy(w.data(), h->d_work.p, w.size() * 8, hipMemcpyDeviceToHost));
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < 8; ++j) out[((size_t)c * K + k) * 8 + j] = w[((size_t)c * PMDI_KMAX_I + k) * 8 + j];
    return PMDI_OK;
}
int pmdi_block_threads(const pmdi_handle *h) { return h ? h->T : 0; }
int pmdi_is_split(const pmdi_handle *h) { return h ? h->ksplit : 0; }
int64_t pmdi_shader_clock_hz(const pmdi_handle *h)
{
    if (!h) return 0;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, h->cfg.device) != hipSuccess) return 0;
    return (int64_t)khz * 1000;
}
int64_t pmdi_lds_bytes(const pmdi_handle *h)
{
    if (!h) return 0;
    SweepArgs a;
    fill_sweep_common(h, a);
    return (int64_t)pmdi_sweep_lds_bytes(a, h->T);
}
int pmdi_sum_D(const pmdi_handle *h) { return h ? h->sumD : 0; }
int pmdi_settled_kernel(pmdi_handle *h, int64_t *given_back4)
{
    if (!h) return 0;
    if (given_back4) {
        given_back4[0] = given_back4[1] = given_back4[2] = given_back4[3] = 0;
        if (h->s2_ok) {
            if (hipSetDevice(h->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(given_back4, h->d_requeue_total.p, 32, hipMemcpyDeviceToHost) != hipSuccess)
                return fail(PMDI_E_DEVICE, "pmdi_settled_kernel: reading the counters failed");
        }
    }
    return h->s2_ok ? 1 : 0;
}
int pmdi_chain_swept_by(pmdi_handle *h, int32_t *out)
{
    if (!h || !out) return fail(PMDI_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_swept_by.p, (size_t)h->cfg.n_chains * 4, hipMemcpyDeviceToHost));
    return PMDI_OK;
}
int64_t pmdi_pool_cap(const pmdi_handle *h) { return h ? h->cap : 0; }
int pmdi_categorical_L(const pmdi_handle *h, int32_t k) { return (h && k >= 0 && k < h->cfg.K) ? h->ds[k].L : 0; }

int pmdi_sweep_device(pmdi_handle *h, int64_t iter, const int32_t *s_in, const int32_t *order_obs, int64_t n1,
                      const double *Pi, const double *log1p_phi, const uint8_t *feature_flag, double lw_init,
                      int32_t *s_out, double *logweight, int32_t *p_star, int64_t *stats, int32_t *err, void *stream)
{
    if (!h || !s_in || !order_obs || !Pi || !log1p_phi || !s_out) return fail(PMDI_E_ARG, "null argument");
    if (n1 < 1 || n1 > h->cfg.n) return fail(PMDI_E_ARG, "n1=%lld outside 1..n (rho*n < 1 is undefined in the reference)", (long long)n1);
    HIP_TRY(hipSetDevice(h->cfg.device));
    SweepArgs a;
    fill_sweep_common(h, a);
    a.iter = (unsigned)iter; a.n1 = n1; a.lw_init = lw_init;
    a.s_in = s_in; a.order = order_obs; a.Pi = Pi; a.logphi = log1p_phi; a.flags = feature_flag;
    a.s_out = s_out; a.lw_out = logweight;
    a.pstar = p_star ? p_star : (int *)h->d_pstar.p;
    a.stats = stats ? (long long *)stats : (long long *)h->d_stats.p;
    a.err = err ? err : (int *)h->d_err.p;
    a.trace = nullptr; a.trace_on = 0;
    a.uscratch = (double *)h->d_usc.p; a.partstar = (int *)h->d_partstar.p; a.kstate = (int *)h->d_kstate.p;
    hipStream_t st = (hipStream_t)stream;   // used verbatim: NULL is the device's default (null) stream
    const int lrc = launch_sweep_groups(h, a, st);
    if (lrc) return lrc;
    h->swept = true; h->last_n1 = n1;
    return PMDI_OK;
}

int pmdi_psm_counts_device(int32_t device, const uint8_t *samples, int64_t S, int32_t K, int64_t n,
                           int64_t row_lo, int64_t row_hi, int32_t n_labels, int32_t *counts, void *stream)
{
    if (!samples || !counts) return fail(PMDI_E_ARG, "null argument");
    if (S < 1 || K < 1 || n < 1 || row_lo < 0 || row_hi > n || row_lo > row_hi)
        return fail(PMDI_E_ARG, "S=%lld K=%d n=%lld rows [%lld, %lld): out of range", (long long)S, K, (long long)n,
                    (long long)row_lo, (long long)row_hi);
    if (S > 2147483647LL) return fail(PMDI_E_ARG, "S=%lld samples overflow the int32 counts", (long long)S);
    if ((n + 63) / 64 > 2147483647LL || (row_hi - row_lo + 63) / 64 > 65535 || K > 65535)
        return fail(PMDI_E_ARG, "grid too large: split the rows into blocks of at most 4194240");
    HIP_TRY(hipSetDevice(device));
    if (n_labels < 0 || n_labels > 256) return fail(PMDI_E_ARG, "n_labels=%d outside 0..256", n_labels);
    // labels known to be < 64: one-hot int8 GEMM on the matrix cores; otherwise byte compares on the vector ALUs
    const bool mfma = n_labels >= 1 && n_labels <= 64 && (row_hi - row_lo + 127) / 128 <= 65535;
    hipError_t e = mfma ? pmdi_launch_psm_counts_mfma(samples, S, K, n, row_lo, row_hi, n_labels, counts, (hipStream_t)stream)
                        : pmdi_launch_psm_counts(samples, S, K, n, row_lo, row_hi, counts, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PMDI_E_DEVICE, "psm-count launch: %s", hipGetErrorString(e));
    return PMDI_OK;
}

int pmdi_label_counts_device(pmdi_handle *h, const int32_t *s, int32_t *counts, void *stream)
{
    if (!h || !s || !counts) return fail(PMDI_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipError_t e = pmdi_launch_label_counts(s, counts, h->cfg.n_chains * h->cfg.K, h->cfg.n, h->cfg.N, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PMDI_E_DEVICE, "label-count launch: %s", hipGetErrorString(e));
    return PMDI_OK;
}

int pmdi_sweep(pmdi_handle *h, int64_t iter, const int64_t *s_in, const int64_t *order_obs, int64_t n1,
               const double *Pi, const double *Phi, const uint8_t *feature_flag, double lw_init, int64_t *s_out,
               double *logweight, int64_t *p_star, pmdi_sweep_stats *stats, double *trace)
{
    if (!h || !s_in || !order_obs || !Pi || !Phi) return fail(PMDI_E_ARG, "null argument");
    const int K = h->cfg.K, N = h->cfg.N, P = h->cfg.P, C = h->cfg.n_chains;
    const long long n = h->cfg.n;
    if (n1 < 1 || n1 > n) return fail(PMDI_E_ARG, "n1=%lld outside 1..n (rho*n < 1 is undefined in the reference)", (long long)n1);
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<int> s32((size_t)C * K * n), o32((size_t)C * n);
    for (size_t i = 0; i < s32.size(); ++i) {
        const long long v = s_in[i];
        if (v < 1 || v > N) return fail(PMDI_E_DATA, "s_in[%zu]=%lld outside 1..N", i, v);
        s32[i] = (int)(v - 1);
    }
    for (size_t i = 0; i < o32.size(); ++i) {
        const long long v = order_obs[i];
        if (v < 1 || v > n) return fail(PMDI_E_DATA, "order_obs[%zu]=%lld outside 1..n", i, v);
        o32[i] = (int)(v - 1);
    }
    std::vector<double> lphi((size_t)C * h->npairs);
    for (size_t i = 0; i < lphi.size(); ++i) lphi[i] = log(1.0 + Phi[i]);   // src/misc.jl:53
    int rc;
    const long long ns = n - n1 + 1;
    if ((rc = h->d_s_in.ensure(s32.size() * 4)) || (rc = h->d_order.ensure(o32.size() * 4)) ||
        (rc = h->d_Pi.ensure((size_t)C * K * N * 8)) || (rc = h->d_logphi.ensure(lphi.size() * 8)) ||
        (rc = h->d_s_out.ensure(s32.size() * 4)) || (rc = h->d_lw.ensure((size_t)C * P * 8)))
        return rc;
    if (feature_flag && (rc = h->d_flags.ensure((size_t)C * h->sumD))) return rc;
    if (trace && (rc = h->d_trace.ensure((size_t)C * ns * (2 + 2 * K) * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_s_in.p, s32.data(), s32.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_order.p, o32.data(), o32.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_Pi.p, Pi, (size_t)C * K * N * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_logphi.p, lphi.data(), lphi.size() * 8, hipMemcpyHostToDevice, h->stream));
    if (feature_flag) HIP_TRY(hipMemcpyAsync(h->d_flags.p, feature_flag, (size_t)C * h->sumD, hipMemcpyHostToDevice, h->stream));

    SweepArgs a;
    fill_sweep_common(h, a);
    a.iter = (unsigned)iter; a.n1 = n1; a.lw_init = lw_init;
    a.s_in = (const int *)h->d_s_in.p; a.order = (const int *)h->d_order.p; a.Pi = (const double *)h->d_Pi.p;
    a.logphi = (const double *)h->d_logphi.p; a.flags = feature_flag ? (const unsigned char *)h->d_flags.p : nullptr;
    a.s_out = (int *)h->d_s_out.p; a.lw_out = (double *)h->d_lw.p; a.pstar = (int *)h->d_pstar.p;
    a.stats = (long long *)h->d_stats.p; a.err = (int *)h->d_err.p;
    a.trace = trace ? (double *)h->d_trace.p : nullptr; a.trace_on = trace ? 1 : 0;
    a.uscratch = (double *)h->d_usc.p; a.partstar = (int *)h->d_partstar.p; a.kstate = (int *)h->d_kstate.p;
    if ((rc = launch_sweep_groups(h, a, h->stream))) return rc;
    std::vector<int> so(s32.size()), ps(C), er(C);
    std::vector<long long> stv((size_t)C * 8);
    HIP_TRY(hipMemcpyAsync(so.data(), h->d_s_out.p, so.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ps.data(), h->d_pstar.p, (size_t)C * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(er.data(), h->d_err.p, (size_t)C * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(stv.data(), h->d_stats.p, stv.size() * 8, hipMemcpyDeviceToHost, h->stream));
    if (logweight) HIP_TRY(hipMemcpyAsync(logweight, h->d_lw.p, (size_t)C * P * 8, hipMemcpyDeviceToHost, h->stream));
    if (trace) HIP_TRY(hipMemcpyAsync(trace, h->d_trace.p, (size_t)C * ns * (2 + 2 * K) * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->swept = true; h->last_n1 = n1;
    for (int c = 0; c < C; ++c)
        if (er[c] != 0)
            return fail(er[c] == PMDI_E_POOL ? PMDI_E_POOL : PMDI_E_STATE,
                        er[c] == PMDI_E_POOL ? "chain %d: cluster pool capacity %lld exceeded" : "chain %d: kernel reported error", c, h->cap);
    if (s_out) for (size_t i = 0; i < so.size(); ++i) s_out[i] = (int64_t)so[i] + 1;
    if (p_star) for (int c = 0; c < C; ++c) p_star[c] = (int64_t)ps[c] + 1;
    if (stats)
        for (int c = 0; c < C; ++c) {
            memset(&stats[c], 0, sizeof(pmdi_sweep_stats));
            stats[c].n_operations = stv[(size_t)c * 8 + ST_NOPS];
            stats[c].n_resamples = stv[(size_t)c * 8 + ST_NRESAMPLE];
            stats[c].n_clones = stv[(size_t)c * 8 + ST_NCLONES];
            stats[c].max_id = stv[(size_t)c * 8 + ST_MAXID];
            stats[c].sum_classes = stv[(size_t)c * 8 + ST_SUMCLASSES];
            stats[c].steps_fast = stv[(size_t)c * 8 + 5];
            stats[c].steps_converted = stv[(size_t)c * 8 + 6];
            stats[c].steps_fallback = stv[(size_t)c * 8 + 7];
        }
    return PMDI_OK;
}

int pmdi_export_state(pmdi_handle *h, int32_t chain, int64_t *particle, int64_t *counts, int64_t *cluster_n,
                      int64_t *max_id)
{
    if (!h) return fail(PMDI_E_ARG, "null handle");
    if (!h->swept) return fail(PMDI_E_STATE, "no sweep has run on this handle");
    if (chain < 0 || chain >= h->cfg.n_chains) return fail(PMDI_E_ARG, "chain out of range");
    const int K = h->cfg.K, N = h->cfg.N, P = h->cfg.P;
    const long long cap = h->cap;
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<int> ks((size_t)PMDI_KMAX_I * 2);
    HIP_TRY(hipMemcpy(ks.data(), (char *)h->d_kstate.p + (size_t)chain * PMDI_KMAX_I * 2 * 4, ks.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int> tmp((size_t)N * P), tc((size_t)cap + 1), col((size_t)P);
    for (int k = 0; k < K; ++k) {
        const DsetDev &d = h->ds[k];
        const char *base = d.arena + (size_t)chain * d.stride;
        const int cur = ks[(size_t)k * 2 + 1];
        if (particle) {
            // the device keeps the distinct columns of particle[:, :, k] and a column index per particle
            HIP_TRY(hipMemcpy(tmp.data(), base + d.o_particle[cur], tmp.size() * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(col.data(), base + d.o_col, col.size() * 4, hipMemcpyDeviceToHost));
            for (int p = 0; p < P; ++p) {                             // particle[n, p, k], column-major
                if (col[p] < 0 || col[p] >= P) return fail(PMDI_E_STATE, "corrupt column index %d of particle %d", col[p], p);
                for (int nn = 0; nn < N; ++nn) particle[((size_t)k * P + p) * N + nn] = tmp[(size_t)col[p] * N + nn];
            }
        }
        if (counts) {
            HIP_TRY(hipMemcpy(tc.data(), base + d.o_counts, tc.size() * 4, hipMemcpyDeviceToHost));
            for (long long id = 1; id <= cap; ++id) counts[(size_t)k * cap + (id - 1)] = tc[id];
        }
        if (cluster_n) {
            HIP_TRY(hipMemcpy(tc.data(), base + d.o_cn, tc.size() * 4, hipMemcpyDeviceToHost));
            for (long long id = 1; id <= cap; ++id) cluster_n[(size_t)k * cap + (id - 1)] = tc[id];
        }
        if (max_id) max_id[k] = ks[(size_t)k * 2];
    }
    return PMDI_OK;
}

int pmdi_feature_select(pmdi_handle *h, int64_t iter, const int64_t *s_traj, uint8_t *feature_flag, double *feature_prob)
{
    if (!h || !s_traj) return fail(PMDI_E_ARG, "null argument");
    const int K = h->cfg.K, N = h->cfg.N, C = h->cfg.n_chains;
    const long long n = h->cfg.n;
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<int> t32((size_t)C * K * n);
    for (size_t i = 0; i < t32.size(); ++i) {
        const long long v = s_traj[i];
        if (v < 1 || v > N) return fail(PMDI_E_DATA, "s_traj[%zu]=%lld outside 1..N", i, v);
        t32[i] = (int)(v - 1);
    }
    HIP_TRY(hipMemcpyAsync(h->d_traj.p, t32.data(), t32.size() * 4, hipMemcpyHostToDevice, h->stream));
    FeatSelArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.K = K; fa.N = N; fa.sumD = h->sumD; fa.n = n; fa.iter = (unsigned)iter; fa.seed = h->cfg.seed;
    for (int k = 0; k < K; ++k) fa.ds[k] = h->ds[k];
    fa.traj = (const int *)h->d_traj.p; fa.lm = (double *)h->d_lm.p; fa.firstpos = (int *)h->d_firstpos.p;
    fa.fnull = (const double *)h->d_fnull.p; fa.flags_out = (unsigned char *)h->d_fflags.p; fa.prob_out = (double *)h->d_fprob.p;
    hipError_t e = pmdi_launch_featsel(fa, C, h->stream);
    if (e != hipSuccess) return fail(PMDI_E_DEVICE, "feature-select launch: %s", hipGetErrorString(e));
    if (feature_flag) HIP_TRY(hipMemcpyAsync(feature_flag, h->d_fflags.p, (size_t)C * h->sumD, hipMemcpyDeviceToHost, h->stream));
    if (feature_prob) HIP_TRY(hipMemcpyAsync(feature_prob, h->d_fprob.p, (size_t)C * h->sumD * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PMDI_OK;
}

// ---- stand-alone cluster batches ------------------------------------------
int pmdi_clusters_new(pmdi_handle *h, int32_t k, int32_t B, pmdi_cluster_batch **out)
{
    if (!h || !out) return fail(PMDI_E_ARG, "null argument");
    *out = nullptr;
    if (k < 0 || k >= h->cfg.K || B < 1) return fail(PMDI_E_ARG, "bad dataset index or batch size");
    HIP_TRY(hipSetDevice(h->cfg.device));
    pmdi_cluster_batch *cb = new (std::nothrow) pmdi_cluster_batch();
    if (!cb) return fail(PMDI_E_MEMORY, "out of host memory");
    cb->h = h; cb->k = k; cb->B = B;
    cb->d = h->ds[k];
    h->children += 1;
    cb->d.stride = layout_arena(cb->d, h->cfg.N, h->cfg.P, B, 0, false);
    hipError_t e = hipMalloc(&cb->arena, cb->d.stride);
    if (e != hipSuccess) { h->children -= 1; delete cb; return fail(PMDI_E_MEMORY, "hipMalloc: %s", hipGetErrorString(e)); }
    cb->d.arena = (char *)cb->arena;
    (void)hipMemset(cb->arena, 0, cb->d.stride);
    if (cb->d.kind == K_GAUSSIAN) {     // GaussianCluster(dataFile): mu=Sigma=0, lambda=1, beta=0.5 (gaussian_cluster.jl:17-21)
        std::vector<double> ml((size_t)(B + 1) * cb->d.D * 2), sb(ml.size());
        for (size_t i = 0; i < ml.size(); i += 2) { ml[i] = 0.0; ml[i + 1] = 1.0; sb[i] = 0.0; sb[i + 1] = 0.5; }
        (void)hipMemcpy(cb->d.arena + cb->d.o_ml, ml.data(), ml.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(cb->d.arena + cb->d.o_sb, sb.data(), sb.size() * 8, hipMemcpyHostToDevice);
    }
    *out = cb;
    return PMDI_OK;
}

int pmdi_clusters_free(pmdi_cluster_batch *cb)
{
    if (!cb) return PMDI_OK;
    (void)hipSetDevice(cb->h->cfg.device);
    if (cb->arena) (void)hipFree(cb->arena);
    cb->d_rows.release(); cb->d_flags.release(); cb->d_out.release();
    if (cb->h->children > 0) cb->h->children -= 1;
    delete cb;
    return PMDI_OK;
}

static int batch_args(pmdi_cluster_batch *cb, const int64_t *rows, const uint8_t *flag, ClusterBatchArgs &a)
{
    const long long n = cb->h->cfg.n;
    int rc;
    memset(&a, 0, sizeof(a));
    a.ds = cb->d; a.B = cb->B;
    if (rows) {
        std::vector<int> r32(cb->B);
        for (int b = 0; b < cb->B; ++b) {
            if (rows[b] < 1 || rows[b] > n) return fail(PMDI_E_ARG, "row %lld outside 1..n", (long long)rows[b]);
            r32[b] = (int)(rows[b] - 1);
        }
        if ((rc = cb->d_rows.ensure((size_t)cb->B * 4))) return rc;
        HIP_TRY(hipMemcpy(cb->d_rows.p, r32.data(), (size_t)cb->B * 4, hipMemcpyHostToDevice));
        a.rows = (const int *)cb->d_rows.p;
    }
    if (flag) {
        if ((rc = cb->d_flags.ensure((size_t)cb->d.D))) return rc;
        HIP_TRY(hipMemcpy(cb->d_flags.p, flag, (size_t)cb->d.D, hipMemcpyHostToDevice));
        a.flags = (const unsigned char *)cb->d_flags.p;
    }
    return 0;
}

int pmdi_cluster_add(pmdi_cluster_batch *cb, const int64_t *rows, const uint8_t *feature_flag)
{
    if (!cb || !rows) return fail(PMDI_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(cb->h->cfg.device));
    ClusterBatchArgs a;
    int rc = batch_args(cb, rows, feature_flag, a);
    if (rc) return rc;
    hipError_t e = pmdi_launch_cluster_add(a, cb->h->stream);
    if (e != hipSuccess) return fail(PMDI_E_DEVICE, "cluster_add launch: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamSynchronize(cb->h->stream));
    return PMDI_OK;
}

int pmdi_calc_logprob(pmdi_cluster_batch *cb, const int64_t *obs_rows, const uint8_t *feature_flag, double *out)
{
    if (!cb || !obs_rows || !out) return fail(PMDI_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(cb->h->cfg.device));
    ClusterBatchArgs a;
    int rc = batch_args(cb, obs_rows, feature_flag, a);
    if (rc) return rc;
    if ((rc = cb->d_out.ensure((size_t)cb->B * 8))) return rc;
    a.out = (double *)cb->d_out.p;
    hipError_t e = pmdi_launch_cluster_logprob(a, cb->h->stream);
    if (e != hipSuccess) return fail(PMDI_E_DEVICE, "calc_logprob launch: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(out, cb->d_out.p, (size_t)cb->B * 8, hipMemcpyDeviceToHost, cb->h->stream));
    HIP_TRY(hipStreamSynchronize(cb->h->stream));
    return PMDI_OK;
}

int pmdi_calc_logmarginal(pmdi_cluster_batch *cb, double *out)
{
    if (!cb || !out) return fail(PMDI_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(cb->h->cfg.device));
    ClusterBatchArgs a;
    int rc = batch_args(cb, nullptr, nullptr, a);
    if (rc) return rc;
    const size_t bytes = (size_t)cb->B * cb->d.D * 8;
    if ((rc = cb->d_out.ensure(bytes))) return rc;
    a.out = (double *)cb->d_out.p;
    hipError_t e = pmdi_launch_cluster_logmarginal(a, cb->h->stream);
    if (e != hipSuccess) return fail(PMDI_E_DEVICE, "calc_logmarginal launch: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(out, cb->d_out.p, bytes, hipMemcpyDeviceToHost, cb->h->stream));
    HIP_TRY(hipStreamSynchronize(cb->h->stream));
    return PMDI_OK;
}

int pmdi_cluster_stats(pmdi_cluster_batch *cb, double *out, int64_t *stride)
{
    if (!cb || !stride) return fail(PMDI_E_ARG, "null argument");
    const DsetDev &d = cb->d;
    const int D = d.D, B = cb->B;
    const int64_t st = 1 + (d.kind == K_GAUSSIAN ? 4 * (int64_t)D : d.kind == K_CATEGORICAL ? (int64_t)D * d.L : (int64_t)D);
    *stride = st;
    if (!out) return PMDI_OK;
    HIP_TRY(hipSetDevice(cb->h->cfg.device));
    HIP_TRY(hipStreamSynchronize(cb->h->stream));
    std::vector<int> cn((size_t)B + 1);
    HIP_TRY(hipMemcpy(cn.data(), d.arena + d.o_cn, cn.size() * 4, hipMemcpyDeviceToHost));
    if (d.kind == K_GAUSSIAN) {
        std::vector<double> ml((size_t)(B + 1) * D * 2), sb(ml.size());
        HIP_TRY(hipMemcpy(ml.data(), d.arena + d.o_ml, ml.size() * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(sb.data(), d.arena + d.o_sb, sb.size() * 8, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            double *o = out + (size_t)b * st;
            o[0] = cn[b + 1];
            for (int q = 0; q < D; ++q) {
                const size_t e = ((size_t)(b + 1) * D + q) * 2;
                o[1 + q] = ml[e]; o[1 + D + q] = sb[e]; o[1 + 2 * D + q] = ml[e + 1]; o[1 + 3 * D + q] = sb[e + 1];
            }
        }
    } else if (d.kind == K_CATEGORICAL) {
        std::vector<int> c((size_t)(B + 1) * D * d.L);
        HIP_TRY(hipMemcpy(c.data(), d.arena + d.o_cnt, c.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            double *o = out + (size_t)b * st;
            o[0] = cn[b + 1];
            for (int q = 0; q < D; ++q)
                for (int l = 0; l < d.L; ++l) o[1 + (size_t)q * d.L + l] = c[((size_t)(b + 1) * D + q) * d.L + l];
        }
    } else {
        std::vector<long long> s((size_t)(B + 1) * D);
        HIP_TRY(hipMemcpy(s.data(), d.arena + d.o_nbs, s.size() * 8, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            double *o = out + (size_t)b * st;
            o[0] = cn[b + 1];
            for (int q = 0; q < D; ++q) o[1 + q] = (double)s[(size_t)(b + 1) * D + q];
        }
    }
    return PMDI_OK;
}

}  // extern "C"

// ---- device-resident Gibbs chains ---------------------------------------------------------------
namespace {
template <class Tp>
int galloc(pmdi_gibbs *g, Tp **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = count * sizeof(Tp);
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) { *p = nullptr; return fail(PMDI_E_MEMORY, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    g->owned.push_back(q);
    *p = (Tp *)q;
    return 0;
}
}  // namespace

extern "C" {

int pmdi_gibbs_destroy(pmdi_gibbs *g)
{
    if (!g) return PMDI_OK;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();
    for (void *p : g->owned) (void)hipFree(p);
    if (g->h && g->h->children > 0) g->h->children -= 1;      // (the handle outlives its children: pmdi_destroy refuses otherwise)
    delete g;
    return PMDI_OK;
}

int pmdi_gibbs_create(pmdi_handle *h, double rho, int32_t feature_select, pmdi_gibbs **out)
{
    if (!h || !out) return fail(PMDI_E_ARG, "null argument");
    *out = nullptr;
    if (!(rho < 1.0 && rho > 0.0)) return fail(PMDI_E_ARG, "rho must be between 0 and 1");          // src/pmdi.jl:53
    const int K = h->cfg.K, N = h->cfg.N, P = h->cfg.P, C = h->cfg.n_chains;
    const long long n = h->cfg.n;
    const long long n1 = (long long)floor(rho * (double)n);                                           // :161
    if (n1 < 1) return fail(PMDI_E_ARG, "floor(rho*n) = %lld < 1 (the reference indexes order_obs[0], SURVEY Q8)", n1);
    HIP_TRY(hipSetDevice(h->cfg.device));
    pmdi_gibbs *g = new (std::nothrow) pmdi_gibbs();
    if (!g) return fail(PMDI_E_MEMORY, "out of host memory");
    g->h = h; g->device = h->cfg.device; g->n1 = n1; g->feature_select = feature_select ? 1 : 0;
    h->children += 1;
    GibbsArgs &a = g->ga;
    a.K = K; a.N = N; a.npairs = h->npairs; a.n_chains = C; a.n = n; a.iter = 0; a.seed = h->cfg.seed;
    int rc = 0;
    double *lg = nullptr;
    const size_t tabsz = (size_t)K * K * N * N;
    if ((rc = galloc(g, &a.M, (size_t)C * K)) || (rc = galloc(g, &a.gamma, (size_t)C * K * N)) ||
        (rc = galloc(g, &a.gamma0, (size_t)C * K * N)) || (rc = galloc(g, &a.Phi, (size_t)C * h->npairs)) ||
        (rc = galloc(g, &a.vZ, (size_t)C * 2)) || (rc = galloc(g, &a.s, (size_t)C * K * n)) ||
        (rc = galloc(g, &g->s_next, (size_t)C * K * n)) || (rc = galloc(g, &a.order, (size_t)C * n)) ||
        (rc = galloc(g, &a.Pi, (size_t)C * K * N)) || (rc = galloc(g, &a.logphi, (size_t)C * h->npairs)) ||
        (rc = galloc(g, &a.wscr, (size_t)C * (n + 1))) || (rc = galloc(g, &lg, (size_t)n + 3)) ||
        (rc = galloc(g, &g->flags, (size_t)C * h->sumD)) || (rc = galloc(g, &g->fprob, (size_t)C * h->sumD)) ||
        (rc = galloc(g, &g->lw, (size_t)C * P)) || (rc = galloc(g, &g->pstar, (size_t)C)) ||
        (rc = galloc(g, &g->stats, (size_t)C * 8)) || (rc = galloc(g, &g->err, (size_t)C))) {
        pmdi_gibbs_destroy(g);
        return rc;
    }
    a.ctab_lds = (tabsz * 4 <= 96 * 1024) ? 1 : 0;
    a.order_lds = ((size_t)n * 4 <= 96 * 1024) ? 1 : 0;
    a.ctab = nullptr;
    if (!a.ctab_lds && K > 1 && (rc = galloc(g, &a.ctab, (size_t)C * tabsz))) { pmdi_gibbs_destroy(g); return rc; }
    {
        std::vector<double> t((size_t)n + 3);
        for (size_t m = 0; m < t.size(); ++m) t[m] = lgamma((double)m);          // t[0] = inf is never read
        if (hipMemcpy(lg, t.data(), t.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { pmdi_gibbs_destroy(g); return fail(PMDI_E_DEVICE, "memcpy failed"); }
        a.lgtab = lg;
    }
    // featureFlag (src/pmdi.jl:106-110): rand(Bool) per feature when feature selection is on, else all true
    {
        std::vector<unsigned char> fl((size_t)C * h->sumD, 1);
        if (g->feature_select)
            for (int c = 0; c < C; ++c)
                for (int k = 0; k < K; ++k)
                    for (int q = 0; q < h->ds[k].D; ++q)
                        fl[(size_t)c * h->sumD + h->ds[k].flag_off + q] =
                            host_uniform01(h->cfg.seed + (unsigned long long)c, 0, 0, (unsigned)k, (unsigned)q, SITE_INIT_FLAGS) < 0.5 ? 1 : 0;
        if (hipMemcpy(g->flags, fl.data(), fl.size(), hipMemcpyHostToDevice) != hipSuccess) { pmdi_gibbs_destroy(g); return fail(PMDI_E_DEVICE, "memcpy failed"); }
    }
    if (hipMemset(g->err, 0, (size_t)C * 4) != hipSuccess || hipMemset(g->stats, 0, (size_t)C * 64) != hipSuccess ||
        hipMemset(g->s_next, 0, (size_t)C * K * n * 4) != hipSuccess) {
        pmdi_gibbs_destroy(g);
        return fail(PMDI_E_DEVICE, "memset failed");
    }
    hipError_t e = pmdi_launch_gibbs_init(a, h->stream);
    if (e != hipSuccess) { pmdi_gibbs_destroy(g); return fail(PMDI_E_DEVICE, "gibbs init launch: %s", hipGetErrorString(e)); }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { pmdi_gibbs_destroy(g); return fail(PMDI_E_DEVICE, "gibbs init kernel failed"); }
    *out = g;
    return PMDI_OK;
}

int pmdi_gibbs_step(pmdi_gibbs *g, int32_t what, void *stream)
{
    if (!g) return fail(PMDI_E_ARG, "null argument");
    pmdi_handle *h = g->h;
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    GibbsArgs a = g->ga;
    hipError_t e;
    switch (what) {
    case PMDI_STEP_BEGIN:
        g->iter += 1;
        return PMDI_OK;
    case PMDI_STEP_HYPERS:
        a.iter = (unsigned)g->iter;
        e = pmdi_launch_hypers(a, 1, st);
        if (e != hipSuccess) return fail(PMDI_E_DEVICE, "hypers launch: %s", hipGetErrorString(e));
        return PMDI_OK;
    case PMDI_STEP_SWEEP: {
        // the first error of a chain sticks (a later, successful sweep does not reset it): pmdi_gibbs_results reports it,
        // pmdi_gibbs_set of that chain clears it.  A failed chain keeps its allocations (the kernel copies s_in to s_out).
        h->err_keep = 1;
        const int rc = pmdi_sweep_device(h, g->iter, a.s, a.order, g->n1, a.Pi, a.logphi, g->flags, g->iter == 1 ? 0.0 : 1.0,
                                         g->s_next, g->lw, g->pstar, (int64_t *)g->stats, g->err, stream);
        h->err_keep = 0;
        if (rc) return rc;
        std::swap(g->ga.s, g->s_next);               // s = sstar[p_star, :, :] (src/pmdi.jl:373)
        return PMDI_OK;
    }
    case PMDI_STEP_FEATSEL: {
        FeatSelArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.K = h->cfg.K; fa.N = h->cfg.N; fa.sumD = h->sumD; fa.n = h->cfg.n; fa.iter = (unsigned)g->iter; fa.seed = h->cfg.seed;
        for (int k = 0; k < h->cfg.K; ++k) fa.ds[k] = h->ds[k];
        fa.traj = a.s; fa.lm = (double *)h->d_lm.p; fa.firstpos = (int *)h->d_firstpos.p;
        fa.fnull = (const double *)h->d_fnull.p; fa.flags_out = g->flags; fa.prob_out = g->fprob;
        e = pmdi_launch_featsel(fa, h->cfg.n_chains, st);
        if (e != hipSuccess) return fail(PMDI_E_DEVICE, "feature-select launch: %s", hipGetErrorString(e));
        return PMDI_OK;
    }
    case PMDI_STEP_ALIGN:
        a.iter = (unsigned)g->iter;
        e = pmdi_launch_align(a, st);
        if (e != hipSuccess) return fail(PMDI_E_DEVICE, "align launch: %s", hipGetErrorString(e));
        return PMDI_OK;
    default:
        return fail(PMDI_E_ARG, "unknown step %d", what);
    }
}

int pmdi_gibbs_iterate(pmdi_gibbs *g, int64_t n_iter, uint8_t *samples, void *stream)
{
    if (!g || n_iter < 0) return fail(PMDI_E_ARG, "bad argument");
    const pmdi_handle *h = g->h;
    const long long per = (long long)h->cfg.n_chains * h->cfg.K * h->cfg.n;
    for (int64_t t = 0; t < n_iter; ++t) {
        int rc;
        if ((rc = pmdi_gibbs_step(g, PMDI_STEP_BEGIN, stream)) || (rc = pmdi_gibbs_step(g, PMDI_STEP_HYPERS, stream)) ||
            (rc = pmdi_gibbs_step(g, PMDI_STEP_SWEEP, stream)))
            return rc;
        if (g->feature_select && (rc = pmdi_gibbs_step(g, PMDI_STEP_FEATSEL, stream))) return rc;
        if ((rc = pmdi_gibbs_step(g, PMDI_STEP_ALIGN, stream))) return rc;
        if (samples) {
            hipError_t e = pmdi_launch_pack_samples(g->ga.s, samples + (size_t)t * per, per, (hipStream_t)stream);
            if (e != hipSuccess) return fail(PMDI_E_DEVICE, "pack-samples launch: %s", hipGetErrorString(e));
        }
    }
    return PMDI_OK;
}

int64_t pmdi_gibbs_iterations(const pmdi_gibbs *g) { return g ? g->iter : 0; }

int pmdi_gibbs_pack_samples(pmdi_gibbs *g, uint8_t *out, void *stream)
{
    if (!g || !out) return fail(PMDI_E_ARG, "null argument");
    const pmdi_handle *h = g->h;
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipError_t e = pmdi_launch_pack_samples(g->ga.s, out, (long long)h->cfg.n_chains * h->cfg.K * h->cfg.n, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PMDI_E_DEVICE, "pack-samples launch: %s", hipGetErrorString(e));
    return PMDI_OK;
}

int pmdi_gibbs_device_view(pmdi_gibbs *g, pmdi_gibbs_view *v)
{
    if (!g || !v) return fail(PMDI_E_ARG, "null argument");
    v->M = g->ga.M; v->gamma = g->ga.gamma; v->gamma0 = g->ga.gamma0; v->Phi = g->ga.Phi; v->vZ = g->ga.vZ;
    v->s = g->ga.s; v->order_obs = g->ga.order; v->Pi = g->ga.Pi; v->log1p_phi = g->ga.logphi;
    v->feature_flag = g->flags; v->feature_prob = g->fprob; v->logweight = g->lw; v->p_star = g->pstar;
    v->stats = (int64_t *)g->stats; v->err = g->err; v->n1 = g->n1;
    return PMDI_OK;
}

int pmdi_gibbs_get(pmdi_gibbs *g, int32_t chain, double *M, double *gamma, double *gamma0, double *Phi, double *vZ,
                   int64_t *s, int64_t *order_obs, uint8_t *feature_flag)
{
    if (!g) return fail(PMDI_E_ARG, "null argument");
    const pmdi_handle *h = g->h;
    const int K = h->cfg.K, N = h->cfg.N;
    const long long n = h->cfg.n;
    if (chain < 0 || chain >= h->cfg.n_chains) return fail(PMDI_E_ARG, "chain out of range");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    const GibbsArgs &a = g->ga;
    if (M) HIP_TRY(hipMemcpy(M, a.M + (size_t)chain * K, (size_t)K * 8, hipMemcpyDeviceToHost));
    if (gamma) HIP_TRY(hipMemcpy(gamma, a.gamma + (size_t)chain * K * N, (size_t)K * N * 8, hipMemcpyDeviceToHost));    // N x K column-major
    if (gamma0) HIP_TRY(hipMemcpy(gamma0, a.gamma0 + (size_t)chain * K * N, (size_t)K * N * 8, hipMemcpyDeviceToHost));
    if (Phi) HIP_TRY(hipMemcpy(Phi, a.Phi + (size_t)chain * h->npairs, (size_t)h->npairs * 8, hipMemcpyDeviceToHost));
    if (vZ) HIP_TRY(hipMemcpy(vZ, a.vZ + (size_t)chain * 2, 16, hipMemcpyDeviceToHost));
    if (s) {
        std::vector<int> t((size_t)K * n);
        HIP_TRY(hipMemcpy(t.data(), a.s + (size_t)chain * K * n, t.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < t.size(); ++i) s[i] = (int64_t)t[i] + 1;         // n x K column-major, labels 1..N
    }
    if (order_obs) {
        std::vector<int> t((size_t)n);
        HIP_TRY(hipMemcpy(t.data(), a.order + (size_t)chain * n, t.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < t.size(); ++i) order_obs[i] = (int64_t)t[i] + 1;
    }
    if (feature_flag) HIP_TRY(hipMemcpy(feature_flag, g->flags + (size_t)chain * h->sumD, (size_t)h->sumD, hipMemcpyDeviceToHost));
    return PMDI_OK;
}

int pmdi_gibbs_set(pmdi_gibbs *g, int32_t chain, const double *M, const double *gamma, const double *gamma0, const double *Phi,
                   const double *vZ, const int64_t *s, const int64_t *order_obs, const uint8_t *feature_flag)
{
    if (!g) return fail(PMDI_E_ARG, "null argument");
    const pmdi_handle *h = g->h;
    const int K = h->cfg.K, N = h->cfg.N;
    const long long n = h->cfg.n;
    if (chain < 0 || chain >= h->cfg.n_chains) return fail(PMDI_E_ARG, "chain out of range");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    const GibbsArgs &a = g->ga;
    if (M) HIP_TRY(hipMemcpy(a.M + (size_t)chain * K, M, (size_t)K * 8, hipMemcpyHostToDevice));
    if (gamma) HIP_TRY(hipMemcpy(a.gamma + (size_t)chain * K * N, gamma, (size_t)K * N * 8, hipMemcpyHostToDevice));
    if (gamma0) HIP_TRY(hipMemcpy(a.gamma0 + (size_t)chain * K * N, gamma0, (size_t)K * N * 8, hipMemcpyHostToDevice));
    if (Phi) HIP_TRY(hipMemcpy(a.Phi + (size_t)chain * h->npairs, Phi, (size_t)h->npairs * 8, hipMemcpyHostToDevice));
    if (vZ) HIP_TRY(hipMemcpy(a.vZ + (size_t)chain * 2, vZ, 16, hipMemcpyHostToDevice));
    if (s) {
        std::vector<int> t((size_t)K * n);
        for (size_t i = 0; i < t.size(); ++i) {
            if (s[i] < 1 || s[i] > N) return fail(PMDI_E_DATA, "s[%zu]=%lld outside 1..N", i, (long long)s[i]);
            t[i] = (int)(s[i] - 1);
        }
        HIP_TRY(hipMemcpy(a.s + (size_t)chain * K * n, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    if (order_obs) {
        std::vector<int> t((size_t)n);
        for (size_t i = 0; i < t.size(); ++i) {
            if (order_obs[i] < 1 || order_obs[i] > n) return fail(PMDI_E_DATA, "order_obs[%zu]=%lld outside 1..n", i, (long long)order_obs[i]);
            t[i] = (int)(order_obs[i] - 1);
        }
        HIP_TRY(hipMemcpy(a.order + (size_t)chain * n, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    if (feature_flag) HIP_TRY(hipMemcpy(g->flags + (size_t)chain * h->sumD, feature_flag, (size_t)h->sumD, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(g->err + chain, 0, 4));                 // a chain given a new state starts without its sticky error
    return PMDI_OK;
}

int pmdi_gibbs_results(pmdi_gibbs *g, int64_t *stats, int32_t *err, int64_t *p_star, double *logweight)
{
    if (!g) return fail(PMDI_E_ARG, "null argument");
    const pmdi_handle *h = g->h;
    const int C = h->cfg.n_chains;
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    if (stats) HIP_TRY(hipMemcpy(stats, g->stats, (size_t)C * 64, hipMemcpyDeviceToHost));
    std::vector<int> er(C);
    HIP_TRY(hipMemcpy(er.data(), g->err, (size_t)C * 4, hipMemcpyDeviceToHost));
    if (err) memcpy(err, er.data(), (size_t)C * 4);
    if (p_star) {
        std::vector<int> ps(C);
        HIP_TRY(hipMemcpy(ps.data(), g->pstar, (size_t)C * 4, hipMemcpyDeviceToHost));
        for (int c = 0; c < C; ++c) p_star[c] = (int64_t)ps[c] + 1;
    }
    if (logweight) HIP_TRY(hipMemcpy(logweight, g->lw, (size_t)C * h->cfg.P * 8, hipMemcpyDeviceToHost));
    for (int c = 0; c < C; ++c)
        if (er[c] != 0)
            return fail(er[c] == PMDI_E_POOL ? PMDI_E_POOL : PMDI_E_STATE,
                        er[c] == PMDI_E_POOL ? "chain %d: cluster pool capacity %lld exceeded" : "chain %d: kernel reported error", c, h->cap);
    return PMDI_OK;
}

}  // extern "C"
