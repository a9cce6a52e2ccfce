// pmdi_kernels.hip -- gfx950 (CDNA4) kernels for ParticleMDI's conditional-SMC
// sweep.  Compile with -ffp-contract=off: the floating-point expression order
// below is the reference's and must not be contracted into FMAs.
//
// Design (DESIGN.md has the full account):
//   * one workgroup = one Gibbs chain, persistent over the whole sweep
//     (known-prefix build, n_s x K sequential steps, resampling, particle
//     pick) -- a kernel boundary per step (>= 1.5 us) would cost more than a
//     step's work;
//   * one particle per lane for everything that is per particle (allocation
//     draw, weight update, class/cluster bookkeeping);
//   * the reference's de-duplication is kept: predictive log-probabilities are
//     evaluated once per LIVE CLUSTER (lanes = cluster x feature), mutation
//     CDFs once per particle CLASS (lanes = class x label, wave shuffles),
//     sufficient statistics live in a copy-on-write pool;
//   * observation row, Pi, log-weights, per-feature terms and the class tables
//     are staged in LDS; wave shuffles + LDS for scans / ESS reductions.
//
// Reference lines are cited as file:line relative to /root/reference.
#include "pmdi_internal.h"

namespace {

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
    unsigned long long m = ((unsigned long long)(c0 >> 5) << 26) | (unsigned long long)(c1 >> 6);
    return (double)m * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------------------
struct KS {  // pointers of one (chain, dataset)
    int *part[2];
    int *pid, *sid, *newid, *counts, *ncop, *firstc, *cn, *clslead, *clsval, *dl;
    double *lp, *cdf;
    double2 *ml, *sb;
    int *cnt;
    long long *nbs;
    unsigned char *sstar;
};

__device__ __forceinline__ KS make_ks(const DsetDev &d, int chain)
{
    KS s;
    char *b = d.arena + (size_t)chain * d.stride;
    s.part[0] = (int *)(b + d.o_particle[0]);
    s.part[1] = (int *)(b + d.o_particle[1]);
    s.pid = (int *)(b + d.o_pid);
    s.sid = (int *)(b + d.o_sid);
    s.newid = (int *)(b + d.o_newid);
    s.counts = (int *)(b + d.o_counts);
    s.ncop = (int *)(b + d.o_ncop);
    s.firstc = (int *)(b + d.o_firstc);
    s.lp = (double *)(b + d.o_lp);
    s.cn = (int *)(b + d.o_cn);
    s.ml = (double2 *)(b + d.o_ml);
    s.sb = (double2 *)(b + d.o_sb);
    s.cnt = (int *)(b + d.o_cnt);
    s.nbs = (long long *)(b + d.o_nbs);
    s.sstar = (unsigned char *)(b + d.o_sstar);
    s.clslead = (int *)(b + d.o_clslead);
    s.clsval = (int *)(b + d.o_clsval);
    s.cdf = (double *)(b + d.o_cdf);
    s.dl = (int *)(b + d.o_dl);
    return s;
}

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

template <int T>
__device__ __forceinline__ double block_max(double v, double *scr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double t = __shfl_xor(v, off);
        v = (t > v) ? t : v;
    }
    if (lane == 0) scr[wave] = v;
    __syncthreads();
    double m = scr[0];
#pragma unroll
    for (int w = 1; w < T / 64; ++w) { double t = scr[w]; m = (t > m) ? t : m; }
    __syncthreads();
    return m;
}

template <int T>
__device__ __forceinline__ void block_sum2(double &a, double &b, double *scr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if (lane == 0) { scr[wave] = a; scr[16 + wave] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) { sa += scr[w]; sb += scr[16 + w]; }
    __syncthreads();
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
        int k0 = __shfl(key, l0);
        unsigned long long m = __ballot(valid && key == k0);
        if (lane == l0) { leader = true; count = __popcll(m); }
        active &= ~m;
    }
    return leader;
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

// the two per-feature terms of calc_logprob(::GaussianCluster): gaussian_cluster.jl:45-48
__device__ __forceinline__ void gauss_terms(double x, double n, double2 ml, double &ta, double &tb)
{
    ta = 0.5 * log(ml.y / (n + 1.0));
    const double d = x - ml.x;
    tb = (0.5 * n + 1.0) * log(1.0 + (1.0 / (n + 1.0)) * (d * d) * ml.y);
}

// calc_logprob(::NegBinomCluster) per-feature term: negbinom_cluster.jl:33-37;
// loggamma of integers comes from the host-built table LG[m] = lgamma(m)
__device__ __forceinline__ double negbin_term(const double *lg, long long n, long long x, long long S)
{
    return lg[1 + n + 1] + lg[1 + x + S] + lg[1 + n + 1 + S] - lg[1 + n + 1 + 1 + x + S] -
           lg[1 + n] - lg[1 + S];
}

// ---------------------------------------------------------------------------
struct Sh {  // LDS carve
    double *xs;        // [Dmax] observation row (doubles) / int view
    double *pis;       // [N]
    double *lw;        // [P]
    double *term;      // [terms_cap]
    unsigned long long *scan;  // [16]
    double *red;       // [32]
    int *lead_of;      // [P+1] class value -> min particle (INF between uses)
    int *slot_of;      // [P+1] class value -> class slot
    int *kmaxid, *kncls, *kcur;  // [KMAX]
    int *lab;          // [256*3] prefix scratch: first position, id, count
    unsigned char *fl; // [Dmax] feature flags of the current dataset
    unsigned char *news; // [K][P]
};

// Rebuild the class list of dataset k from pid[]: class slot r <-> (leader
// particle, class value); leader = lowest p of the class (the particle whose
// CDF the reference caches under fprob_dict, src/pmdi.jl:225-248).
// Precondition: lead_of == INF for every class value, threads synchronised.
template <int T>
__device__ __forceinline__ int rebuild_classes(const KS &s, const Sh &sh, int P)
{
    const int tid = threadIdx.x;
    for (int pb = 0; pb < P; pb += T) {
        const int p = pb + tid;
        const bool valid = p < P;
        const int cls = valid ? s.pid[p] : 0;
        int cnt;
        if (wave_group(cls, valid, cnt)) atomicMin(&sh.lead_of[cls], p);
    }
    __syncthreads();
    unsigned long long carry = 0;
    for (int pb = 0; pb < P; pb += T) {
        const int p = pb + tid;
        const bool valid = p < P;
        const int cls = valid ? s.pid[p] : 0;
        const bool isl = valid && sh.lead_of[cls] == p;
        unsigned long long tot;
        const unsigned long long ex = block_excl_scan<T>(isl ? 1ull : 0ull, tot, sh.scan) + carry;
        if (isl) {
            s.clslead[ex] = p;
            s.clsval[ex] = cls;
            sh.slot_of[cls] = (int)ex;
        }
        carry += tot;
    }
    return (int)carry;
}

// ---------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(T) pmdi_sweep_kernel(const SweepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.x;
    const int K = a.K, N = a.N, P = a.P, cap = a.cap;
    const long long n = a.n, n1 = a.n1;
    const unsigned long long seed = a.seed + (unsigned long long)chain;
    const unsigned iter = a.iter;

    Sh sh;
    {
        size_t o = 0;
        sh.xs = (double *)(smem + o);   o += (size_t)a.Dmax * 8;
        sh.pis = (double *)(smem + o);  o += (size_t)N * 8;
        sh.lw = (double *)(smem + o);   o += (size_t)P * 8;
        sh.term = (double *)(smem + o); o += (size_t)a.terms_cap * 8;
        sh.scan = (unsigned long long *)(smem + o); o += 16 * 8;
        sh.red = (double *)(smem + o);  o += 32 * 8;
        sh.lead_of = (int *)(smem + o); o += (size_t)(P + 1) * 4;
        sh.slot_of = (int *)(smem + o); o += (size_t)(P + 1) * 4;
        sh.kmaxid = (int *)(smem + o);  o += PMDI_KMAX_I * 4;
        sh.kncls = (int *)(smem + o);   o += PMDI_KMAX_I * 4;
        sh.kcur = (int *)(smem + o);    o += PMDI_KMAX_I * 4;
        sh.lab = (int *)(smem + o);     o += 256 * 3 * 4;
        sh.fl = (unsigned char *)(smem + o); o += (size_t)((a.Dmax + 15) & ~15);
        sh.news = (unsigned char *)(smem + o);
    }

    const int *s_in = a.s_in + (size_t)chain * K * n;
    const int *order = a.order + (size_t)chain * n;
    const double *Pi = a.Pi + (size_t)chain * K * N;
    const double *logphi = a.logphi + (size_t)chain * a.npairs;
    const unsigned char *flags = a.flags ? a.flags + (size_t)chain * a.sumD : nullptr;
    double *usc = a.uscratch + (size_t)chain * P;
    int *pstar_raw = a.partstar + (size_t)chain * P;

    long long st_nops = 0, st_nres = 0, st_nclones = 0, st_maxid = 0, st_sumcls = 0;

    for (int p = tid; p < P; p += T) sh.lw[p] = a.lw_init;
    for (int c = tid; c <= P; c += T) { sh.lead_of[c] = PMDI_INF_I; sh.slot_of[c] = 0; }

    // ---- reset (src/pmdi.jl:165-171) and known prefix (src/pmdi.jl:188-207) ----
    for (int k = 0; k < K; ++k) {
        const DsetDev &d = a.ds[k];
        const KS s = make_ks(d, chain);
        const int D = d.D;
        for (int idx = tid; idx <= cap; idx += T) { s.counts[idx] = 0; s.ncop[idx] = 0; s.firstc[idx] = PMDI_INF_I; }
        for (int idx = tid; idx < N * P; idx += T) { s.newid[idx] = 0; s.part[0][idx] = 1; }
        for (int p = tid; p < P; p += T) s.pid[p] = 1;
        for (int u = tid; u < 256; u += T) { sh.lab[u] = PMDI_INF_I; sh.lab[256 + u] = 0; sh.lab[512 + u] = 0; }
        for (int q = tid; q < D; q += T) sh.fl[q] = flags ? flags[d.flag_off + q] : (unsigned char)1;
        __syncthreads();
        // unique(s[order_obs[1:n1-1], k]) in first-appearance order (:192)
        for (long long j = tid; j < n1 - 1; j += T) {
            const int u = s_in[(size_t)k * n + order[j]];
            atomicMin(&sh.lab[u], (int)j);
            atomicAdd(&sh.lab[512 + u], 1);
        }
        __syncthreads();
        if (tid < N) {
            const int u = tid;
            const int fp = sh.lab[u];
            if (fp != PMDI_INF_I) {
                int r = 0;
                for (int v = 0; v < N; ++v) r += (sh.lab[v] < fp) ? 1 : 0;
                sh.lab[256 + u] = 2 + r;        // cluster id of label u (:197)
            }
        }
        __syncthreads();
        int nu = 0;
        for (int v = 0; v < N; ++v) nu += (sh.lab[v] != PMDI_INF_I) ? 1 : 0;
        // particle[u, :, k] .= id ; counts (:195-198)
        for (int idx = tid; idx < N * P; idx += T) {
            const int u = idx / P;
            const int id = sh.lab[256 + u];
            if (id) s.part[0][idx] = id;
        }
        if (tid < N && sh.lab[256 + tid]) s.counts[sh.lab[256 + tid]] = P;
        if (tid == 0) s.counts[1] = P * N - nu * P;
        // fresh clusters 1..nu+1 (:189,:194)
        if (tid == 0) s.cn[1] = 0;
        if (tid < N && sh.lab[256 + tid]) s.cn[sh.lab[256 + tid]] = sh.lab[512 + tid];
        if (d.kind == K_GAUSSIAN) {
            for (int it = tid; it < (nu + 1) * D; it += T) {
                s.ml[D + it] = make_double2(0.0, 1.0);
                s.sb[D + it] = make_double2(0.0, 0.5);
            }
        } else if (d.kind == K_CATEGORICAL) {
            for (int it = tid; it < (nu + 1) * D * d.L; it += T) s.cnt[(size_t)D * d.L + it] = 0;
        } else {
            for (int it = tid; it < (nu + 1) * D; it += T) s.nbs[D + it] = 0;
        }
        __syncthreads();
        // the first n1-1 shuffled observations join their previous cluster,
        // sequentially in shuffled order (:201-206); lanes = (label, feature)
        for (int it = tid; it < N * D; it += T) {
            const int u = it / D, q = it - u * D;
            const int id = sh.lab[256 + u];
            if (!id || !sh.fl[q]) continue;
            if (d.kind == K_GAUSSIAN) {
                double2 ml = make_double2(0.0, 1.0), sb = make_double2(0.0, 0.5);
                int c = 0;
                for (long long j = 0; j < n1 - 1; ++j) {
                    const int i = order[j];
                    if (s_in[(size_t)k * n + i] != u) continue;
                    ++c;
                    gauss_add(d.xf[(size_t)i * D + q], c, ml, sb);
                }
                s.ml[(size_t)id * D + q] = ml;
                s.sb[(size_t)id * D + q] = sb;
            } else if (d.kind == K_CATEGORICAL) {
                int *cn_ = s.cnt + ((size_t)id * D + q) * d.L;
                for (long long j = 0; j < n1 - 1; ++j) {
                    const int i = order[j];
                    if (s_in[(size_t)k * n + i] != u) continue;
                    cn_[d.xi[(size_t)i * D + q] - 1] += 1;
                }
            } else {
                long long S = 0;
                for (long long j = 0; j < n1 - 1; ++j) {
                    const int i = order[j];
                    if (s_in[(size_t)k * n + i] != u) continue;
                    S += d.xi[(size_t)i * D + q];
                }
                s.nbs[(size_t)id * D + q] = S;
            }
        }
        if (tid == 0) {
            sh.kmaxid[k] = nu + 1;
            sh.kncls[k] = 1;
            sh.kcur[k] = 0;
            s.clslead[0] = 0;
            s.clsval[0] = 1;
        }
        __syncthreads();
    }
    if (tid == 0) sh.slot_of[1] = 0;
    __syncthreads();

    // ---- the sweep: src/pmdi.jl:209-342 ----
    int failed = 0;
    for (long long pos = n1 - 1; pos < n && !failed; ++pos) {
        const int i = order[pos];
        for (int k = 0; k < K && !failed; ++k) {
            const DsetDev &d = a.ds[k];
            const KS s = make_ks(d, chain);
            const int D = d.D;
            const int maxid = sh.kmaxid[k];
            const int ncls = sh.kncls[k];
            const int cur = sh.kcur[k];
            int *part = s.part[cur];

            // stage the observation row, Pi[:,k] and the feature flags in LDS
            if (d.kind == K_GAUSSIAN) {
                for (int q = tid; q < D; q += T) sh.xs[q] = d.xf[(size_t)i * D + q];
            } else {
                int *xi_s = (int *)sh.xs;
                for (int q = tid; q < D; q += T) xi_s[q] = d.xi[(size_t)i * D + q];
            }
            for (int q = tid; q < D; q += T) sh.fl[q] = flags ? flags[d.flag_off + q] : (unsigned char)1;
            for (int nn = tid; nn < N; nn += T) sh.pis[nn] = Pi[(size_t)k * N + nn];
            // slot_of is shared by the K datasets: rebuild it from this dataset's class list
            for (int r = tid; r < ncls; r += T) sh.slot_of[s.clsval[r]] = r;
            __syncthreads();
            int nflag = 0;
            for (int q = 0; q < D; ++q) nflag += sh.fl[q];

            // -- A: logprob table over live ids (:218-220): lanes = (id, feature)
            // terms in parallel, then one lane per id adds them in feature order
            {
                const int RS = 2 * D + 1;
                int CH = a.terms_cap / RS;
                if (CH < 1) CH = 1;
                for (int id0 = 1; id0 <= maxid; id0 += CH) {
                    const int nid = min(CH, maxid - id0 + 1);
                    for (int it = tid; it < nid * D; it += T) {
                        const int il = it / D, q = it - il * D;
                        const int id = id0 + il;
                        if (!sh.fl[q]) continue;
                        double ta = 0.0, tb = 0.0;
                        const int cn = s.cn[id];
                        if (d.kind == K_GAUSSIAN) {
                            gauss_terms(sh.xs[q], (double)cn, s.ml[(size_t)id * D + q], ta, tb);
                        } else if (d.kind == K_CATEGORICAL) {
                            const int x = ((const int *)sh.xs)[q];
                            ta = d.lhtab[d.maxcol[q] + 2 * cn];                 // log(nlevels_q + n)
                            const int c = s.cnt[((size_t)id * D + q) * d.L + (x - 1)];
                            tb = (cn == 0) ? d.lhtab[1] : d.lhtab[2 * c + 1];   // log(0.5 + counts)
                        } else {
                            const int x = ((const int *)sh.xs)[q];
                            ta = negbin_term(d.lgtab, cn, x, s.nbs[(size_t)id * D + q]);
                        }
                        sh.term[il * RS + 2 * q] = ta;
                        sh.term[il * RS + 2 * q + 1] = tb;
                    }
                    __syncthreads();
                    for (int il = tid; il < nid; il += T) {
                        const int id = id0 + il;
                        const double *t = sh.term + il * RS;
                        double out;
                        if (d.kind == K_GAUSSIAN) {
                            out = (double)nflag * d.gtab[s.cn[id]];            // gaussian_cluster.jl:38-40
                            for (int q = 0; q < D; ++q)
                                if (sh.fl[q]) { out += t[2 * q]; out -= t[2 * q + 1]; }
                        } else if (d.kind == K_CATEGORICAL) {
                            double acc = 0.0;                                  // categorical_cluster.jl:30
                            for (int q = 0; q < D; ++q) if (sh.fl[q]) acc += t[2 * q];
                            out = -acc;
                            for (int q = 0; q < D; ++q) if (sh.fl[q]) out += t[2 * q + 1];
                        } else {
                            out = 0.0;                                         // negbinom_cluster.jl:25
                            for (int q = 0; q < D; ++q) if (sh.fl[q]) out += t[2 * q];
                        }
                        s.lp[id] = out;
                    }
                    __syncthreads();
                }
            }

            // -- B: mutation CDF per particle class (:231-248): lanes = (class, label)
            // inside a wave; max / cumsum / normalise by shuffles.  The cumsum
            // follows Julia's accumulate_pairwise!: c[n] = e[0] + (e[1]+...+e[n]).
            {
                const int G = 64 / N;
                const int g = lane / N, nn = lane - g * N;
                const int gbase = (g < G) ? g * N : lane;
                for (int r0 = 0; r0 < ncls; r0 += (T / 64) * G) {
                    const int r = r0 + wave * G + g;
                    const bool valid = (g < G) && (r < ncls);
                    double v = 0.0;
                    if (valid) {
                        const int lead = s.clslead[r];
                        v = s.lp[part[nn * P + lead]];
                    }
                    double m = v;
                    for (int j = 0; j < N; ++j) {
                        const double t = __shfl(v, (g < G) ? gbase + j : lane);
                        m = (t > m) ? t : m;
                    }
                    double e = v - m;
                    e = exp(e);
                    e = e * sh.pis[valid ? nn : 0];
                    const double e0 = __shfl(e, gbase);
                    double s_ = 0.0;
                    for (int j = 1; j < N; ++j) {
                        const double t = __shfl(e, (g < G) ? gbase + j : lane);
                        if (j <= nn) s_ = (j == 1) ? t : s_ + t;
                    }
                    const double c = (nn == 0) ? e : e0 + s_;
                    const double fN = __shfl(c, (g < G) ? gbase + N - 1 : lane);
                    if (valid) {
                        s.cdf[(size_t)r * (N + 1) + nn] = c / fN;
                        if (nn == N - 1) s.cdf[(size_t)r * (N + 1) + N] = log(fN) + m;
                    }
                }
            }
            __syncthreads();

            // -- C: allocation draw (:251-265) + class key / chosen-cluster census
            for (int pb = 0; pb < P; pb += T) {
                const int p = pb + tid;
                const bool valid = p < P;
                int cls = 0, ns = 0, c = 0, key = 0;
                bool fresh = false;
                if (valid) {
                    cls = s.pid[p];
                    const double *row = s.cdf + (size_t)sh.slot_of[cls] * (N + 1);
                    if (p != 0) {
                        const double u = uniform01(seed, iter, (unsigned)pos, (unsigned)k, (unsigned)p, SITE_DRAW);
                        for (int t = 0; t < N - 1; ++t) {
                            if (row[ns] > u) break;
                            ++ns;
                        }
                    } else {
                        ns = s_in[(size_t)k * n + i];            // reference trajectory (:262)
                    }
                    sh.lw[p] += row[N];
                    c = part[ns * P + p];
                    s.sid[p] = c;                                // sstar_id (:264)
                    sh.news[k * P + p] = (unsigned char)ns;
                    s.sstar[(size_t)pos * P + p] = (unsigned char)ns;   // (:265)
                    key = (cls - 1) * N + ns;
                    fresh = s.newid[key] <= 0;
                }
                int cnt;
                if (wave_group(key, fresh, cnt)) atomicMin(&s.newid[key], p - P);
                if (wave_group(c, valid, cnt)) {
                    atomicAdd(&s.ncop[c], cnt);
                    atomicMin(&s.firstc[c], p);
                }
            }
            __syncthreads();

            // -- D: ranks in particle order: fresh class keys (:266-269) and
            // distinct chosen clusters, clone-or-in-place (:276-299)
            unsigned long long carry = 0;
            for (int pb = 0; pb < P; pb += T) {
                const int p = pb + tid;
                const bool valid = p < P;
                int key = 0, c = 0;
                bool fk = false, fc = false, nc = false;
                if (valid) {
                    key = (s.pid[p] - 1) * N + sh.news[k * P + p];
                    c = s.sid[p];
                    fk = s.newid[key] == p - P;
                    fc = s.firstc[c] == p;
                    nc = fc && (s.ncop[c] != s.counts[c]);
                }
                unsigned long long tot;
                const unsigned long long pk = (fk ? 1ull : 0ull) | (fc ? (1ull << 20) : 0ull) | (nc ? (1ull << 40) : 0ull);
                const unsigned long long ex = block_excl_scan<T>(pk, tot, sh.scan) + carry;
                if (fk) s.newid[key] = (int)(ex & 0xfffffull) + 1;
                if (fc) {
                    const int rc = (int)((ex >> 20) & 0xfffffull);
                    const int tgt = nc ? maxid + (int)(ex >> 40) + 1 : c;
                    if (tgt <= cap) {
                        const int ncp = s.ncop[c];
                        const int nnew = s.cn[c] + 1;
                        if (nc) { s.counts[c] -= ncp; s.counts[tgt] = ncp; }   // (:293-294)
                        s.cn[tgt] = nnew;
                        s.dl[rc] = c; s.dl[P + rc] = tgt; s.dl[2 * P + rc] = nnew;
                        s.ncop[c] = tgt;                                        // chosen id -> updated id
                    }
                }
                carry += tot;
            }
            const int nd = (int)((carry >> 20) & 0xfffffull);
            const int nclone = (int)(carry >> 40);
            if (maxid + nclone > cap) { failed = 1; }
            __syncthreads();
            if (failed) break;

            // -- E: apply: new class ids, remap cloned labels (:301-308)
            for (int pb = 0; pb < P; pb += T) {
                const int p = pb + tid;
                const bool valid = p < P;
                int newcls = 0;
                if (valid) {
                    const int ns = sh.news[k * P + p];
                    const int key = (s.pid[p] - 1) * N + ns;
                    newcls = s.newid[key];
                    const int c = s.sid[p];
                    const int tgt = s.ncop[c];
                    if (tgt != c) part[ns * P + p] = tgt;
                    s.pid[p] = newcls;
                    s.sid[p] = key;
                }
                int cnt;
                if (wave_group(newcls, valid, cnt)) atomicMin(&sh.lead_of[newcls], p);
            }
            __syncthreads();

            // -- F: class list for the next step; scratch clean-up; and the
            // sufficient-statistic update of every distinct chosen cluster
            // (deepcopy + cluster_add!, :297,:300): lanes = (cluster, feature)
            {
                unsigned long long ccarry = 0;
                for (int pb = 0; pb < P; pb += T) {
                    const int p = pb + tid;
                    const bool valid = p < P;
                    const int cls = valid ? s.pid[p] : 0;
                    const bool isl = valid && sh.lead_of[cls] == p;
                    unsigned long long tot;
                    const unsigned long long ex = block_excl_scan<T>(isl ? 1ull : 0ull, tot, sh.scan) + ccarry;
                    if (isl) { s.clslead[ex] = p; s.clsval[ex] = cls; sh.slot_of[cls] = (int)ex; }
                    if (valid && a.q1 == 1) s.newid[s.sid[p]] = 0;   // corrected mode: new_id per step
                    ccarry += tot;
                }
                for (int j = tid; j < nd; j += T) { const int c = s.dl[j]; s.ncop[c] = 0; s.firstc[c] = PMDI_INF_I; }
                for (int it = tid; it < nd * D; it += T) {
                    const int j = it / D, q = it - j * D;
                    const int src = s.dl[j], dst = s.dl[P + j], nnew = s.dl[2 * P + j];
                    const bool on = sh.fl[q];
                    if (d.kind == K_GAUSSIAN) {
                        double2 ml = s.ml[(size_t)src * D + q], sb = s.sb[(size_t)src * D + q];
                        if (on) gauss_add(sh.xs[q], nnew, ml, sb);
                        if (on || dst != src) { s.ml[(size_t)dst * D + q] = ml; s.sb[(size_t)dst * D + q] = sb; }
                    } else if (d.kind == K_CATEGORICAL) {
                        const int x = ((const int *)sh.xs)[q];
                        const int *cs = s.cnt + ((size_t)src * D + q) * d.L;
                        int *cd = s.cnt + ((size_t)dst * D + q) * d.L;
                        if (dst != src) for (int l = 0; l < d.L; ++l) cd[l] = cs[l];
                        if (on) cd[x - 1] = cs[x - 1] + 1;
                    } else {
                        const int x = ((const int *)sh.xs)[q];
                        s.nbs[(size_t)dst * D + q] = s.nbs[(size_t)src * D + q] + (on ? x : 0);
                    }
                }
                st_nops += maxid;                     // src/__pmdi.jl:187
                st_sumcls += ncls;
                st_nclones += nclone;
                if (maxid + nclone > st_maxid) st_maxid = maxid + nclone;
                __syncthreads();
                for (int r = tid; r < (int)ccarry; r += T) sh.lead_of[s.clsval[r]] = PMDI_INF_I;
                if (tid == 0) { sh.kmaxid[k] = maxid + nclone; sh.kncls[k] = (int)ccarry; }
            }
            __syncthreads();
        }
        if (failed) break;

        // -- Phi_upweight! (src/misc.jl:50-59)
        if (K > 1) {
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
        }

        // -- calc_ESS (src/misc.jl:15-25): wave shuffles + LDS
        double mx = -INFINITY;
        for (int p = tid; p < P; p += T) { const double v = sh.lw[p]; mx = (v > mx) ? v : mx; }
        mx = block_max<T>(mx, sh.red);
        double sa = 0.0, sb2 = 0.0;
        for (int p = tid; p < P; p += T) { const double w = exp(sh.lw[p] - mx); sa += w; sb2 += w * w; }
        block_sum2<T>(sa, sb2, sh.red);
        const double ess = (sa * sa) / sb2;
        const bool resample = ess <= 0.5 * (double)P;            // src/pmdi.jl:317

        if (resample) {
            // draw_partstar (src/misc.jl:27-47)
            ++st_nres;
            const double u01 = uniform01(seed, iter, (unsigned)pos, 0, 0, SITE_RESAMPLE_U);
            const double usl = uniform01(seed, iter, (unsigned)pos, 0, 0, SITE_RESAMPLE_SLOT);
            double *wb = sh.term;
            for (int p = tid; p < P; p += T) wb[p] = exp(sh.lw[p] - mx);
            __syncthreads();
            if (tid == 0) jl_cumsum_inplace(wb, P);               // cumsum (:29), Julia's pairwise order
            if (tid == T - 64) {                                  // u += 1/particles by repeated addition (:34)
                double u = u01 / (double)P;
                const double h = 1.0 / (double)P;
                usc[0] = u;
                for (int j = 1; j < P; ++j) { u += h; usc[j] = u; }
            }
            __syncthreads();
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
#define PMDI_ANC(pp) ((pp) == 0 ? 0 : ((pp) <= js ? pstar_raw[(pp) - 1] : pstar_raw[(pp)]))
            for (int k = 0; k < K; ++k) {                         // src/pmdi.jl:320-340
                const DsetDev &d = a.ds[k];
                const KS s = make_ks(d, chain);
                const int D = d.D;
                const int cur = sh.kcur[k];
                const int oldmax = sh.kmaxid[k];
                const int *src = s.part[cur];
                int *dst = s.part[cur ^ 1];
                for (int idx = tid; idx < N * P; idx += T) {      // particle[:, partstar, k] (:322)
                    const int nn = idx / P, p = idx - nn * P;
                    const int v = src[nn * P + PMDI_ANC(p)];
                    dst[idx] = v;
                    s.ncop[v] = 1;                                // live-id marks
                }
                for (int p = tid; p < P; p += T) s.sid[p] = s.pid[PMDI_ANC(p)];   // (:323)
                for (int id = 1 + tid; id <= oldmax; id += T) s.counts[id] = 0;   // (:326)
                __syncthreads();
                for (int p = tid; p < P; p += T) s.pid[p] = s.sid[p];
                // sort(unique(particle)) ascending -> 1..U' (:329): scan of live marks
                unsigned long long carry = 0;
                for (int b = 0; b < oldmax; b += T) {
                    const int id = 1 + b + tid;
                    const bool live = (id <= oldmax) && s.ncop[id];
                    unsigned long long tot;
                    const unsigned long long ex = block_excl_scan<T>(live ? 1ull : 0ull, tot, sh.scan) + carry;
                    if (live) s.firstc[id] = (int)ex + 1;
                    carry += tot;
                }
                const int newmax = (int)carry;
                __syncthreads();
                for (int idx = tid; idx < N * P; idx += T) {      // relabel + recount (:331-338)
                    const int v = s.firstc[dst[idx]];
                    dst[idx] = v;
                    atomicAdd(&s.counts[v], 1);
                }
                // clusters[k][i] = deepcopy(clusters[k][id]) for id > i, ascending (:336):
                // batches in ascending order, load -> barrier -> store
                for (int b = 0; b < oldmax; b += T) {
                    const int id = 1 + b + tid;
                    const bool mv = (id <= oldmax) && s.ncop[id] && s.firstc[id] != id;
                    const int v = mv ? s.cn[id] : 0;
                    __syncthreads();
                    if (mv) s.cn[s.firstc[id]] = v;
                }
                const long long items = (long long)oldmax * D;
                if (d.kind == K_GAUSSIAN) {
                    for (long long b = 0; b < items; b += T) {
                        const long long it = b + tid;
                        const int id = 1 + (int)(it / D), q = (int)(it - (long long)(id - 1) * D);
                        const bool mv = (it < items) && s.ncop[id] && s.firstc[id] != id;
                        double2 ml = make_double2(0, 0), sb = make_double2(0, 0);
                        if (mv) { ml = s.ml[(size_t)id * D + q]; sb = s.sb[(size_t)id * D + q]; }
                        __syncthreads();
                        if (mv) { const int nid = s.firstc[id]; s.ml[(size_t)nid * D + q] = ml; s.sb[(size_t)nid * D + q] = sb; }
                    }
                } else if (d.kind == K_CATEGORICAL) {
                    const long long itemsL = items * d.L;
                    const int DL = D * d.L;
                    for (long long b = 0; b < itemsL; b += T) {
                        const long long it = b + tid;
                        const int id = 1 + (int)(it / DL), r = (int)(it - (long long)(id - 1) * DL);
                        const bool mv = (it < itemsL) && s.ncop[id] && s.firstc[id] != id;
                        const int v = mv ? s.cnt[(size_t)id * DL + r] : 0;
                        __syncthreads();
                        if (mv) s.cnt[(size_t)s.firstc[id] * DL + r] = v;
                    }
                } else {
                    for (long long b = 0; b < items; b += T) {
                        const long long it = b + tid;
                        const int id = 1 + (int)(it / D), q = (int)(it - (long long)(id - 1) * D);
                        const bool mv = (it < items) && s.ncop[id] && s.firstc[id] != id;
                        const long long v = mv ? s.nbs[(size_t)id * D + q] : 0;
                        __syncthreads();
                        if (mv) s.nbs[(size_t)s.firstc[id] * D + q] = v;
                    }
                }
                __syncthreads();
                for (int id = 1 + tid; id <= oldmax; id += T) { s.ncop[id] = 0; s.firstc[id] = PMDI_INF_I; }
                __syncthreads();
                const int nc2 = rebuild_classes<T>(s, sh, P);
                __syncthreads();
                for (int r = tid; r < nc2; r += T) sh.lead_of[s.clsval[r]] = PMDI_INF_I;
                if (tid == 0) { sh.kmaxid[k] = newmax; sh.kncls[k] = nc2; sh.kcur[k] = cur ^ 1; }
                __syncthreads();
            }
#undef PMDI_ANC
        }

        if (a.trace_on && tid == 0) {
            double *tr = a.trace + ((size_t)chain * (n - n1 + 1) + (pos - (n1 - 1))) * (2 + 2 * K);
            tr[0] = ess;
            tr[1] = resample ? 1.0 : 0.0;
            for (int k = 0; k < K; ++k) { tr[2 + k] = (double)sh.kmaxid[k]; tr[2 + K + k] = (double)sh.kncls[k]; }
        }
    }

    if (failed) {
        if (tid == 0) a.err[chain] = -4;  // PMDI_E_POOL
        return;
    }

    // ---- particle pick (src/pmdi.jl:345-350) + s = sstar[p_star,:,:] (:373) ----
    {
        double mx = -INFINITY;
        for (int p = tid; p < P; p += T) { const double v = sh.lw[p]; mx = (v > mx) ? v : mx; }
        mx = block_max<T>(mx, sh.red);
        double *wb = sh.term;
        for (int p = tid; p < P; p += T) wb[p] = exp(sh.lw[p] - mx);
        __syncthreads();
        if (tid == 0) {   // StatsBase.sample(::Weights): sequential sum and scan, as the oracle
            double sum = 0.0;
            for (int p = 0; p < P; ++p) sum += wb[p];
            const double t = uniform01(seed, iter, 0, 0, 0, SITE_PSTAR) * sum;
            int ip = 0;
            double cw = wb[0];
            while (cw < t && ip < P - 1) { ++ip; cw += wb[ip]; }
            sh.lab[0] = ip;
        }
        __syncthreads();
        const int pstar = sh.lab[0];
        for (long long pp = tid; pp < n; pp += T) {
            const int i = order[pp];
            for (int k = 0; k < K; ++k) {
                int v;
                if (pp < n1 - 1) v = s_in[(size_t)k * n + i];   // sstar[:, i, k] .= s[i, k] (:204)
                else {
                    const unsigned char *ss = (const unsigned char *)(a.ds[k].arena + (size_t)chain * a.ds[k].stride + a.ds[k].o_sstar);
                    v = ss[(size_t)pp * P + pstar];
                }
                a.s_out[((size_t)chain * K + k) * n + i] = v;
            }
        }
        if (a.lw_out) for (int p = tid; p < P; p += T) a.lw_out[(size_t)chain * P + p] = sh.lw[p];
        if (tid < K) {
            a.kstate[((size_t)chain * PMDI_KMAX_I + tid) * 2] = sh.kmaxid[tid];
            a.kstate[((size_t)chain * PMDI_KMAX_I + tid) * 2 + 1] = sh.kcur[tid];
        }
        if (tid == 0) {
            a.pstar[chain] = pstar;
            long long *st = a.stats + (size_t)chain * 8;
            st[ST_NOPS] = st_nops; st[ST_NRESAMPLE] = st_nres; st[ST_NCLONES] = st_nclones;
            st[ST_MAXID] = st_maxid; st[ST_SUMCLASSES] = st_sumcls;
            a.err[chain] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// Stand-alone cluster batches: the calc_logprob / cluster_add! / calc_logmarginal
// protocol on the device.  Cluster b uses pool id b+1 of the batch's arena.
__global__ void cluster_add_kernel(const ClusterBatchArgs a)
{
    const DsetDev &d = a.ds;
    const KS s = make_ks(d, 0);
    const int D = d.D;
    const long long items = (long long)a.B * D;
    for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(it / D), q = (int)(it - (long long)b * D);
        const int id = b + 1;
        const int row = a.rows[b];
        const int nnew = s.cn[id] + 1;
        const bool on = a.flags ? a.flags[q] != 0 : true;
        if (!on) continue;
        if (d.kind == K_GAUSSIAN) {
            double2 ml = s.ml[(size_t)id * D + q], sb = s.sb[(size_t)id * D + q];
            gauss_add(d.xf[(size_t)row * D + q], nnew, ml, sb);
            s.ml[(size_t)id * D + q] = ml; s.sb[(size_t)id * D + q] = sb;
        } else if (d.kind == K_CATEGORICAL) {
            s.cnt[((size_t)id * D + q) * d.L + (d.xi[(size_t)row * D + q] - 1)] += 1;
        } else {
            s.nbs[(size_t)id * D + q] += d.xi[(size_t)row * D + q];
        }
    }
}

__global__ void cluster_ninc_kernel(const ClusterBatchArgs a)
{
    const KS s = make_ks(a.ds, 0);
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += gridDim.x * blockDim.x) s.cn[b + 1] += 1;
}

__global__ void cluster_logprob_kernel(const ClusterBatchArgs a)
{
    const DsetDev &d = a.ds;
    const KS s = make_ks(d, 0);
    const int D = d.D;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += gridDim.x * blockDim.x) {
        const int id = b + 1, row = a.rows[b];
        const int cn = s.cn[id];
        int nflag = 0;
        for (int q = 0; q < D; ++q) nflag += (a.flags ? a.flags[q] != 0 : 1);
        double out;
        if (d.kind == K_GAUSSIAN) {
            out = (double)nflag * d.gtab[cn];
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                double ta, tb;
                gauss_terms(d.xf[(size_t)row * D + q], (double)cn, s.ml[(size_t)id * D + q], ta, tb);
                out += ta; out -= tb;
            }
        } else if (d.kind == K_CATEGORICAL) {
            double acc = 0.0;
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                acc += d.lhtab[d.maxcol[q] + 2 * cn];
            }
            out = -acc;
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                const int c = s.cnt[((size_t)id * D + q) * d.L + (d.xi[(size_t)row * D + q] - 1)];
                out += (cn == 0) ? d.lhtab[1] : d.lhtab[2 * c + 1];
            }
        } else {
            out = 0.0;
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                out += negbin_term(d.lgtab, cn, d.xi[(size_t)row * D + q], s.nbs[(size_t)id * D + q]);
            }
        }
        a.out[b] = out;
    }
}

// calc_logmarginal: gaussian_cluster.jl:68-83, categorical_cluster.jl:53-66,
// negbinom_cluster.jl:53-60
__device__ __forceinline__ double logmarginal_one(const DsetDev &d, int cn, double beta, const int *cnt_q,
                                                  long long S, int q)
{
    if (d.kind == K_GAUSSIAN) {
        const double a_n = ((double)cn / 2.0 + 0.5);
        return (-a_n) * log(beta) + d.lmtab[cn];
    } else if (d.kind == K_CATEGORICAL) {
        const int mc = d.maxcol[q];                       // 2*nlevels_q
        double v = 0.0;
        v += d.lghtab[2 * mc] - d.lghtab[2 * (mc + cn)];
        for (int r = 0; r < mc; ++r) v += d.lghtab[2 * cnt_q[r] + 1];
        return v;
    } else {
        return d.lgtab[S + 1] - d.lgtab[S + (cn + 1 + 1)] + d.lgtab[1 + cn];
    }
}

__global__ void cluster_logmarginal_kernel(const ClusterBatchArgs a)
{
    const DsetDev &d = a.ds;
    const KS s = make_ks(d, 0);
    const int D = d.D;
    const long long items = (long long)a.B * D;
    for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(it / D), q = (int)(it - (long long)b * D);
        const int id = b + 1;
        const double beta = d.kind == K_GAUSSIAN ? s.sb[(size_t)id * D + q].y : 0.0;
        const int *cq = d.kind == K_CATEGORICAL ? s.cnt + ((size_t)id * D + q) * d.L : nullptr;
        const long long S = d.kind == K_NEGBINOM ? s.nbs[(size_t)id * D + q] : 0;
        a.out[it] = logmarginal_one(d, s.cn[id], beta, cq, S, q);
    }
}

// ---------------------------------------------------------------------------
// Feature selection (src/pmdi.jl:354-370).  Block = (chain, dataset, label):
// rebuild the label's cluster from all n rows with every feature on (members
// in ascending row order, :360-364), one lane per feature; emit its
// calc_logmarginal.  A second kernel adds the per-label marginals in the
// order of unique(sstar[p_star,:,k]) and draws the flags (:365-367).
__global__ void __launch_bounds__(256) featsel_label_kernel(const FeatSelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *lcnt = (int *)smem;                  // [256][L] categorical level counts
    __shared__ int s_first, s_count;
    const int N = a.N, K = a.K;
    const long long n = a.n;
    const int u = blockIdx.x % N;
    const int k = (blockIdx.x / N) % K;
    const int chain = blockIdx.x / (N * K);
    const DsetDev &d = a.ds[k];
    const int D = d.D, tid = threadIdx.x;
    const int *traj = a.traj + ((size_t)chain * K + k) * n;
    if (tid == 0) { s_first = PMDI_INF_I; s_count = 0; }
    __syncthreads();
    int myfirst = PMDI_INF_I, mycount = 0;
    for (long long i = tid; i < n; i += 256)
        if (traj[i] == u) { if (myfirst == PMDI_INF_I) myfirst = (int)i; ++mycount; }
    if (mycount) { atomicMin(&s_first, myfirst); atomicAdd(&s_count, mycount); }
    __syncthreads();
    const int first = s_first, cn = s_count;
    if (tid == 0) a.firstpos[((size_t)chain * K + k) * N + u] = first;
    if (first == PMDI_INF_I) return;
    for (int q = tid; q < D; q += 256) {
        double val;
        if (d.kind == K_GAUSSIAN) {
            double2 ml = make_double2(0.0, 1.0), sb = make_double2(0.0, 0.5);
            int c = 0;
            for (long long i = first; i < n; ++i) {
                if (traj[i] != u) continue;
                ++c;
                gauss_add(d.xf[(size_t)i * D + q], c, ml, sb);
            }
            val = logmarginal_one(d, cn, sb.y, nullptr, 0, q);
        } else if (d.kind == K_CATEGORICAL) {
            int *mine = lcnt + (size_t)tid * d.L;
            for (int l = 0; l < d.L; ++l) mine[l] = 0;
            for (long long i = first; i < n; ++i) {
                if (traj[i] != u) continue;
                mine[d.xi[(size_t)i * D + q] - 1] += 1;
            }
            val = logmarginal_one(d, cn, 0.0, mine, 0, q);
        } else {
            long long S = 0;
            for (long long i = first; i < n; ++i) {
                if (traj[i] != u) continue;
                S += d.xi[(size_t)i * D + q];
            }
            val = logmarginal_one(d, cn, 0.0, nullptr, S, q);
        }
        a.lm[((size_t)chain * a.sumD + d.flag_off + q) * N + u] = val;
    }
}

__global__ void featsel_combine_kernel(const FeatSelArgs a)
{
    const int N = a.N, K = a.K;
    const int chain = blockIdx.y;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < a.sumD; g += gridDim.x * blockDim.x) {
        int k = 0;
        for (int kk = 0; kk < K; ++kk) if (g >= a.ds[kk].flag_off) k = kk;
        const int q = g - a.ds[k].flag_off;
        const int *fp = a.firstpos + ((size_t)chain * K + k) * N;
        const double *lm = a.lm + ((size_t)chain * a.sumD + g) * N;
        double prob = a.fnull[g] + 0.0;                            // :357
        int lastpos = -1;
        for (;;) {                                                 // labels by first appearance (:358)
            int best = PMDI_INF_I, bu = -1;
            for (int u = 0; u < N; ++u) { const int f = fp[u]; if (f > lastpos && f < best) { best = f; bu = u; } }
            if (bu < 0) break;
            prob += lm[bu];                                        // :365
            lastpos = best;
        }
        const double r = uniform01(a.seed + (unsigned long long)chain, a.iter, 0, (unsigned)k, (unsigned)q, SITE_FEATSEL);
        const double pr = 1.0 - 1.0 / (exp(prob + 1.0));           // :367
        a.flags_out[(size_t)chain * a.sumD + g] = (unsigned char)(pr > r);
        a.prob_out[(size_t)chain * a.sumD + g] = prob;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
size_t pmdi_sweep_lds_bytes(const SweepArgs &a, int T)
{
    (void)T;
    size_t o = 0;
    o += (size_t)a.Dmax * 8 + (size_t)a.N * 8 + (size_t)a.P * 8 + (size_t)a.terms_cap * 8;
    o += 16 * 8 + 32 * 8;
    o += (size_t)(a.P + 1) * 4 * 2;
    o += PMDI_KMAX_I * 4 * 3 + 256 * 3 * 4;
    o += (size_t)((a.Dmax + 15) & ~15);
    o += (size_t)a.K * a.P;
    return (o + 15) & ~(size_t)15;
}

hipError_t pmdi_launch_sweep(const SweepArgs &a, int n_chains, int T, hipStream_t stream)
{
    const size_t lds = pmdi_sweep_lds_bytes(a, T);
    hipError_t e = hipSuccess;
    if (T == 1024) {
        e = hipFuncSetAttribute((const void *)pmdi_sweep_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(pmdi_sweep_kernel<1024>, dim3(n_chains), dim3(1024), lds, stream, a);
    } else if (T == 512) {
        e = hipFuncSetAttribute((const void *)pmdi_sweep_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(pmdi_sweep_kernel<512>, dim3(n_chains), dim3(512), lds, stream, a);
    } else if (T == 256) {
        e = hipFuncSetAttribute((const void *)pmdi_sweep_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(pmdi_sweep_kernel<256>, dim3(n_chains), dim3(256), lds, stream, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t pmdi_launch_cluster_add(const ClusterBatchArgs &a, hipStream_t stream)
{
    const long long items = (long long)a.B * a.ds.D;
    int grid = (int)((items + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cluster_add_kernel, dim3(grid), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(cluster_ninc_kernel, dim3((a.B + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t pmdi_launch_cluster_logprob(const ClusterBatchArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(cluster_logprob_kernel, dim3((a.B + 63) / 64), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t pmdi_launch_cluster_logmarginal(const ClusterBatchArgs &a, hipStream_t stream)
{
    const long long items = (long long)a.B * a.ds.D;
    int grid = (int)((items + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cluster_logmarginal_kernel, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t pmdi_launch_featsel(const FeatSelArgs &a, int n_chains, hipStream_t stream)
{
    int Lmax = 1;
    for (int k = 0; k < a.K; ++k) if (a.ds[k].kind == K_CATEGORICAL && a.ds[k].L > Lmax) Lmax = a.ds[k].L;
    const size_t lds = (size_t)256 * Lmax * 4;
    if (lds > 60 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(featsel_label_kernel, dim3(n_chains * a.K * a.N), dim3(256), lds, stream, a);
    hipLaunchKernelGGL(featsel_combine_kernel, dim3((a.sumD + 255) / 256, n_chains), dim3(256), 0, stream, a);
    return hipGetLastError();
}
