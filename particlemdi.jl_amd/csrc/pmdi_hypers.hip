// pmdi_hypers.hip -- the per-iteration work AROUND the sweep, device-resident for every chain of a
// handle, so that a Gibbs iteration never leaves the MI355X:
//   gibbs_init_kernel   src/pmdi.jl:59-66,95-96      M, gamma_c, Phi, s, Z, v
//   hypers_kernel       src/pmdi.jl:172-185          shuffle!(order_obs); update_M!, update_gamma!, Pi,
//                                                    update_Phi!, update_Z, update_v (src/update_hypers.jl)
//   align_kernel        src/pmdi.jl:375              align_labels! (src/misc.jl:61-96)
// One workgroup per chain.  Compile with -ffp-contract=off.
//
// No N^K tables.  The reference materialises norm_temp[c] = prod_k g[c_k, k] * prod_{a<b} (1 + Phi_ab)^[c_a == c_b]
// over all N^K label combinations c (src/pmdi.jl:69-92, update_hypers.jl:33,75,101) and sums slices of it.
// Expanding the pair product over edge subsets E of the complete graph on the K datasets,
//     prod_{a<b} (1 + Phi_ab [c_a == c_b]) = sum_E prod_{e in E} Phi_e * [c constant on every component of E],
// and grouping the subsets by the partition of {1..K} their components induce, every such sum becomes a sum
// over set partitions of products of
//     T(B)    = sum_m prod_{j in B} g[m, j]                  (one label shared by the datasets of block B)
//     conn(B) = sum over connected spanning edge sets of B of prod Phi_e,
// i.e. Z = sum_{partitions pi} prod_{B in pi} conn(B) T(B), evaluated by a subset recursion (Zs below), and
//     sum_{c : c_k = m} norm_temp[c]   = sum_{B containing k} conn(B) prod_{j in B} g[m, j] * Zs(complement of B)
//     sum_{c : c_a = c_b} norm_temp[c] = (1 + Phi_ab) * the same sum with a and b contracted into one vertex.
// conn(B) is built from positive terms only (no inclusion-exclusion differences): remove the lowest vertex v of
// B; the rest falls into connected pieces C_i, each tied to v by a non-empty edge set:
//     conn(B) = sum_{partitions {C_i} of B - v} prod_i conn(C_i) * (prod_{u in C_i} (1 + Phi_vu) - 1).
// Cost per chain and iteration: O(K 2^K N + 4^K) instead of O(K N^K).  The stale-Gamma_c behaviour of the
// reference (norm_temp is always rebuilt from the INITIAL gamma) is kept: g starts from gamma0.
//
// align_labels! recounts `count_equals` over the n observations for every (label, new_label) proposal; the
// counts it needs are entries of the N x N contingency tables T[k][j][a][b] = #{i : s[i,k] == a and s[i,j] == b},
// which are built once per call and updated by row/column exchanges when a swap is accepted.
//
// Random numbers: the counter-based Philox stream of the sweep with further draw sites; the samplers
// (Box-Muller normal, Marsaglia-Tsang gamma, StatsBase weighted sample, Fisher-Yates shuffle) are specified in
// oracle/pmdi_oracle_hypers.c, which restates the reference literally (N^K tables, recounts) and is what
// tests/test_gpu_hypers.py compares these kernels with.
#include "pmdi_device.h"

using namespace pmdi_dev;

namespace {

#define EPS_F64 2.220446049250313e-16

__device__ __forceinline__ double normal01(unsigned long long seed, unsigned iter, unsigned pos, unsigned k, unsigned p0, unsigned site)
{
    const double u1 = uniform01(seed, iter, pos, k, p0, site);
    const double u2 = uniform01(seed, iter, pos, k, p0 + 1, site);
    const double r = sqrt(-2.0 * log(u1));
    return r * cos(6.283185307179586 * u2);
}

// Gamma(shape, 1), Marsaglia-Tsang; attempt t uses uniforms 4t, 4t+1 (normal), 4t+2 (accept); boost p = 3
__device__ double gamma_draw(double shape, unsigned long long seed, unsigned iter, unsigned pos, unsigned k, unsigned site)
{
    const double a = shape < 1.0 ? shape + 1.0 : shape;
    const double d = a - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (unsigned t = 0; t < 1000; ++t) {
        const double x = normal01(seed, iter, pos, k, 4 * t, site);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = uniform01(seed, iter, pos, k, 4 * t + 2, site);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) { g = d * v; break; }
    }
    if (shape < 1.0) {
        const double u = uniform01(seed, iter, pos, k, 3, site);
        g = g * exp(log(u) / shape);
    }
    return g;
}

__device__ __forceinline__ double gamma_logpdf(double k, double theta, double x)
{
    return -lgamma(k) - k * log(theta) + (k - 1.0) * log(x) - x / theta;
}

// ---- subset recursions (one lane; K <= 8: masks < 256) ------------------------------------------
// conn[B] for every non-empty B inside `active`; W is the symmetric 8x8 pair-weight matrix.
__device__ void conn_all(unsigned active, const double *W, double *conn, double *F, double *link)
{
    unsigned B = 0;
    while ((B = (B - active) & active) != 0) {
        const int v = __ffs((int)B) - 1;
        const unsigned R = B & (B - 1);
        if (R == 0) { conn[B] = 1.0; continue; }
        link[0] = 0.0;
        unsigned C = 0;
        while ((C = (C - R) & R) != 0) {                       // prod_{u in C} (1 + W[v][u]) - 1, positive terms only
            const int hb = 31 - __clz((int)C);
            const double w = W[v * 8 + hb], r = link[C ^ (1u << hb)];
            link[C] = r + w + r * w;
        }
        F[0] = 1.0;
        unsigned S = 0;
        while ((S = (S - R) & R) != 0) {
            const unsigned l = S & (0u - S), Sp = S ^ l;
            double acc = 0.0;
            unsigned Cp = 0;
            do {
                const unsigned Cc = Cp | l;
                acc += conn[Cc] * link[Cc] * F[Sp ^ Cp];
                Cp = (Cp - Sp) & Sp;
            } while (Cp != 0);
            F[S] = acc;
        }
        conn[B] = F[R];
    }
}

// Zs[S] = sum over set partitions of S of prod conn(B) T(B), for every S inside `active`
__device__ void zs_all(unsigned active, const double *conn, const double *T, double *Zs)
{
    Zs[0] = 1.0;
    unsigned S = 0;
    while ((S = (S - active) & active) != 0) {
        const unsigned l = S & (0u - S), Sp = S ^ l;
        double acc = 0.0;
        unsigned Cp = 0;
        do {
            const unsigned B = Cp | l;
            acc += conn[B] * T[B] * Zs[Sp ^ Cp];
            Cp = (Cp - Sp) & Sp;
        } while (Cp != 0);
        Zs[S] = acc;
    }
}

__device__ __forceinline__ void fill_W(int K, const double *Phi, double *W)
{
    int i = 0;
    for (int a = 0; a < K - 1; ++a)
        for (int b = a + 1; b < K; ++b) { W[a * 8 + b] = Phi[i]; W[b * 8 + a] = Phi[i]; ++i; }
}

struct HCarve { size_t gam, g0, ge, T, conn, Zs, F, link, conn2, W, W2, alpha, Mv, Phi, red, cnt, nag, jbuf, misc, ord, total; };

__host__ __device__ inline void carve_hypers(const GibbsArgs &a, HCarve &c)
{
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
    const size_t KN = (size_t)a.K * a.N;
    c.gam = take(KN * 8); c.g0 = take(KN * 8); c.ge = take(KN * 8); c.alpha = take(KN * 8);
    c.T = take(256 * 8); c.conn = take(256 * 8); c.Zs = take(256 * 8); c.F = take(256 * 8); c.link = take(256 * 8);
    c.conn2 = take(256 * 8);
    c.W = take(64 * 8); c.W2 = take(64 * 8);
    c.Mv = take(16 * 8); c.Phi = take(32 * 8);
    c.red = take(260 * 8);
    c.cnt = take(KN * 4); c.nag = take(32 * 4); c.jbuf = take(256 * 4); c.misc = take(16 * 4);
    c.ord = take(a.order_lds ? (size_t)a.n * 4 : 0);
    c.total = o;
}

// T[B] = sum_m prod_{j in B} ge[j][m] for every non-empty B (lanes = subsets)
__device__ __forceinline__ void compute_T(int K, int N, const double *ge, double *T, int tid, int nthreads)
{
    for (unsigned B = 1 + tid; B < (1u << K); B += nthreads) {
        double acc = 0.0;
        for (int m = 0; m < N; ++m) {
            double p = 1.0;
            for (int j = 0; j < K; ++j) if (B & (1u << j)) p = p * ge[j * N + m];
            acc += p;
        }
        T[B] = acc;
    }
}

// shuffle!(order_obs) (src/pmdi.jl:172): Fisher-Yates from the top, j = rand(1:i); the uniforms of 256
// positions are drawn by the whole workgroup, one lane applies the exchanges in order
template <class Ptr>
__device__ __forceinline__ void shuffle_order(Ptr ord, long long n, unsigned long long seed, unsigned iter, int *jbuf, int tid)
{
    for (long long top = n; top >= 2; top -= 256) {
        const long long i = top - tid;                       // 1-based position handled by this lane
        if (i >= 2) {
            const double u = uniform01(seed, iter, (unsigned)i, 0, 0, SITE_SHUFFLE);
            long long j = 1 + (long long)(u * (double)i);
            if (j > i) j = i;
            jbuf[tid] = (int)(j - 1);
        }
        __syncthreads();
        if (tid == 0) {
            for (int t = 0; t < 256; ++t) {
                const long long ii = top - t;
                if (ii < 2) break;
                const int j = jbuf[t];
                const int x = ord[ii - 1], y = ord[j];
                ord[ii - 1] = y; ord[j] = x;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// src/pmdi.jl:172-185 for one chain per workgroup (256 threads)
__global__ void __launch_bounds__(256) hypers_kernel(const GibbsArgs a, int do_shuffle)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, chain = blockIdx.x;
    const int K = a.K, N = a.N, np = a.npairs;
    const long long n = a.n;
    const unsigned long long seed = a.seed + (unsigned long long)chain;
    const unsigned iter = a.iter;
    HCarve c;
    carve_hypers(a, c);
    double *gam = (double *)(smem + c.gam), *g0 = (double *)(smem + c.g0), *ge = (double *)(smem + c.ge);
    double *alpha = (double *)(smem + c.alpha), *T = (double *)(smem + c.T), *conn = (double *)(smem + c.conn);
    double *Zs = (double *)(smem + c.Zs), *F = (double *)(smem + c.F), *link = (double *)(smem + c.link);
    double *conn2 = (double *)(smem + c.conn2), *W = (double *)(smem + c.W), *W2 = (double *)(smem + c.W2);
    double *Mv = (double *)(smem + c.Mv), *Phi = (double *)(smem + c.Phi), *red = (double *)(smem + c.red);
    int *cnt = (int *)(smem + c.cnt), *nag = (int *)(smem + c.nag), *jbuf = (int *)(smem + c.jbuf), *misc = (int *)(smem + c.misc);
    const unsigned full = (1u << K) - 1u;

    double *gM = a.M + (size_t)chain * K, *gG = a.gamma + (size_t)chain * K * N, *gPhi = a.Phi + (size_t)chain * np;
    const double *gG0 = a.gamma0 + (size_t)chain * K * N;
    double *gvZ = a.vZ + (size_t)chain * 2;
    const int *s = a.s + (size_t)chain * K * n;
    int *order = a.order + (size_t)chain * n;

    for (int e = tid; e < K * N; e += 256) { gam[e] = gG[e]; g0[e] = gG0[e]; ge[e] = gG0[e]; cnt[e] = 0; }
    if (tid < K) Mv[tid] = gM[tid];
    if (tid < np) { Phi[tid] = gPhi[tid]; nag[tid] = 0; }
    if (tid < 64) { W[tid] = 0.0; W2[tid] = 0.0; }
    __syncthreads();
    double v = gvZ[0];

    // ---- shuffle!(order_obs) (:172) ----
    if (do_shuffle) {
        if (a.order_lds) {
            int *ord = (int *)(smem + c.ord);
            for (long long i = tid; i < n; i += 256) ord[i] = order[i];
            __syncthreads();
            shuffle_order(ord, n, seed, iter, jbuf, tid);
            for (long long i = tid; i < n; i += 256) order[i] = ord[i];
        } else {
            shuffle_order(order, n, seed, iter, jbuf, tid);
        }
    }

    // ---- countn(s[:, k], label) for every label (update_hypers.jl:72) and n_agree per pair (:108-115) ----
    for (long long i = tid; i < n; i += 256)
        for (int k = 0; k < K; ++k) atomicAdd(&cnt[k * N + s[(size_t)k * n + i]], 1);
    if (K > 1) {
        int pr = 0;
        for (int k1 = 0; k1 < K - 1; ++k1)
            for (int k2 = k1 + 1; k2 < K; ++k2) {
                int loc = 0;
                for (long long i = tid; i < n; i += 256) loc += (s[(size_t)k1 * n + i] == s[(size_t)k2 * n + i]) ? 1 : 0;
                if (loc) atomicAdd(&nag[pr], loc);
                ++pr;
            }
    }
    __syncthreads();

    // ---- update_M! (update_hypers.jl:5-26): lane k ----
    if (tid < K) {
        const int k = tid;
        const double *cg = gam + k * N;
        const double cur = Mv[k];
        double ll = 0.0;
        for (int m = 0; m < N; ++m) ll += gamma_logpdf(cur / N, 1.0, cg[m]);
        ll = -ll;
        const double ll0 = -gamma_logpdf(2.0, 0.25, cur);
        const double prop = cur + normal01(seed, iter, 0, (unsigned)k, 0, SITE_M_NORMAL) / 10.0;
        double al;
        if (prop <= 0.0) {
            al = 0.0;
        } else {
            double nll = 0.0;
            for (int m = 0; m < N; ++m) nll += gamma_logpdf(prop / N, 1.0, cg[m]);
            nll = -nll;
            const double nll0 = -gamma_logpdf(2.0, 0.25, prop);
            al = exp(-nll - nll0 + ll + ll0);
        }
        if (uniform01(seed, iter, 0, (unsigned)k, 0, SITE_M_ACCEPT) < al) Mv[k] = prop;
    }
    __syncthreads();

    // ---- update_gamma! (update_hypers.jl:64-92) ----
    for (int e = tid; e < K * N; e += 256) alpha[e] = Mv[e / N] / N + (double)cnt[e];
    if (tid == 0) { fill_W(K, Phi, W); conn_all(full, W, conn, F, link); }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        compute_T(K, N, ge, T, tid, 256);
        __syncthreads();
        const unsigned U = full ^ (1u << k);
        if (tid == 0) zs_all(U, conn, T, Zs);
        __syncthreads();
        if (tid < N) {
            const int m = tid;
            // sum(view(norm_temp, pertinent_rows, :)) without the table: rows whose k-th digit is m
            double S = 0.0;
            unsigned Cq = 0;
            do {
                const unsigned B = Cq | (1u << k);
                double p = 1.0;
                for (int j = 0; j < K; ++j) if (B & (1u << j)) p = p * ge[j * N + m];
                S += conn[B] * p * Zs[U ^ Cq];
                Cq = (Cq - U) & U;
            } while (Cq != 0);
            const double old_g = gam[k * N + m];
            const double beta_star = 1.0 + v * S / old_g;
            const double draw = gamma_draw(alpha[k * N + m], seed, iter, (unsigned)m, (unsigned)k, SITE_GAMMA);
            const double g_new = draw * (1.0 / beta_star) + EPS_F64;
            gam[k * N + m] = g_new;
            ge[k * N + m] = ge[k * N + m] * (g_new / old_g);       // norm_temp[i] *= gamma / old_gamma
        }
        __syncthreads();
    }
    // Pi = gamma ./ sum(gamma, dims = 1) (src/pmdi.jl:179)
    for (int e = tid; e < K * N; e += 256) {
        const int k = e / N;
        double tot = 0.0;
        for (int m = 0; m < N; ++m) tot += gam[k * N + m];
        a.Pi[(size_t)chain * K * N + e] = gam[e] / tot;
        gG[e] = gam[e];
    }
    if (tid < K) gM[tid] = Mv[tid];

    // ---- update_Phi! (update_hypers.jl:95-128); norm_temp starts again from the initial gamma ----
    for (int e = tid; e < K * N; e += 256) ge[e] = g0[e];
    __syncthreads();
    compute_T(K, N, ge, T, tid, 256);
    __syncthreads();
    if (K > 1) {
        double *wv = a.wscr + (size_t)chain * (n + 1);
        int pr = 0;
        for (int ka = 0; ka < K - 1; ++ka)
            for (int kb = ka + 1; kb < K; ++kb, ++pr) {
                const double cur = Phi[pr];
                const int na = nag[pr];
                if (tid == 0) {
                    fill_W(K, Phi, W);
                    conn_all(full, W, conn, F, link);
                    const unsigned rest = full ^ (1u << ka) ^ (1u << kb);
                    zs_all(rest, conn, T, Zs);
                    // contract a and b: the merged vertex keeps index a; weight to u = (1+W_au)(1+W_bu) - 1
                    for (int e = 0; e < 64; ++e) W2[e] = W[e];
                    for (int u = 0; u < K; ++u) {
                        if (u == ka || u == kb) continue;
                        const double wa = W[ka * 8 + u], wb = W[kb * 8 + u];
                        const double w = wa + wb + wa * wb;
                        W2[ka * 8 + u] = w; W2[u * 8 + ka] = w;
                    }
                    conn_all(full ^ (1u << kb), W2, conn2, F, link);
                    double S = 0.0;
                    unsigned Cq = 0;
                    do {
                        S += conn2[Cq | (1u << ka)] * T[Cq | (1u << ka) | (1u << kb)] * Zs[rest ^ Cq];
                        Cq = (Cq - rest) & rest;
                    } while (Cq != 0);
                    S = S * (1.0 + cur);
                    red[256] = 5.0 + (v * S / (1.0 + cur));         // beta_star
                }
                __syncthreads();
                const double beta_star = red[256];
                // weights = loggamma.((0:n_agree) .+ 1) + logpdf.(Binomial(n_agree, 0.5), 0:n_agree) - (0:n_agree) .* log(1 / beta_star)
                const double lb = log(1.0 / beta_star);
                const double lh = -0.6931471805599453;             // log(0.5)
                double mx = -INFINITY;
                for (int r = tid; r <= na; r += 256) {
                    double t = a.lgtab[r + 1];
                    t += a.lgtab[na + 1] - a.lgtab[r + 1] - a.lgtab[na - r + 1] + (double)r * lh + (double)(na - r) * lh;
                    t -= (double)r * lb;
                    wv[r] = t;
                    mx = (t > mx) ? t : mx;
                }
                mx = block_max<256>(mx, red);
                __syncthreads();
                // sample(0:n_agree, Weights(exp.(weights .- maximum(weights)))): each lane owns a contiguous chunk
                const int chunk = (na + 1 + 255) / 256;
                const int lo = tid * chunk, hi = min(lo + chunk, na + 1);
                double part = 0.0;
                for (int r = lo; r < hi; ++r) { const double e = exp(wv[r] - mx); wv[r] = e; part += e; }
                red[tid] = part;
                __syncthreads();
                if (tid == 0) {
                    double sum = 0.0;
                    for (int t = 0; t < 256; ++t) sum += red[t];
                    const double tt = uniform01(seed, iter, 0, (unsigned)pr, 0, SITE_PHI_ALPHA) * sum;
                    // first index whose running sum reaches tt (StatsBase sample(::AbstractWeights))
                    double cw = 0.0;
                    int ch = 0;
                    while (ch < 255 && (ch + 1) * chunk <= na && cw + red[ch] < tt) { cw += red[ch]; ++ch; }
                    int pick = ch * chunk;
                    if (pick > na) pick = na;
                    cw += wv[pick];
                    while (cw < tt && pick < na) { ++pick; cw += wv[pick]; }
                    const double alpha_star = 1.0 + (double)pick;
                    const double draw = gamma_draw(alpha_star, seed, iter, 0, (unsigned)pr, SITE_PHI_GAMMA);
                    Phi[pr] = draw * (1.0 / beta_star);
                }
                __syncthreads();
            }
    }
    // ---- update_Z (update_hypers.jl:29-39), update_v (:1-3) ----
    if (tid == 0) {
        fill_W(K, Phi, W);
        conn_all(full, W, conn, F, link);
        zs_all(full, conn, T, Zs);
        const double Z = Zs[full];
        const double g = gamma_draw((double)n, seed, iter, 0, 0, SITE_V);
        gvZ[0] = g * (1.0 / Z);
        gvZ[1] = Z;
    }
    if (tid < np) {
        gPhi[tid] = Phi[tid];
        a.logphi[(size_t)chain * np + tid] = log(1.0 + Phi[tid]);   // src/misc.jl:53
    }
    (void)misc;
}

// src/pmdi.jl:59-66, 95-96 (iteration key 0)
__global__ void __launch_bounds__(256) gibbs_init_kernel(const GibbsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, chain = blockIdx.x;
    const int K = a.K, N = a.N, np = a.npairs;
    const long long n = a.n;
    const unsigned long long seed = a.seed + (unsigned long long)chain;
    HCarve c;
    carve_hypers(a, c);
    double *gam = (double *)(smem + c.gam), *T = (double *)(smem + c.T), *conn = (double *)(smem + c.conn);
    double *Zs = (double *)(smem + c.Zs), *F = (double *)(smem + c.F), *link = (double *)(smem + c.link);
    double *W = (double *)(smem + c.W), *Phi = (double *)(smem + c.Phi);
    const unsigned full = (1u << K) - 1u;
    if (tid < 64) W[tid] = 0.0;
    for (int e = tid; e < K * N; e += 256) {
        const int k = e / N, m = e - k * N;
        gam[e] = gamma_draw(1.0 / N, seed, 0, (unsigned)m, (unsigned)k, SITE_INIT_GAMMA) * 1.0 + EPS_F64;   // :60
    }
    if (tid < np) Phi[tid] = K > 1 ? gamma_draw(1.0, seed, 0, 0, (unsigned)tid, SITE_INIT_PHI) * 0.2 : 0.0;   // :61
    __syncthreads();
    for (int e = tid; e < K * N; e += 256) {
        a.gamma[(size_t)chain * K * N + e] = gam[e];
        a.gamma0[(size_t)chain * K * N + e] = gam[e];
        double tot = 0.0;
        const int k = e / N;
        for (int m = 0; m < N; ++m) tot += gam[k * N + m];
        a.Pi[(size_t)chain * K * N + e] = gam[e] / tot;
    }
    if (tid < K) a.M[(size_t)chain * K + tid] = 1.0 * 2.0;                                                  // :59
    if (tid < np) { a.Phi[(size_t)chain * np + tid] = Phi[tid]; a.logphi[(size_t)chain * np + tid] = log(1.0 + Phi[tid]); }
    // s[:, k] = sampleCategorical(n_obs, gamma[:, k]) (:63-66): inverse CDF, sequential running sum
    for (int k = 0; k < K; ++k) {
        const double *g = gam + k * N;
        double tot = 0.0;
        for (int m = 0; m < N; ++m) tot += g[m];
        for (long long i = tid; i < n; i += 256) {
            const double t = uniform01(seed, 0, (unsigned)i, (unsigned)k, 0, SITE_INIT_S) * tot;
            int pick = 0;
            double cw = g[0];
            while (cw < t && pick < N - 1) { ++pick; cw += g[pick]; }
            a.s[((size_t)chain * K + k) * n + i] = pick;
        }
    }
    for (long long i = tid; i < n; i += 256) a.order[(size_t)chain * n + i] = (int)i;                      // :160
    compute_T(K, N, gam, T, tid, 256);
    __syncthreads();
    if (tid == 0) {                                                                                         // :95-96
        fill_W(K, Phi, W);
        conn_all(full, W, conn, F, link);
        zs_all(full, conn, T, Zs);
        const double Z = Zs[full];
        const double g = gamma_draw((double)n, seed, 0, 0, 0, SITE_V);
        a.vZ[(size_t)chain * 2] = g * (1.0 / Z);
        a.vZ[(size_t)chain * 2 + 1] = Z;
    }
}

// ---------------------------------------------------------------------------------------------
// align_labels! (src/misc.jl:61-96) through contingency tables; one chain per workgroup (256 threads).
// tab(k, j)[x * N + y] = #{i : s[i, k] == x and s[i, j] == y}, k != j.
__global__ void __launch_bounds__(256) align_kernel(const GibbsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, chain = blockIdx.x;
    const int K = a.K, N = a.N, np = a.npairs;
    const long long n = a.n;
    if (K == 1) return;
    const unsigned long long seed = a.seed + (unsigned long long)chain;
    const unsigned iter = a.iter;
    const size_t tabsz = (size_t)K * K * N * N;
    int *tabs = a.ctab_lds ? (int *)smem : a.ctab + (size_t)chain * tabsz;
    unsigned char *sm2 = smem + (a.ctab_lds ? ((tabsz * 4 + 15) & ~(size_t)15) : 0);
    double *philog = (double *)sm2;                 // [32]
    double *rel = philog + 32;                      // [8]
    double *keep = rel + 8;                         // [8]
    double *swp = keep + 8;                         // [8]
    double *gk = swp + 8;                           // [N <= 256]
    int *firstpos = (int *)(gk + 256);              // [256]
    int *occ = firstpos + 256;                      // [256]
    int *perm = occ + 256;                          // [256]
    int *lcount = perm + 256;                       // [256]
    int *misc = lcount + 256;                       // [8]
    int *s = a.s + (size_t)chain * K * n;
    double *gG = a.gamma + (size_t)chain * K * N;

    for (size_t e = tid; e < tabsz; e += 256) tabs[e] = 0;
    if (tid < np) philog[tid] = log(a.Phi[(size_t)chain * np + tid] + 1.0);
    __syncthreads();
    for (long long i = tid; i < n; i += 256) {
        int lab[PMDI_KMAX_I];
        for (int k = 0; k < K; ++k) lab[k] = s[(size_t)k * n + i];
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < K; ++j)
                if (j != k) atomicAdd(&tabs[((size_t)(k * K + j) * N + lab[k]) * N + lab[j]], 1);
    }
    __syncthreads();

    for (int k = 0; k < K; ++k) {
        int *sk = s + (size_t)k * n;
        // occupied = unique(s[:, k]): labels in order of first appearance over i = 1..n
        if (tid < 256) { firstpos[tid] = PMDI_INF_I; lcount[tid] = 0; perm[tid] = tid; }
        if (tid < N) gk[tid] = gG[k * N + tid];
        __syncthreads();
        for (long long i = tid; i < n; i += 256) { const int l = sk[i]; atomicMin(&firstpos[l], (int)i); atomicAdd(&lcount[l], 1); }
        __syncthreads();
        if (tid < N && firstpos[tid] != PMDI_INF_I) {
            int r = 0;
            for (int u = 0; u < N; ++u) r += (firstpos[u] < firstpos[tid]) ? 1 : 0;
            occ[r] = tid;
        }
        if (tid == 0) {
            int nocc = 0, nrel = 0, pr = 0;
            for (int u = 0; u < N; ++u) nocc += (firstpos[u] != PMDI_INF_I) ? 1 : 0;
            misc[0] = nocc;
            for (int k1 = 0; k1 < K - 1; ++k1)                    // relevant Phis: the pairs that involve k, in Phi_lab order
                for (int k2 = k1 + 1; k2 < K; ++k2) { if (k1 == k || k2 == k) rel[nrel++] = philog[pr]; ++pr; }
        }
        __syncthreads();
        // the proposals are strictly sequential: one wave walks them, the other waves wait
        if (tid < 64) {
            const int lane = tid;
            const int nocc = misc[0];
            for (int oi = 0; oi < nocc; ++oi) {
                int label = occ[oi];
                if (lcount[label] == 0) continue;                  // all(label_ind .== false) && continue
                for (int nl = 0; nl < N; ++nl) {
                    if (nl == label) continue;
                    if (lane < K - 1) {
                        const int j = lane < k ? lane : lane + 1;  // the other datasets in increasing order (setdiff2)
                        const int *t = tabs + (size_t)(k * K + j) * N * N;
                        const double c_ll = (double)t[label * N + label], c_nn = (double)t[nl * N + nl];
                        const double c_ln = (double)t[label * N + nl], c_nl = (double)t[nl * N + label];
                        keep[lane] = c_ll * rel[lane] + c_nn * rel[lane];
                        swp[lane] = c_ln * rel[lane] + c_nl * rel[lane];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double sk_ = 0.0, ss_ = 0.0;
                    for (int c = 0; c < K - 1; ++c) { sk_ += keep[c]; ss_ += swp[c]; }
                    const double accept = exp(ss_ - sk_);
                    const double u = uniform01(seed, iter, (unsigned)(oi * N + nl), (unsigned)k, 0, SITE_ALIGN);
                    __builtin_amdgcn_wave_barrier();
                    if (u < accept) {
                        // s[label_ind, k] .= new_label; s[new_label_ind, k] .= label  ->  exchange rows of tab(k, j)
                        // and columns of tab(j, k)
                        for (int e = lane; e < (K - 1) * N; e += 64) {
                            const int c = e / N, y = e - c * N;
                            const int j = c < k ? c : c + 1;
                            int *t1 = tabs + (size_t)(k * K + j) * N * N;
                            const int x0 = t1[label * N + y], x1 = t1[nl * N + y];
                            t1[label * N + y] = x1; t1[nl * N + y] = x0;
                            int *t2 = tabs + (size_t)(j * K + k) * N * N;
                            const int y0 = t2[y * N + label], y1 = t2[y * N + nl];
                            t2[y * N + label] = y1; t2[y * N + nl] = y0;
                        }
                        for (int o = lane; o < N; o += 64) {
                            const int pv = perm[o];
                            if (pv == label) perm[o] = nl; else if (pv == nl) perm[o] = label;
                        }
                        if (lane == 0) {
                            const double tg = gk[nl]; gk[nl] = gk[label]; gk[label] = tg;
                            const int tc = lcount[nl]; lcount[nl] = lcount[label]; lcount[label] = tc;
                        }
                        label = nl;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        __syncthreads();
        for (long long i = tid; i < n; i += 256) sk[i] = perm[sk[i]];
        if (tid < N) gG[k * N + tid] = gk[tid];
        __syncthreads();
    }
}

__global__ void pack_samples_kernel(const int *__restrict__ s, unsigned char *__restrict__ out, long long count)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        out[i] = (unsigned char)s[i];
}

}  // namespace

size_t pmdi_hypers_lds_bytes(const GibbsArgs &a)
{
    HCarve c;
    carve_hypers(a, c);
    return c.total;
}

size_t pmdi_align_lds_bytes(const GibbsArgs &a)
{
    const size_t tabsz = (size_t)a.K * a.K * a.N * a.N * 4;
    return (a.ctab_lds ? ((tabsz + 15) & ~(size_t)15) : 0) + (32 + 8 + 8 + 8 + 256) * 8 + (256 * 4 + 8) * 4 + 64;
}

hipError_t pmdi_launch_gibbs_init(const GibbsArgs &a, hipStream_t stream)
{
    const size_t lds = pmdi_hypers_lds_bytes(a);
    hipError_t e = hipFuncSetAttribute((const void *)gibbs_init_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gibbs_init_kernel, dim3(a.n_chains), dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t pmdi_launch_hypers(const GibbsArgs &a, int do_shuffle, hipStream_t stream)
{
    const size_t lds = pmdi_hypers_lds_bytes(a);
    hipError_t e = hipFuncSetAttribute((const void *)hypers_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(hypers_kernel, dim3(a.n_chains), dim3(256), lds, stream, a, do_shuffle);
    return hipGetLastError();
}

hipError_t pmdi_launch_align(const GibbsArgs &a, hipStream_t stream)
{
    if (a.K == 1) return hipSuccess;
    const size_t lds = pmdi_align_lds_bytes(a);
    hipError_t e = hipFuncSetAttribute((const void *)align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(align_kernel, dim3(a.n_chains), dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t pmdi_launch_pack_samples(const int *s, unsigned char *out, long long count, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    long long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_samples_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, s, out, count);
    return hipGetLastError();
}
