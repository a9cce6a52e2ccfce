/*
 * pmdi_oracle_hypers.c -- CPU restatement of ParticleMDI's per-iteration host work around the
 * sweep: src/update_hypers.jl (update_v, update_M!, update_Z, update_gamma!, update_Phi!),
 * align_labels! (src/misc.jl:61-96), shuffle!(order_obs) (src/pmdi.jl:172) and the
 * initialisation of src/pmdi.jl:59-96.
 *
 * TEST INFRASTRUCTURE ONLY (see pmdi_oracle.h).  This file keeps the reference's data
 * structures literally: the N^K-row tables c_combn, Gamma_c and Phi_index (src/pmdi.jl:69-92),
 * the dense norm_temp vector, findZindices, the per-swap recounts of align_labels!.  The product
 * (csrc/pmdi_hypers.hip) evaluates the same quantities without the N^K tables; this file is what
 * it is checked against.
 *
 * Parity status: UNPINNED by the reference beyond T3 (update_Z against the brute-force sum,
 * test/runtests.jl:57-108) and T4 (align_labels! on perfectly permuted datasets, :111-134), both
 * restated in tests/test_oracle_hypers.py.  The reference draws from Julia's global RNG (never
 * seeded) through Distributions.jl samplers that are not vendored; here every draw is a
 * counter-based Philox variate (same generator and key layout as the sweep), with the samplers
 * below as the specification shared with the HIP path:
 *   normal  : Box-Muller on two uniforms
 *   gamma   : Marsaglia-Tsang (2000), shape < 1 by the U^(1/shape) boost
 *   sample(0:n, Weights(w)) : StatsBase v0.33 `sample(::AbstractWeights)` (t = rand()*sum(w), scan)
 *   shuffle!: Fisher-Yates from the top (Random.shuffle!: for i = n:-1:2, swap(i, rand(1:i)))
 * Sums that Julia evaluates with `sum` (pairwise/@simd, order not pinnable) run sequentially.
 */
#include "pmdi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS_F64 2.220446049250313e-16 /* eps(Float64) */

/* draw sites (continue the numbering of pmdi_oracle.c; shared with csrc/pmdi_internal.h) */
enum {
    SITE_SHUFFLE = 5, SITE_M_NORMAL = 6, SITE_M_ACCEPT = 7, SITE_GAMMA = 8, SITE_PHI_ALPHA = 9,
    SITE_PHI_GAMMA = 10, SITE_V = 11, SITE_ALIGN = 12, SITE_INIT_GAMMA = 13, SITE_INIT_PHI = 14,
    SITE_INIT_S = 15
};

struct pmdi_oracle_hypers {
    int32_t K, N, npairs;
    int64_t n, NK;
    uint64_t seed;
    double *M;       /* K */
    double *gamma;   /* N x K column-major (gamma_c) */
    double *Phi;     /* npairs */
    double v, Z;
    int64_t *s;      /* n x K column-major, labels 1..N */
    int64_t *order;  /* n, 1-based */
    int32_t *c_combn;   /* NK x K column-major  (src/pmdi.jl:69-72) */
    double *Gamma_c;    /* NK x K column-major  (src/pmdi.jl:75-79), built ONCE */
    uint8_t *Phi_index; /* NK x npairs          (src/pmdi.jl:83-92) */
    int32_t Phi_lab[64][2];
    double *norm_temp;  /* NK scratch */
};

/* ---- samplers (specification shared with the HIP path) ------------------------------------- */
double pmdi_oracle_normal(uint64_t seed, uint32_t iter, uint32_t pos, uint32_t k, uint32_t p0, uint32_t site)
{
    const double u1 = pmdi_oracle_uniform(seed, iter, pos, k, p0, site);
    const double u2 = pmdi_oracle_uniform(seed, iter, pos, k, p0 + 1, site);
    const double r = sqrt(-2.0 * log(u1));
    return r * cos(6.283185307179586 * u2);
}

/* Gamma(shape, 1): attempt t uses uniforms p = 4t, 4t+1 (the normal) and 4t+2 (accept); the
 * shape < 1 boost uses p = 3. */
double pmdi_oracle_gamma(double shape, uint64_t seed, uint32_t iter, uint32_t pos, uint32_t k, uint32_t site)
{
    const double a = shape < 1.0 ? shape + 1.0 : shape;
    const double d = a - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (uint32_t t = 0; t < 1000; ++t) {
        const double x = pmdi_oracle_normal(seed, iter, pos, k, 4 * t, site);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = pmdi_oracle_uniform(seed, iter, pos, k, 4 * t + 2, site);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) { g = d * v; break; }
    }
    if (shape < 1.0) {
        const double u = pmdi_oracle_uniform(seed, iter, pos, k, 3, site);
        g = g * exp(log(u) / shape);
    }
    return g;
}

/* logpdf(Gamma(k, theta), x): StatsFuns gammalogpdf */
static double gamma_logpdf(double k, double theta, double x)
{
    return -lgamma(k) - k * log(theta) + (k - 1.0) * log(x) - x / theta;
}

/* calculate_Phi_lab (src/misc.jl:1-13) */
static void phi_lab(int K, int32_t (*lab)[2])
{
    if (K < 2) { lab[0][0] = 1; lab[0][1] = 1; return; }
    int i = 0;
    for (int k1 = 1; k1 <= K - 1; ++k1)
        for (int k2 = k1 + 1; k2 <= K; ++k2) { lab[i][0] = k1; lab[i][1] = k2; ++i; }
}

/* norm_temp = Phi_index * Phi_log + sum(Gamma, dims = 2)   (update_hypers.jl:33,75,101) */
static void build_norm_temp(pmdi_oracle_hypers *h, int do_exp)
{
    const int K = h->K;
    double philog[64];
    for (int i = 0; i < h->npairs; ++i) philog[i] = log(h->Phi[i] + 1.0);
    for (int64_t r = 0; r < h->NK; ++r) {
        double mv = 0.0;                                 /* Phi_index * Phi_log: row dot product */
        if (K > 1) {
            for (int i = 0; i < h->npairs; ++i) mv += h->Phi_index[(size_t)i * h->NK + r] ? philog[i] : 0.0;
        } else {
            mv = 1.0 * philog[0];                        /* K == 1: Phi_index = fill(1, (N, 1)), Phi = zeros(1) */
        }
        double sg = 0.0;                                 /* sum(Gamma, dims = 2) */
        for (int k = 0; k < K; ++k) sg += h->Gamma_c[(size_t)k * h->NK + r];
        const double t = mv + sg;
        h->norm_temp[r] = do_exp ? exp(t) : t;
    }
}

/* Gamma_c[:, k] = view(log.(gamma[:, k]), c_combn[:, k])  (src/pmdi.jl:75-79).  The reference runs this
 * once, before the first iteration, and never again. */
static void rebuild_gamma_c(pmdi_oracle_hypers *h, const double *gamma0)
{
    for (int k = 0; k < h->K; ++k)
        for (int64_t r = 0; r < h->NK; ++r)
            h->Gamma_c[(size_t)k * h->NK + r] = log(gamma0[(size_t)k * h->N + (h->c_combn[(size_t)k * h->NK + r] - 1)]);
}

/* update_Z (update_hypers.jl:29-39) */
double pmdi_oracle_hypers_update_Z(pmdi_oracle_hypers *h)
{
    build_norm_temp(h, 0);
    double Z = 0.0;
    for (int64_t i = 0; i < h->NK; ++i) Z += exp(h->norm_temp[i]);
    h->Z = Z;
    return Z;
}

/* update_v (update_hypers.jl:1-3): rand(Gamma(n_obs, 1 / Z)) */
double pmdi_oracle_hypers_update_v(pmdi_oracle_hypers *h, int64_t iter)
{
    const double g = pmdi_oracle_gamma((double)h->n, h->seed, (uint32_t)iter, 0, 0, SITE_V);
    h->v = g * (1.0 / h->Z);
    return h->v;
}

/* update_M! (update_hypers.jl:5-26) */
void pmdi_oracle_hypers_update_M(pmdi_oracle_hypers *h, int64_t iter)
{
    const int K = h->K, N = h->N;
    const double prior1 = 2.0, prior2 = 0.25;
    for (int k = 0; k < K; ++k) {
        const double *cg = h->gamma + (size_t)k * N;
        const double cur = h->M[k];
        double ll = 0.0;
        for (int nn = 0; nn < N; ++nn) ll += gamma_logpdf(cur / N, 1.0, cg[nn]);
        ll = -ll;
        const double ll0 = -gamma_logpdf(prior1, prior2, cur);
        const double prop = cur + pmdi_oracle_normal(h->seed, (uint32_t)iter, 0, (uint32_t)k, 0, SITE_M_NORMAL) / 10.0;
        double alpha;
        if (prop <= 0.0) {
            alpha = 0.0;
        } else {
            double nll = 0.0;
            for (int nn = 0; nn < N; ++nn) nll += gamma_logpdf(prop / N, 1.0, cg[nn]);
            nll = -nll;
            const double nll0 = -gamma_logpdf(prior1, prior2, prop);
            alpha = exp(-nll - nll0 + ll + ll0);
        }
        if (pmdi_oracle_uniform(h->seed, (uint32_t)iter, 0, (uint32_t)k, 0, SITE_M_ACCEPT) < alpha) h->M[k] = prop;
    }
}

/* findZindices(k, K, n, N) (src/misc.jl:153-168): rows of the N^K tables whose k-th digit is n (1-based) */
static void find_z_indices(int k, int K, int nlab, int N, int64_t *out)
{
    int64_t pk1 = 1, pKk = 1;
    for (int j = 0; j < k - 1; ++j) pk1 *= N;
    const int64_t pk = pk1 * N;
    for (int j = 0; j < K - k; ++j) pKk *= N;
    int64_t start = (int64_t)(nlab - 1) * pk1 + 1, ind = 0;
    for (int64_t i = 0; i < pKk; ++i) {
        for (int64_t j = start; j <= start - 1 + pk1; ++j) out[ind++] = j;
        start += pk;
    }
}

/* update_gamma! (update_hypers.jl:64-92) */
void pmdi_oracle_hypers_update_gamma(pmdi_oracle_hypers *h, int64_t iter)
{
    const int K = h->K, N = h->N;
    const double beta_0 = 1.0;
    double *alpha_star = (double *)malloc(sizeof(double) * (size_t)N * K);
    for (int k = 0; k < K; ++k)
        for (int nn = 1; nn <= N; ++nn) {
            int64_t cnt = 0;                                              /* countn(s[:, k], n) */
            for (int64_t i = 0; i < h->n; ++i) cnt += h->s[(size_t)k * h->n + i] == nn;
            alpha_star[(size_t)k * N + nn - 1] = h->M[k] / N + (double)cnt;
        }
    build_norm_temp(h, 1);
    int64_t nrows = 1;
    for (int j = 0; j < K - 1; ++j) nrows *= N;
    int64_t *rows = (int64_t *)malloc(sizeof(int64_t) * (size_t)nrows);
    for (int k = 1; k <= K; ++k) {
        find_z_indices(k, K, 1, N, rows);
        int64_t pk1 = 1;
        for (int j = 0; j < k - 1; ++j) pk1 *= N;
        for (int nn = 1; nn <= N; ++nn) {
            double *g = &h->gamma[(size_t)(k - 1) * N + nn - 1];
            const double old_g = *g + 0.0;
            double S = 0.0;
            for (int64_t r = 0; r < nrows; ++r) S += h->norm_temp[rows[r] - 1];
            const double beta_star = beta_0 + h->v * S / *g;
            const double draw = pmdi_oracle_gamma(alpha_star[(size_t)(k - 1) * N + nn - 1], h->seed, (uint32_t)iter,
                                                  (uint32_t)(nn - 1), (uint32_t)(k - 1), SITE_GAMMA);
            *g = draw * (1.0 / beta_star) + EPS_F64;
            for (int64_t r = 0; r < nrows; ++r) h->norm_temp[rows[r] - 1] *= *g / old_g;
            for (int64_t r = 0; r < nrows; ++r) rows[r] += pk1;
        }
    }
    free(rows);
    free(alpha_star);
}

/* update_Phi! (update_hypers.jl:95-128) */
void pmdi_oracle_hypers_update_Phi(pmdi_oracle_hypers *h, int64_t iter)
{
    if (h->K < 2) return;                                                /* src/pmdi.jl:181 */
    const double alpha_0 = 1.0, beta_0 = 5.0;
    build_norm_temp(h, 1);
    double *w = (double *)malloc(sizeof(double) * (size_t)(h->n + 1));
    for (int i = 0; i < h->npairs; ++i) {
        const double cur = h->Phi[i] + 0.0;
        const int a = h->Phi_lab[i][0] - 1, b = h->Phi_lab[i][1] - 1;
        int64_t n_agree = 0;
        for (int64_t j = 0; j < h->n; ++j) n_agree += (h->s[(size_t)a * h->n + j] - h->s[(size_t)b * h->n + j]) == 0;
        const uint8_t *idx = h->Phi_index + (size_t)i * h->NK;           /* pertinent_rows = findall(Phi_index[:, i]) */
        double S = 0.0;
        for (int64_t r = 0; r < h->NK; ++r) if (idx[r]) S += h->norm_temp[r];
        const double beta_star = beta_0 + (h->v * S / (1.0 + cur));
        /* weights = loggamma.((0:n_agree) .+ alpha_0) + logpdf.(Binomial(n_agree, 0.5), 0:n_agree)
         *           - (0:n_agree) .* log(1 / beta_star) */
        const double lb = log(1.0 / beta_star);
        double mx = -INFINITY;
        for (int64_t r = 0; r <= n_agree; ++r) {
            double t = lgamma((double)r + alpha_0);
            t += lgamma((double)n_agree + 1.0) - lgamma((double)r + 1.0) - lgamma((double)(n_agree - r) + 1.0)
                 + (double)r * log(0.5) + (double)(n_agree - r) * log(0.5);
            t -= (double)r * lb;
            w[r] = t;
            if (t > mx) mx = t;
        }
        double sum = 0.0;
        for (int64_t r = 0; r <= n_agree; ++r) { w[r] = exp(w[r] - mx); sum += w[r]; }
        /* sample(0:n_agree, Weights(...)) */
        const double t = pmdi_oracle_uniform(h->seed, (uint32_t)iter, 0, (uint32_t)i, 0, SITE_PHI_ALPHA) * sum;
        int64_t pick = 0;
        double cw = w[0];
        while (cw < t && pick < n_agree) { ++pick; cw += w[pick]; }
        const double alpha_star = alpha_0 + (double)pick;
        const double draw = pmdi_oracle_gamma(alpha_star, h->seed, (uint32_t)iter, 0, (uint32_t)i, SITE_PHI_GAMMA);
        h->Phi[i] = draw * (1.0 / beta_star);
        const double ratio = (1.0 + h->Phi[i]) / (1.0 + cur);
        for (int64_t r = 0; r < h->NK; ++r) if (idx[r]) h->norm_temp[r] *= ratio;
    }
    free(w);
}

/* align_labels! (src/misc.jl:61-96), line by line (recounts over the n observations per proposal) */
void pmdi_oracle_hypers_align_labels(pmdi_oracle_hypers *h, int64_t iter)
{
    const int K = h->K, N = h->N;
    const int64_t n = h->n;
    if (K == 1) return;
    double philog[64], rel[16];
    for (int i = 0; i < h->npairs; ++i) philog[i] = log(h->Phi[i] + 1.0);
    uint8_t *seen = (uint8_t *)malloc((size_t)N + 1);
    int *occupied = (int *)malloc(sizeof(int) * (size_t)N);
    for (int k = 1; k <= K; ++k) {
        int64_t *sk = h->s + (size_t)(k - 1) * n;
        int nocc = 0;                                                    /* occupied = unique(s[:, k]) */
        memset(seen, 0, (size_t)N + 1);
        for (int64_t i = 0; i < n; ++i) if (!seen[sk[i]]) { seen[sk[i]] = 1; occupied[nocc++] = (int)sk[i]; }
        int nrel = 0;                                                    /* relevant Phis: pairs that involve k, in Phi_lab order */
        for (int i = 0; i < h->npairs; ++i)
            if (h->Phi_lab[i][0] == k || h->Phi_lab[i][1] == k) rel[nrel++] = philog[i];
        for (int oi = 0; oi < nocc; ++oi) {
            int label = occupied[oi];
            int any = 0;
            for (int64_t i = 0; i < n; ++i) if (sk[i] == label) { any = 1; break; }
            if (!any) continue;
            for (int new_label = 1; new_label <= N; ++new_label) {
                if (new_label == label) continue;
                /* count_equals over the other datasets' columns (setdiff2(K, k)), in increasing dataset order */
                double sum_keep = 0.0, sum_swap = 0.0;
                int c = 0;
                for (int j = 1; j <= K; ++j) {
                    if (j == k) continue;
                    const int64_t *sj = h->s + (size_t)(j - 1) * n;
                    double c_ll = 0.0, c_nn = 0.0, c_ln = 0.0, c_nl = 0.0;
                    for (int64_t i = 0; i < n; ++i) {
                        if (sk[i] == label) { c_ll += sj[i] == label; c_ln += sj[i] == new_label; }
                        else if (sk[i] == new_label) { c_nn += sj[i] == new_label; c_nl += sj[i] == label; }
                    }
                    sum_keep += c_ll * rel[c] + c_nn * rel[c];
                    sum_swap += c_ln * rel[c] + c_nl * rel[c];
                    ++c;
                }
                const double accept = exp(sum_swap - sum_keep);
                const double u = pmdi_oracle_uniform(h->seed, (uint32_t)iter, (uint32_t)(oi * N + (new_label - 1)),
                                                     (uint32_t)(k - 1), 0, SITE_ALIGN);
                if (u < accept) {
                    for (int64_t i = 0; i < n; ++i) {
                        if (sk[i] == label) sk[i] = new_label;
                        else if (sk[i] == new_label) sk[i] = label;
                    }
                    double *g = h->gamma + (size_t)(k - 1) * N;
                    const double t = g[new_label - 1]; g[new_label - 1] = g[label - 1]; g[label - 1] = t;
                    label = new_label;
                }
            }
        }
    }
    free(seen);
    free(occupied);
}

/* shuffle!(order_obs) (src/pmdi.jl:172): Random.shuffle! = for i in n:-1:2, j = rand(1:i), swap */
void pmdi_oracle_hypers_shuffle(pmdi_oracle_hypers *h, int64_t iter)
{
    for (int64_t i = h->n; i >= 2; --i) {
        const double u = pmdi_oracle_uniform(h->seed, (uint32_t)iter, (uint32_t)i, 0, 0, SITE_SHUFFLE);
        int64_t j = 1 + (int64_t)(u * (double)i);
        if (j > i) j = i;
        const int64_t t = h->order[i - 1]; h->order[i - 1] = h->order[j - 1]; h->order[j - 1] = t;
    }
}

void pmdi_oracle_hypers_destroy(pmdi_oracle_hypers *h)
{
    if (!h) return;
    free(h->M); free(h->gamma); free(h->Phi); free(h->s); free(h->order);
    free(h->c_combn); free(h->Gamma_c); free(h->Phi_index); free(h->norm_temp);
    free(h);
}

/* src/pmdi.jl:59-96 (iteration key 0) */
pmdi_oracle_hypers *pmdi_oracle_hypers_create(int64_t n, int32_t N, int32_t K, uint64_t seed)
{
    if (K < 1 || K > 8 || N < 2 || n < 1) return NULL;
    double nk = 1.0;
    for (int k = 0; k < K; ++k) nk *= (double)N;
    if (nk > 8.0e6) return NULL;                         /* the literal tables: this oracle is sized for tests and the CPU baseline */
    pmdi_oracle_hypers *h = (pmdi_oracle_hypers *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->K = K; h->N = N; h->n = n; h->seed = seed;
    h->npairs = K > 1 ? K * (K - 1) / 2 : 1;
    h->NK = (int64_t)nk;
    phi_lab(K, h->Phi_lab);
    h->M = (double *)malloc(sizeof(double) * (size_t)K);
    h->gamma = (double *)malloc(sizeof(double) * (size_t)N * K);
    h->Phi = (double *)malloc(sizeof(double) * (size_t)h->npairs);
    h->s = (int64_t *)malloc(sizeof(int64_t) * (size_t)n * K);
    h->order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    h->c_combn = (int32_t *)malloc(sizeof(int32_t) * (size_t)h->NK * K);
    h->Gamma_c = (double *)malloc(sizeof(double) * (size_t)h->NK * K);
    h->Phi_index = (uint8_t *)malloc((size_t)h->NK * h->npairs);
    h->norm_temp = (double *)malloc(sizeof(double) * (size_t)h->NK);
    if (!h->M || !h->gamma || !h->Phi || !h->s || !h->order || !h->c_combn || !h->Gamma_c || !h->Phi_index || !h->norm_temp) {
        pmdi_oracle_hypers_destroy(h);
        return NULL;
    }
    for (int k = 0; k < K; ++k) h->M[k] = 1.0 * 2.0;                                       /* :59 */
    for (int k = 0; k < K; ++k)                                                             /* :60 */
        for (int nn = 0; nn < N; ++nn)
            h->gamma[(size_t)k * N + nn] =
                pmdi_oracle_gamma(1.0 / N, seed, 0, (uint32_t)nn, (uint32_t)k, SITE_INIT_GAMMA) * 1.0 + EPS_F64;
    if (K > 1) {                                                                            /* :61 */
        for (int i = 0; i < h->npairs; ++i)
            h->Phi[i] = pmdi_oracle_gamma(1.0, seed, 0, 0, (uint32_t)i, SITE_INIT_PHI) * 0.2;
    } else {
        h->Phi[0] = 0.0;
    }
    for (int k = 0; k < K; ++k) {                                                           /* :63-66 sampleCategorical(n_obs, gamma[:, k]) */
        const double *g = h->gamma + (size_t)k * N;
        double tot = 0.0;
        for (int nn = 0; nn < N; ++nn) tot += g[nn];
        for (int64_t i = 0; i < n; ++i) {
            const double t = pmdi_oracle_uniform(seed, 0, (uint32_t)i, (uint32_t)k, 0, SITE_INIT_S) * tot;
            int pick = 0;
            double cw = g[0];
            while (cw < t && pick < N - 1) { ++pick; cw += g[pick]; }
            h->s[(size_t)k * n + i] = pick + 1;
        }
    }
    for (int k = 1; k <= K; ++k) {                                                          /* :69-72 */
        int64_t div = 1;
        for (int j = 0; j < K - k; ++j) div *= N;                                           /* N^(K-k) */
        for (int64_t r = 0; r < h->NK; ++r) h->c_combn[(size_t)(K - k) * h->NK + r] = (int32_t)((r / div) % N + 1);
    }
    rebuild_gamma_c(h, h->gamma);                                                           /* :75-79 */
    if (K > 1)                                                                              /* :83-92 */
        for (int i = 0; i < h->npairs; ++i) {
            const int a = h->Phi_lab[i][0] - 1, b = h->Phi_lab[i][1] - 1;
            for (int64_t r = 0; r < h->NK; ++r)
                h->Phi_index[(size_t)i * h->NK + r] = h->c_combn[(size_t)a * h->NK + r] == h->c_combn[(size_t)b * h->NK + r];
        }
    else
        memset(h->Phi_index, 1, (size_t)h->NK);
    for (int64_t i = 0; i < n; ++i) h->order[i] = i + 1;                                    /* :160 */
    pmdi_oracle_hypers_update_Z(h);                                                         /* :95 */
    pmdi_oracle_hypers_update_v(h, 0);                                                      /* :96 */
    return h;
}

/* src/pmdi.jl:172-185: shuffle!, update_M!, update_gamma!, Pi, update_Phi!, update_Z, update_v.
 * Pi: N x K column-major. */
void pmdi_oracle_hypers_step(pmdi_oracle_hypers *h, int64_t iter, double *Pi)
{
    pmdi_oracle_hypers_shuffle(h, iter);
    pmdi_oracle_hypers_update_M(h, iter);
    pmdi_oracle_hypers_update_gamma(h, iter);
    for (int k = 0; k < h->K; ++k) {                                                        /* :179 */
        double tot = 0.0;
        for (int nn = 0; nn < h->N; ++nn) tot += h->gamma[(size_t)k * h->N + nn];
        for (int nn = 0; nn < h->N; ++nn) Pi[(size_t)k * h->N + nn] = h->gamma[(size_t)k * h->N + nn] / tot;
    }
    pmdi_oracle_hypers_update_Phi(h, iter);
    pmdi_oracle_hypers_update_Z(h);
    pmdi_oracle_hypers_update_v(h, iter);
}

/* state access for the tests.  what: 0 M[K], 1 gamma[N*K], 2 Phi[npairs], 3 (v, Z), 4 gamma0 = exp(Gamma_c) per (label, dataset).
 * Setting 4 rebuilds Gamma_c (what the reference does once, from the initial gamma). */
int pmdi_oracle_hypers_get(const pmdi_oracle_hypers *h, int what, double *out)
{
    switch (what) {
    case 0: memcpy(out, h->M, sizeof(double) * (size_t)h->K); return 0;
    case 1: memcpy(out, h->gamma, sizeof(double) * (size_t)h->N * h->K); return 0;
    case 2: memcpy(out, h->Phi, sizeof(double) * (size_t)h->npairs); return 0;
    case 3: out[0] = h->v; out[1] = h->Z; return 0;
    case 4: {
        int64_t pk = 1;
        for (int k = 0; k < h->K; ++k) {
            for (int nn = 0; nn < h->N; ++nn) out[(size_t)k * h->N + nn] = exp(h->Gamma_c[(size_t)k * h->NK + (int64_t)nn * pk]);
            pk *= h->N;
        }
        return 0;
    }
    default: return -1;
    }
}

int pmdi_oracle_hypers_set(pmdi_oracle_hypers *h, int what, const double *in)
{
    switch (what) {
    case 0: memcpy(h->M, in, sizeof(double) * (size_t)h->K); return 0;
    case 1: memcpy(h->gamma, in, sizeof(double) * (size_t)h->N * h->K); return 0;
    case 2: memcpy(h->Phi, in, sizeof(double) * (size_t)h->npairs); return 0;
    case 3: h->v = in[0]; h->Z = in[1]; return 0;
    case 4: rebuild_gamma_c(h, in); return 0;
    default: return -1;
    }
}

int64_t *pmdi_oracle_hypers_s(pmdi_oracle_hypers *h) { return h->s; }
int64_t *pmdi_oracle_hypers_order(pmdi_oracle_hypers *h) { return h->order; }
