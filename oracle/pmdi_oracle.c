/*
 * pmdi_oracle.c -- CPU restatement of ParticleMDI's conditional-SMC sweep.
 *
 * TEST INFRASTRUCTURE ONLY (see pmdi_oracle.h).  Plain C99, single thread,
 * compile with -O2 -ffp-contract=off (no FMA contraction: the arithmetic
 * order below is the reference's).
 *
 * The loops are kept line-faithful to the reference on purpose (sequential
 * over particles, pool of cluster objects with copy-on-write, class cache),
 * including the two behaviours that look like bugs but define its results
 * (Q1 stale new_id, Q2 sstar not resampled).
 *
 * What is NOT the reference's: the random numbers.  Julia's global RNG is
 * replaced at the five draw sites by a counter-based Philox4x32-10 keyed on
 * (seed; iteration, observation position, dataset, particle, site), which
 * the HIP path shares as a specification.
 *
 * Third-party arithmetic restated (not vendored in /root/reference):
 *   - SpecialFunctions v0.8.0 loggamma (Manifest.toml:298-302) -> C lgamma
 *   - Base.cumsum!/cumsum on Float64 vectors = Base.accumulate_pairwise!
 *     (julia base/accumulate.jl, block size 128) -> jl_cumsum() below
 *   - Base.sum on Float64 arrays (pairwise + @simd, order depends on the
 *     vector width of the machine, cannot be pinned) -> sequential sum
 *   - StatsBase v0.33.0 sample(::Weights) (Manifest.toml:314-318) -> inverse
 *     CDF scan, restated in pick_pstar()
 */
#include "pmdi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------ */
/* RNG: Philox4x32-10 (Salmon et al., SC'11).  Specification shared with the
 * HIP path: key = (seed lo, seed hi); ctr = (p, pos, site<<16 | k, iter);
 * u = (2*((w0>>6)*2^26 + (w1>>6)) + 1) * 2^-53 in (0,1). */
enum { SITE_DRAW = 0, SITE_RESAMPLE_U = 1, SITE_RESAMPLE_SLOT = 2, SITE_PSTAR = 3, SITE_FEATSEL = 4 };

void pmdi_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double pmdi_oracle_uniform(uint64_t seed, uint32_t iter, uint32_t pos, uint32_t k,
                           uint32_t p, uint32_t site)
{
    uint32_t ctr[4] = { p, pos, (site << 16) | k, iter };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t w[4];
    pmdi_oracle_philox4x32_10(ctr, key, w);
    /* 52 random bits -> an odd multiple of 2^-53: uniform on the OPEN interval (0,1), exactly
     * representable, never 0 (rand() at src/pmdi.jl:253 is in [0,1); excluding the single point 0
     * lets a numerically one-hot CDF decide a draw without looking at u) */
    uint64_t m = ((uint64_t)(w[0] >> 6) << 26) | (uint64_t)(w[1] >> 6);
    return (double)(2 * m + 1) * (1.0 / 9007199254740992.0);
}

/* ------------------------------------------------------------------------ */
/* Julia Base.cumsum! on a Float64 vector = accumulate_pairwise!(add_sum,..)
 * (base/accumulate.jl): c[1] = v[1]; the rest is _accumulate_pairwise! with
 * carry s = v[1]; blocks shorter than 128 run a local sum s_ and emit
 * c[i] = s + s_.  In-place (c == v) is safe: v[i] is read before c[i]. */
static double jl_acc_pairwise(double *c, const double *v, double s, int64_t i1, int64_t n)
{
    double s_;
    if (n < 128) {
        s_ = v[i1];
        c[i1] = s + s_;
        for (int64_t i = i1 + 1; i < i1 + n; ++i) {
            s_ = s_ + v[i];
            c[i] = s + s_;
        }
    } else {
        int64_t n2 = n >> 1;
        s_ = jl_acc_pairwise(c, v, s, i1, n2);
        s_ += jl_acc_pairwise(c, v, s + s_, i1 + n2, n - n2);
    }
    return s_;
}

static void jl_cumsum(double *c, const double *v, int64_t n)
{
    if (n == 0) return;
    double v1 = v[0];
    c[0] = v1;
    if (n == 1) return;
    jl_acc_pairwise(c, v, v1, 1, n - 1);
}

/* ------------------------------------------------------------------------ */
/* Cluster pool for one dataset: struct-of-arrays over ids 0..cap (id 0
 * unused, ids are 1-based like the reference's clusters[k][id]).           */
typedef struct {
    int32_t kind, D, L;
    int64_t n_obs;
    double  *xf;      /* row-major n x D copy (Gaussian) */
    int64_t *xi;      /* row-major n x D copy (Categorical / NegBinom) */
    double  *nlevels; /* categorical_cluster.jl:10: 0.5 * max of column */
    int64_t cap;      /* number of ids allocated (cap+1 slots) */
    int64_t *cn;      /* cl.n */
    double  *mu, *sum, *lam, *beta;   /* gaussian_cluster.jl:11-22 */
    int64_t *cnt;     /* categorical counts [id][q][level]  (counts[level,q]) */
    int64_t *nbsum;   /* negbinom_cluster.jl:8 */
    double  *feature_null;            /* src/pmdi.jl:127 */
} pool_t;

static int pool_alloc(pool_t *pl, int64_t cap)
{
    int64_t m = cap + 1, D = pl->D;
    pl->cap = cap;
    pl->cn = (int64_t *)calloc((size_t)m, sizeof(int64_t));
    if (!pl->cn) return -1;
    if (pl->kind == PMDI_O_GAUSSIAN) {
        pl->mu = (double *)malloc((size_t)(m * D) * sizeof(double));
        pl->sum = (double *)malloc((size_t)(m * D) * sizeof(double));
        pl->lam = (double *)malloc((size_t)(m * D) * sizeof(double));
        pl->beta = (double *)malloc((size_t)(m * D) * sizeof(double));
        if (!pl->mu || !pl->sum || !pl->lam || !pl->beta) return -1;
    } else if (pl->kind == PMDI_O_CATEGORICAL) {
        pl->cnt = (int64_t *)malloc((size_t)(m * D * pl->L) * sizeof(int64_t));
        if (!pl->cnt) return -1;
    } else {
        pl->nbsum = (int64_t *)malloc((size_t)(m * D) * sizeof(int64_t));
        if (!pl->nbsum) return -1;
    }
    return 0;
}

static void pool_free(pool_t *pl)
{
    free(pl->xf); free(pl->xi); free(pl->nlevels); free(pl->cn);
    free(pl->mu); free(pl->sum); free(pl->lam); free(pl->beta);
    free(pl->cnt); free(pl->nbsum); free(pl->feature_null);
    memset(pl, 0, sizeof(*pl));
}

/* T(dataFile): gaussian_cluster.jl:17-21, categorical_cluster.jl:6-10,
 * negbinom_cluster.jl:9-10 */
static void cl_reset(pool_t *pl, int64_t id)
{
    int64_t D = pl->D;
    pl->cn[id] = 0;
    if (pl->kind == PMDI_O_GAUSSIAN) {
        for (int64_t q = 0; q < D; ++q) {
            pl->mu[id * D + q] = 0.0;
            pl->sum[id * D + q] = 0.0;
            pl->lam[id * D + q] = 1.0;
            pl->beta[id * D + q] = 0.5;
        }
    } else if (pl->kind == PMDI_O_CATEGORICAL) {
        memset(pl->cnt + id * D * pl->L, 0, (size_t)(D * pl->L) * sizeof(int64_t));
    } else {
        memset(pl->nbsum + id * D, 0, (size_t)D * sizeof(int64_t));
    }
}

/* deepcopy(clusters[k][src]): src/pmdi.jl:297,336 */
static void cl_copy(pool_t *pl, int64_t dst, int64_t src)
{
    int64_t D = pl->D;
    if (dst == src) return;
    pl->cn[dst] = pl->cn[src];
    if (pl->kind == PMDI_O_GAUSSIAN) {
        memcpy(pl->mu + dst * D, pl->mu + src * D, (size_t)D * sizeof(double));
        memcpy(pl->sum + dst * D, pl->sum + src * D, (size_t)D * sizeof(double));
        memcpy(pl->lam + dst * D, pl->lam + src * D, (size_t)D * sizeof(double));
        memcpy(pl->beta + dst * D, pl->beta + src * D, (size_t)D * sizeof(double));
    } else if (pl->kind == PMDI_O_CATEGORICAL) {
        memcpy(pl->cnt + dst * D * pl->L, pl->cnt + src * D * pl->L,
               (size_t)(D * pl->L) * sizeof(int64_t));
    } else {
        memcpy(pl->nbsum + dst * D, pl->nbsum + src * D, (size_t)D * sizeof(int64_t));
    }
}

/* cluster_add!: gaussian_cluster.jl:54-66, categorical_cluster.jl:43-51,
 * negbinom_cluster.jl:43-51.  flag == NULL means all features on. */
static void cl_add(pool_t *pl, int64_t id, int64_t row, const uint8_t *flag)
{
    int64_t D = pl->D;
    pl->cn[id] += 1;
    int64_t n = pl->cn[id];
    if (pl->kind == PMDI_O_GAUSSIAN) {
        const double *x = pl->xf + row * D;
        double *mu = pl->mu + id * D, *sm = pl->sum + id * D;
        double *lam = pl->lam + id * D, *beta = pl->beta + id * D;
        for (int64_t q = 0; q < D; ++q) {
            if (flag && !flag[q]) continue;
            sm[q] += x[q];
            double d = x[q] - mu[q];
            beta[q] += ((double)(n - 1) + 0.001) * (d * d) / (2.0 * ((double)n + 0.001));
            mu[q] = sm[q] / ((double)n + 0.001);
            lam[q] = ((0.5 * (double)n + 0.5) * ((double)n + 0.001)) /
                     (beta[q] * ((double)n + 1.001));
        }
    } else if (pl->kind == PMDI_O_CATEGORICAL) {
        const int64_t *x = pl->xi + row * D;
        int64_t *cnt = pl->cnt + id * D * pl->L;
        for (int64_t q = 0; q < D; ++q) {
            if (flag && !flag[q]) continue;
            cnt[q * pl->L + (x[q] - 1)] += 1;
        }
    } else {
        const int64_t *x = pl->xi + row * D;
        int64_t *sm = pl->nbsum + id * D;
        for (int64_t q = 0; q < D; ++q) {
            if (flag && !flag[q]) continue;
            sm[q] += x[q];
        }
    }
}

/* calc_logprob: gaussian_cluster.jl:37-52, categorical_cluster.jl:29-41,
 * negbinom_cluster.jl:22-41 */
static double cl_logprob(const pool_t *pl, int64_t id, int64_t row, const uint8_t *flag)
{
    int64_t D = pl->D;
    int64_t ni = pl->cn[id];
    double n = (double)ni;
    if (pl->kind == PMDI_O_GAUSSIAN) {
        const double *x = pl->xf + row * D;
        const double *mu = pl->mu + id * D, *lam = pl->lam + id * D;
        int64_t F = 0;
        for (int64_t q = 0; q < D; ++q) F += (!flag || flag[q]) ? 1 : 0;
        /* gaussian_cluster.jl:38-40 (@fastmath may re-associate in Julia;
         * restated left to right) */
        double out = (double)F * ((log(1.0 / sqrt(M_PI)) + lgamma(0.5 * n + 1.0)) -
                                  lgamma(0.5 * n + 0.5));
        for (int64_t q = 0; q < D; ++q) {
            if (flag && !flag[q]) continue;
            out += 0.5 * log(lam[q] / (n + 1.0));
            double d = x[q] - mu[q];
            out -= (0.5 * n + 1.0) * log(1.0 + (1.0 / (n + 1.0)) * (d * d) * lam[q]);
        }
        return out;
    } else if (pl->kind == PMDI_O_CATEGORICAL) {
        const int64_t *x = pl->xi + row * D;
        const int64_t *cnt = pl->cnt + id * D * pl->L;
        /* categorical_cluster.jl:30: -sum(log.(nlevels[flag] .+ n)); Base.sum
         * order is not pinnable -> sequential */
        double acc = 0.0;
        for (int64_t q = 0; q < D; ++q) {
            if (flag && !flag[q]) continue;
            acc += log(pl->nlevels[q] + n);
        }
        double out = -acc;
        for (int64_t q = 0; q < D; ++q) {
            if (flag && !flag[q]) continue;
            if (ni == 0) out += log(0.5);
            else out += log(0.5 + (double)cnt[q * pl->L + (x[q] - 1)]);
        }
        return out;
    } else {
        const int64_t *x = pl->xi + row * D;
        const int64_t *sm = pl->nbsum + id * D;
        double out = 0.0;
        for (int64_t q = 0; q < D; ++q) {
            if (flag && !flag[q]) continue;
            int64_t S = sm[q], xo = x[q];
            out += lgamma((double)(1 + ni + 1)) + lgamma((double)(1 + xo + S)) +
                   lgamma((double)(1 + ni + 1 + S)) - lgamma((double)(1 + ni + 1 + 1 + xo + S)) -
                   lgamma((double)(1 + ni)) - lgamma((double)(1 + S));
        }
        return out;
    }
}

/* calc_logmarginal: gaussian_cluster.jl:68-83, categorical_cluster.jl:53-66,
 * negbinom_cluster.jl:53-60 */
static void cl_logmarginal(const pool_t *pl, int64_t id, double *lm)
{
    int64_t D = pl->D;
    int64_t ni = pl->cn[id];
    if (pl->kind == PMDI_O_GAUSSIAN) {
        double a_n = ((double)ni / 2.0 + 0.5), a_0 = 0.5, b_0 = 0.5, k_0 = 0.001;
        double k_n = (double)ni + k_0;
        double c = (a_0 * log(b_0)) + lgamma(a_n) - lgamma(a_0) +
                   0.5 * (log(k_0) - log(k_n)) - ((double)ni * 0.5) * log(2.0 * M_PI);
        const double *beta = pl->beta + id * D;
        for (int64_t q = 0; q < D; ++q) lm[q] = (-a_n) * log(beta[q]) + c;
    } else if (pl->kind == PMDI_O_CATEGORICAL) {
        const int64_t *cnt = pl->cnt + id * D * pl->L;
        for (int64_t q = 0; q < D; ++q) {
            double v = 0.0;
            v += lgamma(pl->nlevels[q] * 2.0) - lgamma(pl->nlevels[q] * 2.0 + (double)ni);
            int64_t R = (int64_t)(2.0 * pl->nlevels[q]);
            for (int64_t r = 0; r < R; ++r) v += lgamma((double)cnt[q * pl->L + r] + 0.5);
            lm[q] = v;
        }
    } else {
        const int64_t *sm = pl->nbsum + id * D;
        for (int64_t q = 0; q < D; ++q)
            lm[q] = lgamma((double)(sm[q] + 1)) - lgamma((double)(sm[q] + (ni + 1 + 1))) +
                    lgamma((double)(1 + ni));
    }
}

static int pool_init_data(pool_t *pl, const pmdi_oracle_dataset *ds, int64_t n)
{
    memset(pl, 0, sizeof(*pl));
    pl->kind = ds->kind; pl->D = ds->D; pl->n_obs = n;
    int64_t D = ds->D;
    if (ds->kind == PMDI_O_GAUSSIAN) {
        if (!ds->xf) return -1;
        pl->xf = (double *)malloc((size_t)(n * D) * sizeof(double));
        if (!pl->xf) return -1;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t q = 0; q < D; ++q) pl->xf[i * D + q] = ds->xf[q * n + i];
    } else {
        if (!ds->xi) return -1;
        pl->xi = (int64_t *)malloc((size_t)(n * D) * sizeof(int64_t));
        if (!pl->xi) return -1;
        for (int64_t i = 0; i < n; ++i)
            for (int64_t q = 0; q < D; ++q) pl->xi[i * D + q] = ds->xi[q * n + i];
    }
    if (ds->kind == PMDI_O_CATEGORICAL) {
        /* categorical_cluster.jl:8-10 */
        pl->nlevels = (double *)malloc((size_t)D * sizeof(double));
        if (!pl->nlevels) return -1;
        int64_t gmax = 0;
        for (int64_t q = 0; q < D; ++q) {
            int64_t cm = pl->xi[q];
            for (int64_t i = 0; i < n; ++i) {
                int64_t v = pl->xi[i * D + q];
                if (v < 1) return -2;
                if (v > cm) cm = v;
            }
            if (cm > gmax) gmax = cm;
            pl->nlevels[q] = 0.5 * (double)cm;
        }
        pl->L = (int32_t)gmax;
    } else if (ds->kind == PMDI_O_NEGBINOM) {
        for (int64_t i = 0; i < n * D; ++i) if (pl->xi[i] < 0) return -2;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
struct pmdi_oracle {
    int32_t K, N, P;
    int64_t n;
    uint64_t seed;
    int32_t q1_mode, q2_mode, faithful_cost;
    int64_t pool_cap;       /* N*P+1, src/pmdi.jl:140 */
    pool_t *pools;          /* K */
    int64_t *particle;      /* [k][p][n]  = particle[n,p,k], src/pmdi.jl:131 */
    int64_t *particle_id;   /* [k][p]     src/pmdi.jl:132 */
    int64_t *new_id;        /* [k][id][n] = new_id[n,id,k], src/pmdi.jl:135 */
    double  *fprob_dict;    /* [id][N+1]  src/pmdi.jl:133 */
    uint8_t *fprob_done;    /* [P]        src/pmdi.jl:134 */
    double  *fprob;         /* [N]        src/pmdi.jl:102 */
    double  *logprob;       /* [k][pool]  src/pmdi.jl:136 */
    double  *logweight;     /* [P]        src/pmdi.jl:99 */
    uint8_t *cluster_update;/* [pool+1]   src/pmdi.jl:103 */
    int64_t *counts;        /* [k][pool+1] src/pmdi.jl:142 */
    int64_t *sstar_id;      /* [k][p]     src/pmdi.jl:145 */
    uint8_t *sstar;         /* [k][i][p]  = sstar[p,i,k], src/pmdi.jl:146 */
    int64_t *maxid;         /* [k] running maximum(particle_k) */
    int64_t last_updates[8], last_moved[8];   /* per dataset, last sweep: cluster_add! calls at :300, deepcopies at :336 */
    int64_t *dbg_cols;      /* analysis only (scripts/column_stats.py): [step][k] distinct columns of particle[:, :, k] before the ESS test */
    int64_t *dbg_steps;     /* analysis / work-counter checks: [step][k][8], see pmdi_oracle_debug_steps */
    uint8_t *dbg_mark;      /* [pool+1] scratch of the above */
    /* scratch */
    int64_t *partstar, *tmp_i64, *idmap;
    double  *tmp_d;
    uint8_t *tmp_u8;
};

pmdi_oracle *pmdi_oracle_create(int32_t K, int64_t n, int32_t N, int32_t P,
                                const pmdi_oracle_dataset *ds, uint64_t seed,
                                int32_t q1_mode, int32_t q2_mode, int32_t faithful_cost)
{
    /* asserts of src/pmdi.jl:50-55 that concern the sweep */
    if (K < 1 || n < 1 || N < 2 || N > n || N > 255 || P < 2) return NULL;
    pmdi_oracle *h = (pmdi_oracle *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->K = K; h->n = n; h->N = N; h->P = P; h->seed = seed;
    h->q1_mode = q1_mode; h->q2_mode = q2_mode; h->faithful_cost = faithful_cost;
    h->pool_cap = (int64_t)N * P + 1;
    int64_t cap = h->pool_cap;
    h->pools = (pool_t *)calloc((size_t)K, sizeof(pool_t));
    if (!h->pools) goto fail;
    for (int k = 0; k < K; ++k) {
        if (pool_init_data(&h->pools[k], &ds[k], n) != 0) goto fail;
        if (pool_alloc(&h->pools[k], cap) != 0) goto fail;
        /* src/pmdi.jl:120-128: null cluster of all observations, all flags on */
        pool_t *pl = &h->pools[k];
        pl->feature_null = (double *)malloc((size_t)pl->D * sizeof(double));
        if (!pl->feature_null) goto fail;
        cl_reset(pl, 1);
        for (int64_t i = 0; i < n; ++i) cl_add(pl, 1, i, NULL);
        cl_logmarginal(pl, 1, pl->feature_null);
        for (int q = 0; q < pl->D; ++q) pl->feature_null[q] = -pl->feature_null[q];
    }
    h->particle = (int64_t *)malloc((size_t)K * P * N * sizeof(int64_t));
    h->particle_id = (int64_t *)malloc((size_t)K * P * sizeof(int64_t));
    h->new_id = (int64_t *)malloc((size_t)K * P * N * sizeof(int64_t));
    h->fprob_dict = (double *)malloc((size_t)P * (N + 1) * sizeof(double));
    h->fprob_done = (uint8_t *)malloc((size_t)P + 1);
    h->fprob = (double *)malloc((size_t)N * sizeof(double));
    h->logprob = (double *)malloc((size_t)K * (cap + 1) * sizeof(double));
    h->logweight = (double *)calloc((size_t)P, sizeof(double));
    h->cluster_update = (uint8_t *)calloc((size_t)cap + 1, 1);
    h->counts = (int64_t *)calloc((size_t)K * (cap + 1), sizeof(int64_t));
    h->sstar_id = (int64_t *)malloc((size_t)K * P * sizeof(int64_t));
    h->sstar = (uint8_t *)calloc((size_t)K * n * P, 1);
    h->maxid = (int64_t *)calloc((size_t)K, sizeof(int64_t));
    h->partstar = (int64_t *)malloc((size_t)P * sizeof(int64_t));
    h->tmp_i64 = (int64_t *)malloc((size_t)P * N * sizeof(int64_t));
    h->idmap = (int64_t *)malloc((size_t)(cap + 1) * sizeof(int64_t));
    h->tmp_d = (double *)malloc((size_t)(P > N ? P : N) * 2 * sizeof(double));
    h->tmp_u8 = (uint8_t *)malloc((size_t)n * (P > 1 ? 1 : 1));
    if (!h->particle || !h->particle_id || !h->new_id || !h->fprob_dict || !h->fprob_done ||
        !h->fprob || !h->logprob || !h->logweight || !h->cluster_update || !h->counts ||
        !h->sstar_id || !h->sstar || !h->maxid || !h->partstar || !h->tmp_i64 || !h->idmap ||
        !h->tmp_d || !h->tmp_u8)
        goto fail;
    for (int64_t i = 0; i < (int64_t)K * P * N; ++i) h->particle[i] = 1;
    for (int k = 0; k < K; ++k) { h->counts[k * (cap + 1) + 1] = (int64_t)P * N; h->maxid[k] = 1; }
    return h;
fail:
    pmdi_oracle_destroy(h);
    return NULL;
}

void pmdi_oracle_destroy(pmdi_oracle *h)
{
    if (!h) return;
    if (h->pools) for (int k = 0; k < h->K; ++k) pool_free(&h->pools[k]);
    free(h->pools); free(h->particle); free(h->particle_id); free(h->new_id);
    free(h->fprob_dict); free(h->fprob_done); free(h->fprob); free(h->logprob);
    free(h->logweight); free(h->cluster_update); free(h->counts); free(h->sstar_id);
    free(h->sstar); free(h->maxid); free(h->partstar); free(h->tmp_i64); free(h->idmap);
    free(h->tmp_d); free(h->tmp_u8); free(h->dbg_mark);
    free(h);
}

/* ------------------------------------------------------------------------ */
/* src/misc.jl:15-25 */
double pmdi_oracle_calc_ess(const double *logweight, int64_t P)
{
    double num = 0.0, den = 0.0, max_l = logweight[0];
    for (int64_t p = 1; p < P; ++p) if (logweight[p] > max_l) max_l = logweight[p];
    for (int64_t p = 0; p < P; ++p) {
        double w = exp(logweight[p] - max_l);
        num += w;
        den += w * w;
    }
    return (num * num) / den;
}

/* src/misc.jl:27-47.  u01 replaces rand() at :28; uslot replaces the
 * shuffle! at :43 (shuffle!, partstar[1] = 1, sort! == overwrite one
 * uniformly chosen slot with 1, then sort).  partstar is 1-based. */
void pmdi_oracle_draw_partstar(const double *logweight, int64_t P, double u01,
                               double uslot, int64_t *partstar)
{
    double *pprob = (double *)malloc((size_t)P * sizeof(double));
    double max_l = logweight[0];
    for (int64_t p = 1; p < P; ++p) if (logweight[p] > max_l) max_l = logweight[p];
    for (int64_t p = 0; p < P; ++p) pprob[p] = exp(logweight[p] - max_l);
    jl_cumsum(pprob, pprob, P);
    double u = u01 / (double)P;
    double last = pprob[P - 1];
    int64_t i = 0;
    for (int64_t p = 0; p < P; ++p) {
        /* the reference has no bound on i (it would throw past P, an event of
         * probability ~2^-53); the guard keeps this restatement memory-safe */
        while (i < P && pprob[p] / last >= u) {
            u += 1.0 / (double)P;
            partstar[i] = p + 1;
            i += 1;
        }
    }
    for (; i < P; ++i) partstar[i] = P; /* unreachable in exact arithmetic */
    int64_t j = (int64_t)(uslot * (double)P);
    if (j >= P) j = P - 1;
    /* partstar is sorted; put 1 in slot j and re-sort = drop element j,
     * shift the ones before it up by one, 1 in front */
    for (int64_t m = j; m > 0; --m) partstar[m] = partstar[m - 1];
    partstar[0] = 1;
    free(pprob);
}

/* src/misc.jl:1-13 and 50-59; sstar_i is P x K column-major (sstar[:, i, :]) */
void pmdi_oracle_phi_upweight(double *logweight, const int64_t *sstar_i, int32_t K,
                              const double *Phi, int64_t P)
{
    int i = 0;
    for (int k1 = 0; k1 < K - 1; ++k1)
        for (int k2 = k1 + 1; k2 < K; ++k2) {
            double phi_log = log(1.0 + Phi[i]);
            for (int64_t p = 0; p < P; ++p)
                logweight[p] += (double)(sstar_i[k1 * P + p] == sstar_i[k2 * P + p]) * phi_log;
            ++i;
        }
}

/* src/pmdi.jl:345-350 + StatsBase.sample(::Weights) */
static int64_t pick_pstar(const double *logweight, int64_t P, double u01, double *w)
{
    double max_l = logweight[0];
    for (int64_t p = 1; p < P; ++p) if (logweight[p] > max_l) max_l = logweight[p];
    double sum = 0.0;
    for (int64_t p = 0; p < P; ++p) { w[p] = exp(logweight[p] - max_l); sum += w[p]; }
    double t = u01 * sum;
    int64_t i = 0;
    double cw = w[0];
    while (cw < t && i < P - 1) { i += 1; cw += w[i]; }
    return i + 1;
}

static double now_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* distinct columns particle[:, p, k] over the particles: sorted insertion of the columns' 64-bit hashes (analysis only) */
static int64_t count_distinct_columns(pmdi_oracle *h, int k)
{
    const int N = h->N, P = h->P;
    const int64_t *particle = h->particle + (int64_t)k * P * N;
    uint64_t *hs = (uint64_t *)h->tmp_d;          /* P doubles = P hashes */
    for (int p = 0; p < P; ++p) {
        uint64_t x = 1469598103934665603ull;
        for (int nn = 0; nn < N; ++nn) { x ^= (uint64_t)particle[(int64_t)p * N + nn]; x *= 1099511628211ull; x ^= x >> 29; }
        hs[p] = x;
    }
    int64_t nc = 0;
    for (int p = 0; p < P; ++p) {                 /* insertion into the sorted distinct prefix: fine for the small counts of interest */
        int64_t lo = 0, hi = nc;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (hs[mid] < hs[p]) lo = mid + 1; else hi = mid; }
        if (lo < nc && hs[lo] == hs[p]) continue;
        uint64_t v = hs[p];
        memmove(hs + lo + 1, hs + lo, (size_t)(nc - lo) * sizeof(uint64_t));
        hs[lo] = v; nc += 1;
    }
    return nc;
}

/* ------------------------------------------------------------------------ */
int pmdi_oracle_sweep(pmdi_oracle *h, int64_t iter, const int64_t *s_in,
                      const int64_t *order_obs, int64_t n1, const double *Pi,
                      const double *Phi, const uint8_t *const *flags, double lw_init,
                      int64_t *s_out, double *logweight_out, int64_t *p_star_out,
                      pmdi_oracle_stats *stats, double *trace)
{
    const int K = h->K, N = h->N, P = h->P;
    const int64_t n = h->n, cap = h->pool_cap;
    if (n1 < 1 || n1 > n) return -1;  /* rho*n < 1 is UB in the reference */
    for (int k = 0; k < K; ++k)
        for (int64_t i = 0; i < n; ++i)
            if (s_in[k * n + i] < 1 || s_in[k * n + i] > N) return -2;

    double t0 = now_seconds();
    pmdi_oracle_stats st; memset(&st, 0, sizeof(st));
    memset(h->last_updates, 0, sizeof(h->last_updates)); memset(h->last_moved, 0, sizeof(h->last_moved));
    double *logweight = h->logweight;
    for (int p = 0; p < P; ++p) logweight[p] = lw_init;

    /* src/pmdi.jl:165-171 */
    memset(h->counts, 0, (size_t)K * (cap + 1) * sizeof(int64_t));
    for (int k = 0; k < K; ++k) h->counts[k * (cap + 1) + 1] = (int64_t)P * N;
    memset(h->new_id, 0, (size_t)K * P * N * sizeof(int64_t));
    for (int64_t i = 0; i < (int64_t)K * P; ++i) h->particle_id[i] = 1;
    for (int64_t i = 0; i < (int64_t)K * P * N; ++i) h->particle[i] = 1;

    /* known prefix: src/pmdi.jl:188-207 */
    for (int k = 0; k < K; ++k) {
        pool_t *pl = &h->pools[k];
        int64_t *counts = h->counts + k * (cap + 1);
        int64_t *particle = h->particle + (int64_t)k * P * N;
        int64_t clust_ids[256];
        memset(clust_ids, 0, sizeof(clust_ids));
        cl_reset(pl, 1);
        int64_t id = 2;
        for (int64_t j = 0; j < n1 - 1; ++j) {       /* unique(), first appearance */
            int64_t u = s_in[k * n + (order_obs[j] - 1)];
            if (clust_ids[u] == 0) {
                cl_reset(pl, id);
                counts[id] = P;
                counts[1] -= P;
                clust_ids[u] = id;
                for (int p = 0; p < P; ++p) particle[(int64_t)p * N + (u - 1)] = id;
                id += 1;
            }
        }
        for (int64_t j = 0; j < n1 - 1; ++j) {
            int64_t i = order_obs[j] - 1;
            int64_t u = s_in[k * n + i];
            uint8_t *ss = h->sstar + ((int64_t)k * n + i) * P;
            memset(ss, (int)u, (size_t)P);            /* sstar[:, i, k] .= s[i, k] */
            cl_add(pl, clust_ids[u], i, flags[k]);
        }
        h->maxid[k] = id - 1;
        if (h->maxid[k] < 1) h->maxid[k] = 1;
    }

    int64_t n_swept = n - n1 + 1;
    for (int64_t pos = n1 - 1; pos < n; ++pos) {      /* src/pmdi.jl:209 */
        int64_t i = order_obs[pos] - 1;
        for (int k = 0; k < K; ++k) {                 /* src/pmdi.jl:210 */
            pool_t *pl = &h->pools[k];
            int64_t *counts = h->counts + k * (cap + 1);
            int64_t *particle = h->particle + (int64_t)k * P * N;
            int64_t *particle_id = h->particle_id + (int64_t)k * P;
            int64_t *new_id = h->new_id + (int64_t)k * P * N;
            int64_t *sstar_id = h->sstar_id + (int64_t)k * P;
            double *logprob = h->logprob + (int64_t)k * (cap + 1);
            uint8_t *ss = h->sstar + ((int64_t)k * n + i) * P;
            const double *Pi_k = Pi + (int64_t)k * N;
            double *fprob = h->fprob;

            if (h->faithful_cost) memset(h->cluster_update, 0, (size_t)cap + 1); /* :211 */
            memset(h->fprob_done, 0, (size_t)P + 1);                              /* :212 */
            if (h->q1_mode == 1) memset(new_id, 0, (size_t)P * N * sizeof(int64_t));

            int64_t maxid = h->maxid[k];
            if (h->faithful_cost) {                   /* maximum(particle_k), :218 */
                int64_t m = 0;
                for (int64_t e = 0; e < (int64_t)P * N; ++e) if (particle[e] > m) m = particle[e];
                maxid = m;
            }
            for (int64_t id = 1; id <= maxid; ++id) { /* :218-220 */
                logprob[id] = cl_logprob(pl, id, i, flags[k]);
                st.n_operations += 1;                  /* src/__pmdi.jl:187 */
            }

            int64_t *dbg = h->dbg_steps ? h->dbg_steps + ((pos - (n1 - 1)) * K + k) * 8 : NULL;
            if (dbg) { memset(dbg, 0, 8 * sizeof(int64_t)); dbg[6] = maxid; }
            int64_t curr_id = 0;                      /* :222 */
            for (int p = 0; p < P; ++p) {             /* :223 */
                int64_t id = particle_id[p];
                const int64_t *part_p = particle + (int64_t)p * N;
                double *dict = h->fprob_dict + (id - 1) * (N + 1);
                if (dbg && !h->fprob_done[id]) {      /* a class leader: the entries of logprob it reads at :232 */
                    dbg[0] += 1;
                    for (int nn = 0; nn < N; ++nn)
                        if (!h->dbg_mark[part_p[nn]]) { h->dbg_mark[part_p[nn]] = 1; dbg[1] += 1; }
                }
                if (h->fprob_done[id]) {              /* :225-229 */
                    for (int nn = 0; nn < N; ++nn) fprob[nn] = dict[nn];
                    logweight[p] += dict[N];
                } else {                              /* :231-248 */
                    for (int nn = 0; nn < N; ++nn) fprob[nn] = logprob[part_p[nn]];
                    double max_logprob = fprob[0];
                    for (int nn = 1; nn < N; ++nn) if (fprob[nn] > max_logprob) max_logprob = fprob[nn];
                    for (int nn = 0; nn < N; ++nn) {
                        fprob[nn] -= max_logprob;
                        fprob[nn] = exp(fprob[nn]);
                        fprob[nn] *= Pi_k[nn];
                    }
                    jl_cumsum(fprob, fprob, N);       /* cumsum!(fprob, fprob) :240 */
                    double logprob_inc = log(fprob[N - 1]) + max_logprob;
                    dict[N] = logprob_inc;
                    logweight[p] += logprob_inc;
                    double fN = fprob[N - 1];
                    for (int nn = 0; nn < N; ++nn) fprob[nn] = fprob[nn] / fN;
                    for (int nn = 0; nn < N; ++nn) dict[nn] = fprob[nn];
                    h->fprob_done[id] = 1;
                    st.sum_classes += 1;
                }
                int64_t new_s;
                if (p != 0) {                         /* :251-260 */
                    new_s = 1;
                    double u = pmdi_oracle_uniform(h->seed, (uint32_t)iter, (uint32_t)pos,
                                                   (uint32_t)k, (uint32_t)p, SITE_DRAW);
                    for (int c = 1; c <= N - 1; ++c) {
                        if (fprob[new_s - 1] > u) break;
                        new_s += 1;
                    }
                } else {
                    new_s = s_in[k * n + i];          /* :262 reference trajectory */
                }
                sstar_id[p] = part_p[new_s - 1];      /* :264 */
                ss[p] = (uint8_t)new_s;               /* :265 */
                int64_t *slot = &new_id[(id - 1) * N + (new_s - 1)];
                if (*slot == 0) {                     /* :266-272 */
                    curr_id += 1;
                    *slot = curr_id;
                    particle_id[p] = curr_id;
                } else {
                    particle_id[p] = *slot;
                }
            }

            if (dbg) {
                /* clear the marks through the leaders (lowest particle of every class at the start of the step); fprob_done
                 * is indexed by the class ids of the step's start, which particle_id no longer holds: walk the table instead */
                for (int64_t e = 0; e < (int64_t)P * N; ++e) h->dbg_mark[particle[e]] = 0;
                int un = 1;
                for (int p = 1; p < P; ++p) if (ss[p] != ss[0] || sstar_id[p] != sstar_id[0]) { un = 0; break; }
                dbg[7] = un;
            }
            /* copy-on-write update: src/pmdi.jl:275-310 */
            int64_t max_k = maxid;
            if (h->faithful_cost) {
                int64_t m = 0;
                for (int64_t e = 0; e < (int64_t)P * N; ++e) if (particle[e] > m) m = particle[e];
                max_k = m;
            }
            for (int pp = 0; pp < P; ++pp) {
                int64_t c = sstar_id[pp];
                if (h->cluster_update[c]) continue;
                h->cluster_update[c] = 1;
                int64_t ncopies = 0;
                for (int q = 0; q < P; ++q) ncopies += (sstar_id[q] == c);
                int64_t id;
                if (ncopies == counts[c]) {
                    id = c;
                } else {
                    id = max_k + 1;
                    if (id > cap) return -3;
                    counts[c] -= ncopies;
                    counts[id] = ncopies;
                    cl_copy(pl, id, c);
                    max_k += 1;
                    st.n_clones += 1;
                }
                cl_add(pl, id, i, flags[k]);          /* :300 */
                h->last_updates[k] += 1;
                if (dbg) { dbg[2] += 1; dbg[3] += (id != c); }
                if (id != c) {                        /* :301-308 */
                    for (int part = 0; part < P; ++part) {
                        int64_t s_id = ss[part];
                        if (particle[(int64_t)part * N + (s_id - 1)] == c)
                            particle[(int64_t)part * N + (s_id - 1)] = id;
                    }
                }
            }
            if (!h->faithful_cost) {
                for (int pp = 0; pp < P; ++pp) h->cluster_update[sstar_id[pp]] = 0;
            }
            h->maxid[k] = max_k;
            if (max_k > st.max_id) st.max_id = max_k;
        }

        if (K > 1) {                                  /* :312-314 */
            int pair = 0;
            for (int k1 = 0; k1 < K - 1; ++k1)
                for (int k2 = k1 + 1; k2 < K; ++k2) {
                    double phi_log = log(1.0 + Phi[pair]);
                    const uint8_t *a = h->sstar + ((int64_t)k1 * n + i) * P;
                    const uint8_t *b = h->sstar + ((int64_t)k2 * n + i) * P;
                    for (int p = 0; p < P; ++p) logweight[p] += (double)(a[p] == b[p]) * phi_log;
                    ++pair;
                }
        }

        if (h->dbg_cols)
            for (int k = 0; k < K; ++k) h->dbg_cols[(pos - (n1 - 1)) * K + k] = count_distinct_columns(h, k);
        if (h->dbg_steps)
            for (int k = 0; k < K; ++k) h->dbg_steps[((pos - (n1 - 1)) * K + k) * 8 + 4] = count_distinct_columns(h, k);
        double ess = pmdi_oracle_calc_ess(logweight, P);
        int resampled = 0;
        if (ess <= 0.5 * (double)P) {                 /* :317 */
            resampled = 1;
            st.n_resamples += 1;
            double u01 = pmdi_oracle_uniform(h->seed, (uint32_t)iter, (uint32_t)pos, 0, 0, SITE_RESAMPLE_U);
            double usl = pmdi_oracle_uniform(h->seed, (uint32_t)iter, (uint32_t)pos, 0, 0, SITE_RESAMPLE_SLOT);
            int64_t *partstar = h->partstar;
            pmdi_oracle_draw_partstar(logweight, P, u01, usl, partstar);
            for (int p = 0; p < P; ++p) logweight[p] = 1.0;   /* :319 */
            for (int k = 0; k < K; ++k) {             /* :320-340 */
                pool_t *pl = &h->pools[k];
                int64_t *counts = h->counts + k * (cap + 1);
                int64_t *particle = h->particle + (int64_t)k * P * N;
                int64_t *particle_id = h->particle_id + (int64_t)k * P;
                int64_t *tmp = h->tmp_i64;
                for (int p = 0; p < P; ++p)
                    memcpy(tmp + (int64_t)p * N, particle + (partstar[p] - 1) * N, (size_t)N * sizeof(int64_t));
                memcpy(particle, tmp, (size_t)P * N * sizeof(int64_t));
                for (int p = 0; p < P; ++p) tmp[p] = particle_id[partstar[p] - 1];
                memcpy(particle_id, tmp, (size_t)P * sizeof(int64_t));
                if (h->q2_mode == 1) {                /* src/__pmdi.jl:285 */
                    uint8_t *col = h->tmp_u8;
                    (void)col;
                    for (int64_t ii = 0; ii < n; ++ii) {
                        uint8_t *ssi = h->sstar + ((int64_t)k * n + ii) * P;
                        uint8_t buf[4096];
                        uint8_t *b = (P <= 4096) ? buf : (uint8_t *)malloc((size_t)P);
                        for (int p = 0; p < P; ++p) b[p] = ssi[partstar[p] - 1];
                        memcpy(ssi, b, (size_t)P);
                        if (b != buf) free(b);
                    }
                }
                /* compact renumbering, :326-339 */
                int64_t old_max = h->maxid[k];
                if (h->faithful_cost) memset(counts, 0, (size_t)(cap + 1) * sizeof(int64_t));
                else memset(counts, 0, (size_t)(old_max + 1) * sizeof(int64_t));
                int64_t *idmap = h->idmap;
                memset(idmap, 0, (size_t)(old_max + 1) * sizeof(int64_t));
                for (int64_t e = 0; e < (int64_t)P * N; ++e) idmap[particle[e]] = 1;
                int64_t next = 0;
                for (int64_t id = 1; id <= old_max; ++id) {  /* sort(unique(..)) ascending */
                    if (!idmap[id]) continue;
                    next += 1;
                    idmap[id] = next;
                    if (id != next) { cl_copy(pl, next, id); h->last_moved[k] += 1; }   /* :336 */
                }
                for (int64_t e = 0; e < (int64_t)P * N; ++e) {
                    particle[e] = idmap[particle[e]];
                    counts[particle[e]] += 1;                /* :338 */
                }
                h->maxid[k] = next;
            }
        }
        if (h->dbg_steps)   /* distinct columns after the (possible) resampling: what the next step starts from */
            for (int k = 0; k < K; ++k)
                h->dbg_steps[((pos - (n1 - 1)) * K + k) * 8 + 5] = resampled ? count_distinct_columns(h, k) : h->dbg_steps[((pos - (n1 - 1)) * K + k) * 8 + 4];
        if (trace) {
            double *tr = trace + (pos - (n1 - 1)) * (2 + 2 * K);
            tr[0] = ess; tr[1] = (double)resampled;
            for (int k = 0; k < K; ++k) {
                tr[2 + k] = (double)h->maxid[k];
                /* distinct classes after this step */
                int64_t *pid = h->particle_id + (int64_t)k * P;
                memset(h->fprob_done, 0, (size_t)P + 1);
                int64_t nc = 0;
                for (int p = 0; p < P; ++p) if (!h->fprob_done[pid[p]]) { h->fprob_done[pid[p]] = 1; nc++; }
                tr[2 + K + k] = (double)nc;
            }
        }
    }
    (void)n_swept;

    /* src/pmdi.jl:345-350 */
    double ups = pmdi_oracle_uniform(h->seed, (uint32_t)iter, 0, 0, 0, SITE_PSTAR);
    int64_t p_star = pick_pstar(logweight, P, ups, h->tmp_d);
    /* src/pmdi.jl:373: s[:] = sstar[p_star, :, :] */
    for (int k = 0; k < K; ++k)
        for (int64_t i = 0; i < n; ++i)
            s_out[k * n + i] = h->sstar[((int64_t)k * n + i) * P + (p_star - 1)];
    if (logweight_out) memcpy(logweight_out, logweight, (size_t)P * sizeof(double));
    if (p_star_out) *p_star_out = p_star;
    st.seconds = now_seconds() - t0;
    if (stats) *stats = st;
    return 0;
}

/* src/pmdi.jl:354-370 */
int pmdi_oracle_feature_select(pmdi_oracle *h, int64_t iter, const int64_t *s_traj,
                               uint8_t *const *flags_out, double *const *feature_prob)
{
    const int K = h->K, N = h->N;
    const int64_t n = h->n;
    for (int k = 0; k < K; ++k) {
        pool_t *pl = &h->pools[k];
        int D = pl->D;
        double *fp = (double *)malloc((size_t)D * sizeof(double));
        double *lm = (double *)malloc((size_t)D * sizeof(double));
        if (!fp || !lm) { free(fp); free(lm); return -1; }
        for (int q = 0; q < D; ++q) fp[q] = pl->feature_null[q] + 0.0;   /* :357 */
        uint8_t seen[256]; memset(seen, 0, sizeof(seen));
        /* pool slot cap is scratch here: the sweep state is rebuilt next iteration */
        int64_t scratch = pl->cap;
        for (int64_t i = 0; i < n; ++i) {             /* unique(), first appearance :358 */
            int64_t c = s_traj[k * n + i];
            if (c < 1 || c > N) { free(fp); free(lm); return -2; }
            if (seen[c]) continue;
            seen[c] = 1;
            cl_reset(pl, scratch);                    /* :361 */
            for (int64_t j = 0; j < n; ++j)           /* findindices ascending :360 */
                if (s_traj[k * n + j] == c) cl_add(pl, scratch, j, NULL);  /* :363 */
            cl_logmarginal(pl, scratch, lm);
            for (int q = 0; q < D; ++q) fp[q] += lm[q];   /* :365 */
        }
        for (int q = 0; q < D; ++q) {                 /* :367 */
            double r = pmdi_oracle_uniform(h->seed, (uint32_t)iter, 0, (uint32_t)k, (uint32_t)q, SITE_FEATSEL);
            double pr = 1.0 - 1.0 / (exp(fp[q] + 1.0));
            flags_out[k][q] = (uint8_t)(pr > r);
            if (feature_prob && feature_prob[k]) feature_prob[k][q] = fp[q];
        }
        free(fp); free(lm);
    }
    return 0;
}

void pmdi_oracle_debug_columns(pmdi_oracle *h, int64_t *buf) { h->dbg_cols = buf; }

/* Per (swept observation, dataset) record of the next sweeps, buf[step][k][8] (NULL switches it off):
 *   0 particle classes at the start of the step (CDFs formed, :231-248)   1 distinct clusters those classes' leaders read at :232
 *   2 distinct chosen clusters (cluster_add! calls, :300)                 3 of which cloned (:292-298)
 *   4 distinct columns of particle[:, :, k] before the ESS test           5 ... after the resampling of this observation, if any
 *   6 maximum(particle_k) at the start of the step                        7 1 if every particle drew the same label and cluster
 * What the device's work counters (clusters evaluated, columns met by resampling events, copy-on-write splits) are checked against. */
int pmdi_oracle_debug_steps(pmdi_oracle *h, int64_t *buf)
{
    h->dbg_steps = buf;
    if (buf && !h->dbg_mark) {
        h->dbg_mark = (uint8_t *)calloc((size_t)h->pool_cap + 2, 1);
        if (!h->dbg_mark) { h->dbg_steps = NULL; return -1; }
    }
    return 0;
}

int pmdi_oracle_export(const pmdi_oracle *h, int64_t *particle, int64_t *counts,
                       int64_t *cluster_n, int64_t *max_id)
{
    const int K = h->K, N = h->N, P = h->P;
    const int64_t cap = h->pool_cap;
    if (particle) memcpy(particle, h->particle, (size_t)K * P * N * sizeof(int64_t));
    for (int k = 0; k < K; ++k) {
        if (counts)
            for (int64_t id = 1; id <= cap; ++id) counts[k * cap + (id - 1)] = h->counts[k * (cap + 1) + id];
        if (cluster_n)
            for (int64_t id = 1; id <= cap; ++id) cluster_n[k * cap + (id - 1)] = h->pools[k].cn[id];
        if (max_id) max_id[k] = h->maxid[k];
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
struct pmdi_oracle_cluster { pool_t pl; };

pmdi_oracle_cluster *pmdi_oracle_cluster_new(const pmdi_oracle_dataset *ds, int64_t n)
{
    pmdi_oracle_cluster *c = (pmdi_oracle_cluster *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    if (pool_init_data(&c->pl, ds, n) != 0 || pool_alloc(&c->pl, 1) != 0) {
        pool_free(&c->pl); free(c); return NULL;
    }
    cl_reset(&c->pl, 1);
    return c;
}
void pmdi_oracle_cluster_free(pmdi_oracle_cluster *c) { if (c) { pool_free(&c->pl); free(c); } }
void pmdi_oracle_cluster_add(pmdi_oracle_cluster *c, int64_t row, const uint8_t *flag) { cl_add(&c->pl, 1, row, flag); }
double pmdi_oracle_cluster_logprob(const pmdi_oracle_cluster *c, int64_t row, const uint8_t *flag) { return cl_logprob(&c->pl, 1, row, flag); }
void pmdi_oracle_cluster_logmarginal(const pmdi_oracle_cluster *c, double *out) { cl_logmarginal(&c->pl, 1, out); }
int64_t pmdi_oracle_cluster_stats(const pmdi_oracle_cluster *c, double *out)
{
    const pool_t *pl = &c->pl;
    int64_t D = pl->D, m = 0;
    out[m++] = (double)pl->cn[1];
    if (pl->kind == PMDI_O_GAUSSIAN) {
        for (int64_t q = 0; q < D; ++q) out[m++] = pl->mu[D + q];
        for (int64_t q = 0; q < D; ++q) out[m++] = pl->sum[D + q];
        for (int64_t q = 0; q < D; ++q) out[m++] = pl->lam[D + q];
        for (int64_t q = 0; q < D; ++q) out[m++] = pl->beta[D + q];
    } else if (pl->kind == PMDI_O_CATEGORICAL) {
        for (int64_t q = 0; q < D; ++q)
            for (int64_t l = 0; l < pl->L; ++l) out[m++] = (double)pl->cnt[(D + q) * pl->L + l];
    } else {
        for (int64_t q = 0; q < D; ++q) out[m++] = (double)pl->nbsum[D + q];
    }
    return m;
}

/* output_analysis/consensus_map.jl:50-56 (generate_psm): the triple loop k, j, i with
 * sum(output[:, i] .== output[:, j]); restated for a block of rows, full rows. */
void pmdi_oracle_psm_counts(const uint8_t *samples, int64_t S, int32_t K, int64_t n,
                            int64_t row_lo, int64_t row_hi, int32_t *counts)
{
    for (int32_t k = 0; k < K; ++k)
        for (int64_t i = row_lo; i < row_hi; ++i)
            for (int64_t j = 0; j < n; ++j) {
                int32_t c = 0;
                for (int64_t t = 0; t < S; ++t) {
                    const uint8_t *row = samples + ((size_t)t * K + k) * n;
                    c += (row[i] == row[j]) ? 1 : 0;
                }
                counts[((size_t)k * (row_hi - row_lo) + (i - row_lo)) * n + j] = c;
            }
}

/* per dataset, last sweep: distinct clusters updated (cluster_add! at src/pmdi.jl:300) and clusters moved down by the
 * renumbering of the resampling events (deepcopy at :336) -- what the device's work counters are checked against */
void pmdi_oracle_work(const pmdi_oracle *h, int64_t *updates, int64_t *moved)
{
    for (int k = 0; k < h->K; ++k) { updates[k] = h->last_updates[k]; moved[k] = h->last_moved[k]; }
}
