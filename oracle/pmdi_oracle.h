/*
 * pmdi_oracle.h -- CPU restatement of ParticleMDI's conditional-SMC sweep.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * import, link or execute it, and only as the checker / timed CPU baseline.
 *
 * Parity status: the reference (Julia, /root/reference) cannot be run in
 * this container or on the GPU box (no julia binary), and its test-suite
 * holds no golden vectors or RNG seeds.  The restatement is pinned by the
 * reference's analytic tests T1/T2 (test/runtests.jl:11-54) and by the T5
 * structural invariants (test/runtests.jl:136-162); everything else
 * (NegBinom, calc_logmarginal, feature selection, Phi_upweight!, calc_ESS,
 * draw_partstar) is "parity unpinned" by the reference and is pinned here
 * by independent scipy known-answer tests (tests/test_oracle_*.py).
 *
 * Each function cites the reference file:line it restates (paths relative
 * to /root/reference).
 */
#ifndef PMDI_ORACLE_H
#define PMDI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PMDI_O_GAUSSIAN = 0, PMDI_O_CATEGORICAL = 1, PMDI_O_NEGBINOM = 2 };

/* One dataset: an n x D column-major matrix (Julia layout, src/pmdi.jl:43).
 * Gaussian uses xf (Float64); Categorical (levels 1..L) and NegBinom
 * (counts >= 0) use xi (Int64). */
typedef struct {
    int32_t kind;
    int32_t D;
    const double  *xf;
    const int64_t *xi;
} pmdi_oracle_dataset;

typedef struct pmdi_oracle pmdi_oracle;

typedef struct {
    int64_t n_operations;  /* calc_logprob evaluations, src/__pmdi.jl:187 */
    int64_t n_resamples;
    int64_t n_clones;      /* deepcopy at src/pmdi.jl:297 */
    int64_t max_id;        /* largest pool id ever live in this sweep */
    int64_t sum_classes;   /* sum over steps of distinct particle classes */
    double  seconds;
} pmdi_oracle_stats;

/* q1_mode: 0 = reference (new_id zeroed once per Gibbs iteration,
 *              src/pmdi.jl:167), 1 = corrected (zeroed per step).
 * q2_mode: 0 = pmdi() (allocation history NOT permuted on resample,
 *              src/pmdi.jl:321-324), 1 = __pmdi() (src/__pmdi.jl:285).
 * faithful_cost: 1 keeps the reference's O(N*P) per-step scans
 *              (fill!, maximum, count) so that timing the oracle is a fair
 *              CPU baseline; 0 tracks them incrementally (same results). */
pmdi_oracle *pmdi_oracle_create(int32_t K, int64_t n, int32_t N, int32_t P,
                                const pmdi_oracle_dataset *ds, uint64_t seed,
                                int32_t q1_mode, int32_t q2_mode,
                                int32_t faithful_cost);
void pmdi_oracle_destroy(pmdi_oracle *h);

/* One Gibbs iteration's sweep: src/pmdi.jl:165-171, 188-350, 373.
 *   iter        1-based Gibbs iteration (keys the RNG)
 *   s_in/s_out  n x K column-major, labels 1..N
 *   order_obs   n entries, 1-based permutation (src/pmdi.jl:172 is host-side)
 *   n1          floor(rho*n), must be >= 1
 *   Pi          N x K column-major,  Phi  max(1,K(K-1)/2)
 *   flags       K pointers to D_k bytes (0/1)
 *   lw_init     initial log-weight (0.0 on the first iteration, 1.0 after:
 *               src/pmdi.jl:99,372)
 *   trace       optional, n_s x (2+2K) doubles per swept observation:
 *               [ESS, resampled, maxid_k..., nclasses_k...]
 */
int pmdi_oracle_sweep(pmdi_oracle *h, int64_t iter, const int64_t *s_in,
                      const int64_t *order_obs, int64_t n1, const double *Pi,
                      const double *Phi, const uint8_t *const *flags,
                      double lw_init, int64_t *s_out, double *logweight,
                      int64_t *p_star, pmdi_oracle_stats *stats, double *trace);

/* Feature selection after a sweep: src/pmdi.jl:120-128 (null marginal, at
 * create time) and 354-370.  s_traj is the n x K selected trajectory
 * (sstar[p_star,:,:]).  Writes flags_out[k][q] and feature_prob[k][q]. */
int pmdi_oracle_feature_select(pmdi_oracle *h, int64_t iter,
                               const int64_t *s_traj, uint8_t *const *flags_out,
                               double *const *feature_prob);

/* State after the last sweep (T5 invariants, test/runtests.jl:147-162).
 * particle: N x P x K (column-major, 1-based ids); counts: (N*P+1) x K;
 * cluster_n: (N*P+1) x K (only ids <= max are meaningful). */
int pmdi_oracle_export(const pmdi_oracle *h, int64_t *particle, int64_t *counts,
                       int64_t *cluster_n, int64_t *max_id);

/* Per dataset, last sweep: distinct clusters updated (cluster_add! at src/pmdi.jl:300) and clusters moved down by the
 * renumbering of resampling events (deepcopy at :336). */
void pmdi_oracle_work(const pmdi_oracle *h, int64_t *updates, int64_t *moved);
/* per-step analysis record [step][k][8] of the next sweeps (see pmdi_oracle.c); NULL = off */
int pmdi_oracle_debug_steps(pmdi_oracle *h, int64_t *buf);

/* --- cluster plugin protocol on stand-alone clusters (unit tests) -------- */
typedef struct pmdi_oracle_cluster pmdi_oracle_cluster;
pmdi_oracle_cluster *pmdi_oracle_cluster_new(const pmdi_oracle_dataset *ds, int64_t n);
void pmdi_oracle_cluster_free(pmdi_oracle_cluster *c);
void pmdi_oracle_cluster_add(pmdi_oracle_cluster *c, int64_t row, const uint8_t *flag);
double pmdi_oracle_cluster_logprob(const pmdi_oracle_cluster *c, int64_t row, const uint8_t *flag);
void pmdi_oracle_cluster_logmarginal(const pmdi_oracle_cluster *c, double *out);
/* stats: Gaussian -> n, then mu[D], Sigma[D], lambda[D], beta[D];
 *        Categorical -> n, then counts[L x D] column-major; NegBinom -> n, Sigma[D] */
int64_t pmdi_oracle_cluster_stats(const pmdi_oracle_cluster *c, double *out);

/* --- helpers restated from src/misc.jl ----------------------------------- */
double pmdi_oracle_calc_ess(const double *logweight, int64_t P);           /* misc.jl:15-25 */
void pmdi_oracle_draw_partstar(const double *logweight, int64_t P, double u01,
                               double uslot, int64_t *partstar);           /* misc.jl:27-47 */
void pmdi_oracle_phi_upweight(double *logweight, const int64_t *sstar_i,
                              int32_t K, const double *Phi, int64_t P);    /* misc.jl:50-59 */

/* --- the co-clustering counts behind generate_psm -----------
 * output_analysis/consensus_map.jl:50-56: psm[k][i, j] = sum(output[:, i] .== output[:, j]) / n_iter.
 * samples: [S][K][n] labels (any byte values), counts: [K][row_hi-row_lo][n] with
 * counts[k][i-row_lo][j] = #{t : samples[t][k][i] == samples[t][k][j]} (full rows; the reference
 * fills i > j only and the caller masks). */
void pmdi_oracle_psm_counts(const uint8_t *samples, int64_t S, int32_t K, int64_t n,
                            int64_t row_lo, int64_t row_hi, int32_t *counts);

/* --- the per-iteration host work around the sweep (oracle/pmdi_oracle_hypers.c) -------------
 * Literal restatement (N^K tables and all) of src/update_hypers.jl, align_labels! (src/misc.jl:61-96),
 * shuffle!(order_obs) (src/pmdi.jl:172) and the initialisation of src/pmdi.jl:59-96, driven by
 * counter-based Philox draws.  Checks the factorised device kernels of csrc/pmdi_hypers.hip. */
typedef struct pmdi_oracle_hypers pmdi_oracle_hypers;
pmdi_oracle_hypers *pmdi_oracle_hypers_create(int64_t n, int32_t N, int32_t K, uint64_t seed); /* pmdi.jl:59-96 */
void pmdi_oracle_hypers_destroy(pmdi_oracle_hypers *h);
void pmdi_oracle_hypers_update_M(pmdi_oracle_hypers *h, int64_t iter);      /* update_hypers.jl:5-26 */
void pmdi_oracle_hypers_update_gamma(pmdi_oracle_hypers *h, int64_t iter);  /* update_hypers.jl:64-92 */
void pmdi_oracle_hypers_update_Phi(pmdi_oracle_hypers *h, int64_t iter);    /* update_hypers.jl:95-128 */
double pmdi_oracle_hypers_update_Z(pmdi_oracle_hypers *h);                  /* update_hypers.jl:29-39 */
double pmdi_oracle_hypers_update_v(pmdi_oracle_hypers *h, int64_t iter);    /* update_hypers.jl:1-3 */
void pmdi_oracle_hypers_align_labels(pmdi_oracle_hypers *h, int64_t iter);  /* misc.jl:61-96 */
void pmdi_oracle_hypers_shuffle(pmdi_oracle_hypers *h, int64_t iter);       /* pmdi.jl:172 */
/* pmdi.jl:172-185 in pmdi()'s order; Pi: N x K column-major */
void pmdi_oracle_hypers_step(pmdi_oracle_hypers *h, int64_t iter, double *Pi);
/* what: 0 M[K], 1 gamma[N x K], 2 Phi[npairs], 3 (v, Z), 4 gamma0 = exp(Gamma_c) (the stale table) */
int pmdi_oracle_hypers_get(const pmdi_oracle_hypers *h, int what, double *out);
int pmdi_oracle_hypers_set(pmdi_oracle_hypers *h, int what, const double *in);
int64_t *pmdi_oracle_hypers_s(pmdi_oracle_hypers *h);      /* n x K column-major, labels 1..N (live pointer) */
int64_t *pmdi_oracle_hypers_order(pmdi_oracle_hypers *h);  /* n, 1-based (live pointer) */
/* samplers: the specification shared with the HIP path */
double pmdi_oracle_normal(uint64_t seed, uint32_t iter, uint32_t pos, uint32_t k, uint32_t p0, uint32_t site);
double pmdi_oracle_gamma(double shape, uint64_t seed, uint32_t iter, uint32_t pos, uint32_t k, uint32_t site);

/* --- counter-based RNG shared (as a specification) with the HIP path ---- */
void pmdi_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double pmdi_oracle_uniform(uint64_t seed, uint32_t iter, uint32_t pos, uint32_t k,
                           uint32_t p, uint32_t site);

#ifdef __cplusplus
}
#endif
#endif
