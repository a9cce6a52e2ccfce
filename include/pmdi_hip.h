/*
 * pmdi_hip.h -- C ABI of libpmdi_hip.so, the MI355X (gfx950) implementation
 * of ParticleMDI's per-Gibbs-iteration conditional-SMC sweep.
 *
 * The reference (pure Julia) has no FFI and no function seam around this
 * path: it is a block inside pmdi() (src/pmdi.jl:164-384).  Each entry point
 * below names the reference lines it replaces; INTEGRATION.md shows the
 * `ccall` stubs a maintainer would add to src/pmdi.jl to bind them.
 *
 * Conventions (the reference's, so that Julia arrays pass through as-is):
 *   - matrices are column-major; labels, cluster ids, particle and
 *     observation indices are 1-based Int64; reals are Float64
 *   - every function returns 0 on success or a negative PMDI_E_* code;
 *     pmdi_last_error() returns a message for the calling thread
 *   - no callbacks, no exceptions across the ABI, no global state: one handle
 *     = a batch of independent chains on one device and one HIP stream;
 *     a handle is not thread-safe, distinct handles are independent
 *   - the library never returns memory the caller must free, and keeps no
 *     pointer to caller memory after a call returns (data are copied to the
 *     device once, in pmdi_create)
 *   - there is NO CPU fallback: without a usable gfx950 device pmdi_create
 *     fails with PMDI_E_DEVICE.
 */
#ifndef PMDI_HIP_H
#define PMDI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMDI_ABI_VERSION 2
#define PMDI_KMAX 8 /* datasets per handle */
/* Other limits of this build (pmdi_create rejects what exceeds them with PMDI_E_ARG / PMDI_E_DATA): N <= 255 clusters (the reference:
 * N <= n, src/pmdi.jl:54; labels travel as bytes), categorical levels <= 4096 per feature (host-built log tables), P <= 1048575. */

/* dataTypes[k] of pmdi(): GaussianCluster (src/datatypes/gaussian_cluster.jl),
 * CategoricalCluster (categorical_cluster.jl), NegBinomCluster (negbinom_cluster.jl) */
enum { PMDI_GAUSSIAN = 0, PMDI_CATEGORICAL = 1, PMDI_NEGBINOM = 2 };

enum {
    PMDI_OK = 0,
    PMDI_E_ARG = -1,      /* an @assert of src/pmdi.jl:50-55 would have fired, or bad pointer */
    PMDI_E_DEVICE = -2,   /* no gfx950 device / HIP error */
    PMDI_E_MEMORY = -3,
    PMDI_E_POOL = -4,     /* cluster pool capacity exceeded (only if pool_cap < N*P+1) */
    PMDI_E_DATA = -5,     /* categorical level < 1, negative count, label outside 1..N */
    PMDI_E_STATE = -6
};

/* dataFiles[k]: an n x D column-major matrix with leading dimension ld >= n
 * (src/pmdi.jl:42-43).  Gaussian: Float64 (xf).  Categorical: Int64 levels
 * 1..L.  NegBinom: Int64 counts >= 0 (xi). */
typedef struct {
    int32_t kind;
    int32_t D;
    int64_t ld;
    const double  *xf;
    const int64_t *xi;
} pmdi_dataset;

/* Kernel-selection and sizing knobs of pmdi_create (results never depend on them).  Every field: -1 = automatic.  The library
 * itself never reads the environment -- one handle's behaviour depends on what its creator passed, on nothing process-wide.
 * A caller that wants the PMDI_* environment variables (the test-suite, bench.py and the profiling scripts do, through
 * particlemdi.jl_amd/_lib.py) fills the struct with pmdi_tuning_from_env() and passes it in pmdi_config.tuning; NULL = all automatic. */
typedef struct {
    int32_t settled;        /* PMDI_SETTLED        0 never / 1 light chains after their first sweep (automatic) / 2 every chain in every
                             *                     sweep (tests) use the settled-chain kernel where the handle has it: pmdi_settled_kernel() */
    int32_t continue_inplace; /* PMDI_CONTINUE     1 (automatic): a chain that kernel cannot carry is carried on by the general kernel's code in
                             *                     the same workgroup from that observation; 0: swept again from the start behind the launch */
    int32_t sticky;         /* PMDI_STICKY         sweeps a handed-over chain starts on the general kernel afterwards (automatic 3) */
    int32_t light_ids;      /* PMDI_LIGHT_IDS      a chain whose last sweep met at most this many live clusters per step is "light"
                             *                     (automatic: that kernel's LDS id capacity where the handle has it, else 40) */
    int32_t s2_cols, s2_idcap, s2_cls; /* PMDI_S2_COLS / _IDCAP / _CLS  columns, cluster ids, particle classes per dataset its LDS tables hold
                             *                     (automatic 64 / 128 / 32, shrunk by pmdi_create to the LDS budget) */
    int32_t ksplit;         /* PMDI_KSPLIT         K > 1: 0 one workgroup per chain (throughput form) / 1 K cooperating workgroups per chain
                             *                     (latency form); automatic: split while n_chains * K workgroups are resident at once and the
                             *                     handle cannot have the settled-chain kernel (whose single workgroup is the faster form) */
    int32_t requeue_ksplit; /* PMDI_REQUEUE_KSPLIT (continue_inplace = 0 only) 0: re-run given-back chains in one workgroup instead of K */
    int32_t split;          /* PMDI_SPLIT          0: one launch per sweep instead of the heaviest / heavy / light launches */
    int32_t heavy_threads;  /* PMDI_HEAVY_T        workgroup width of the heavy group (512 or 1024) */
    int32_t two_per_cu;     /* PMDI_TWO_PER_CU     0: 256-register builds everywhere (one wide chain per CU) */
    int32_t very_heavy;     /* PMDI_VERY_HEAVY     how many of the heaviest chains get a CU each (automatic: 0 with the settled-chain kernel, else 128) */
    int32_t start_gate;     /* PMDI_START_GATE     0: do not hold the heavy / light launches until the heaviest chains' workgroups are placed */
    int32_t terms_cap;      /* PMDI_TERMS_CAP      LDS doubles for the per-feature terms */
    int32_t lds_target;     /* PMDI_LDS_TARGET     LDS bytes per workgroup above which the per-particle tables move to global memory */
    int32_t phase_timers;   /* PMDI_PHASE_TIMERS   1: per-stage shader-clock timers (pmdi_phase_timers) */
    int32_t profiled;       /* 1: a counter-collecting profiler is attached (rocprofv3 --pmc runs the queues one kernel at a time: a launch
                             *                     that waits for another launch's workgroups would never start -- no start gate then) */
    int32_t ticket;         /* PMDI_TICKET         0: workgroup b of the settled-chain launch sweeps chain order[b]; automatic (1): it draws its
                             *                     position in the launch order from a counter, so the next chain goes to whichever workgroup slot
                             *                     of the GPU frees first (the hardware deals block indices to 32 dispatch queues statically) */
    int32_t reserved[5];    /* -1 */
} pmdi_tuning;

typedef struct {
    int32_t abi_version;   /* PMDI_ABI_VERSION */
    int32_t device;        /* HIP device ordinal */
    int32_t K;             /* length(dataFiles)          src/pmdi.jl:42 */
    int32_t N;             /* max clusters               src/pmdi.jl:36 */
    int32_t P;             /* particles                  src/pmdi.jl:36 */
    int32_t n_chains;      /* independent chains swept per call (>= 1) */
    int64_t n;             /* n_obs                      src/pmdi.jl:43 */
    uint64_t seed;         /* chain c uses seed + c */
    int32_t q1_mode;       /* 0 reference: new_id zeroed per iteration (src/pmdi.jl:167); 1: per step */
    int32_t q2_mode;       /* 0 pmdi(): history not permuted on resample (src/pmdi.jl:321-324); 1 __pmdi() (src/__pmdi.jl:285) */
    int64_t pool_cap;      /* cluster pool ids per dataset; 0 = N*P+1 (src/pmdi.jl:140) */
    int32_t block_threads; /* 0 = choose (and split the chains of a sweep into concurrent launches by weight); else 128/256/512/1024 */
    int32_t reserved;
    const pmdi_tuning *tuning; /* NULL = all automatic (read during pmdi_create only) */
} pmdi_config;

typedef struct pmdi_handle pmdi_handle;

/* Per-chain counters of the last sweep. */
typedef struct {
    int64_t n_operations;  /* calc_logprob evaluations as counted by src/__pmdi.jl:187 */
    int64_t n_resamples;   /* src/pmdi.jl:317 taken */
    int64_t n_clones;      /* deepcopy at src/pmdi.jl:297 */
    int64_t max_id;        /* largest pool id live during the sweep */
    int64_t sum_classes;   /* mutation CDFs computed (fprob_done misses, src/pmdi.jl:231) */
    int64_t steps_fast;    /* steps whose working set fit the LDS tables */
    int64_t steps_converted; /* steps that overflowed the LDS census and finished on the fallback */
    int64_t steps_fallback;  /* steps run on the global-memory fallback (burn-in) */
} pmdi_sweep_stats;

/* Replaces the allocations of src/pmdi.jl:99-146 and the null-cluster
 * marginal of :120-128.  Copies the data to the device (row-major). */
int pmdi_create(const pmdi_config *cfg, const pmdi_dataset *datasets, pmdi_handle **out);

/* every field of *t = -1 (automatic) */
void pmdi_tuning_default(pmdi_tuning *t);
/* pmdi_tuning_default, then every PMDI_* environment variable that is set (names beside the fields above); `profiled` = a rocprof
 * tool library is preloaded.  The only place of the library that reads the environment, and only when the caller asks. */
void pmdi_tuning_from_env(pmdi_tuning *t);
int pmdi_destroy(pmdi_handle *h);
const char *pmdi_last_error(void);
int pmdi_abi_version(void);

/* One Gibbs iteration's sweep for every chain of the handle: replaces
 * src/pmdi.jl:165-171 (reset), :188-207 (known prefix), :209-342 (sweep with
 * calc_logprob, allocation draw, copy-on-write cluster_add!, Phi_upweight!,
 * calc_ESS, draw_partstar, renumbering), :345-350 (particle pick) and :373.
 * The shuffle!(order_obs) of :172 and the hyper-parameter updates of
 * :176-185 stay with the caller.
 *
 * All arrays are per chain, chain-major (chain c at base + c*size):
 *   iter        1-based Gibbs iteration (keys the counter-based RNG)
 *   s_in        n x K Int64 (labels 1..N)                [n*K per chain]
 *   order_obs   n Int64, a permutation of 1..n            [n]
 *   n1          floor(rho*n) >= 1                         (src/pmdi.jl:161)
 *   Pi          N x K Float64 = gamma ./ sum(gamma)       [N*K]  (:179)
 *   Phi         max(1, K(K-1)/2) Float64                  (:61)
 *   feature_flag  sum_k D_k bytes (0/1), dataset-major; NULL = all on (:106-110)
 *   lw_init     initial log-weight: 0.0 on the first iteration, 1.0 after (:99,:372)
 * outputs (any may be NULL):
 *   s_out       n x K Int64 = sstar[p_star,:,:]           (:373)
 *   logweight   P Float64                                  (:99)
 *   p_star      Int64 1-based                              (:350)
 *   stats       pmdi_sweep_stats
 *   trace       (n-n1+1) x (2+2K) Float64 row-major per chain:
 *               [ESS, resampled, max id per k, classes per k] per swept obs
 */
int pmdi_sweep(pmdi_handle *h, int64_t iter, const int64_t *s_in, const int64_t *order_obs,
               int64_t n1, const double *Pi, const double *Phi, const uint8_t *feature_flag,
               double lw_init, int64_t *s_out, double *logweight, int64_t *p_star,
               pmdi_sweep_stats *stats, double *trace);

/* The same sweep on buffers already resident on the handle's device, launched
 * asynchronously on `stream` (a hipStream_t, used verbatim: NULL = the default
 * stream), so that it is ordered with the caller's own work on that stream.
 * Internal encodings (no conversion pass): labels and indices 0-based int32,
 *   s_in/s_out [chain][K][n], order_obs [chain][n], Pi [chain][K][N],
 *   log1p_phi [chain][max(1,K(K-1)/2)] = log(1+Phi), feature_flag as above,
 *   p_star int32 0-based, stats int64[8] per chain.  err: int32 per chain. */
int pmdi_sweep_device(pmdi_handle *h, int64_t iter, const int32_t *s_in, const int32_t *order_obs,
                      int64_t n1, const double *Pi, const double *log1p_phi,
                      const uint8_t *feature_flag, double lw_init, int32_t *s_out,
                      double *logweight, int32_t *p_star, int64_t *stats, int32_t *err,
                      void *stream);

/* Feature selection for the trajectories chosen by the last sweep:
 * replaces src/pmdi.jl:354-370 (calc_logmarginal of every occupied cluster
 * rebuilt from all n rows, plus the null marginal of :120-128 computed in
 * pmdi_create).  s_traj: n x K Int64 per chain (normally s_out).  Outputs
 * per chain: feature_flag sum_k D_k bytes, feature_prob sum_k D_k Float64. */
int pmdi_feature_select(pmdi_handle *h, int64_t iter, const int64_t *s_traj,
                        uint8_t *feature_flag, double *feature_prob);

/* Debug export of the SMC state after the last sweep, in the shapes returned
 * by __pmdi() (src/__pmdi.jl:342) so that the invariants of
 * test/runtests.jl:147-162 can be run on the device path.  Per chain:
 *   particle  N x P x K Int64 (cluster ids)      counts   pool_cap x K Int64
 *   cluster_n pool_cap x K Int64 (cl.n per id)   max_id   K Int64
 * (particle is expanded from the device's column table: distinct columns + a column index per particle.) */
int pmdi_export_state(pmdi_handle *h, int32_t chain, int64_t *particle, int64_t *counts,
                      int64_t *cluster_n, int64_t *max_id);

/* ---- cluster plugin protocol on the device (unit-level parity) ----------
 * A batch of B stand-alone clusters of dataset k.  cluster_add!: rows[b] (1-based
 * row of dataFiles[k]) is added to cluster b; calc_logprob: log posterior
 * predictive of row obs_rows[b] under cluster b; calc_logmarginal: D_k values
 * per cluster.  Replaces, per type, gaussian_cluster.jl:37-83,
 * categorical_cluster.jl:29-66, negbinom_cluster.jl:22-60. */
typedef struct pmdi_cluster_batch pmdi_cluster_batch;
int pmdi_clusters_new(pmdi_handle *h, int32_t k, int32_t B, pmdi_cluster_batch **out);
int pmdi_clusters_free(pmdi_cluster_batch *cb);
int pmdi_cluster_add(pmdi_cluster_batch *cb, const int64_t *rows, const uint8_t *feature_flag);
int pmdi_calc_logprob(pmdi_cluster_batch *cb, const int64_t *obs_rows, const uint8_t *feature_flag,
                      double *out);
int pmdi_calc_logmarginal(pmdi_cluster_batch *cb, double *out /* B x D_k, row per cluster */);
/* stats per cluster: Gaussian n, mu[D], Sigma[D], lambda[D], beta[D];
 * Categorical n, counts[L x D col-major]; NegBinom n, Sigma[D]; returns the
 * number of doubles per cluster in *stride */
int pmdi_cluster_stats(pmdi_cluster_batch *cb, double *out, int64_t *stride);

/* Label occupancy of device-resident allocations: counts[chain][k][label] = #{i : s[chain][k][i] == label}
 * (0-based int32 labels, layout of pmdi_sweep_device's s_out).  This is countn(s[:, k], n) of
 * update_gamma! (src/update_hypers.jl:72, src/misc.jl countn) for every label at once, so that
 * only n_chains*K*N integers cross PCIe per iteration.  Asynchronous on `stream`. */
int pmdi_label_counts_device(pmdi_handle *h, const int32_t *s, int32_t *counts, void *stream);

/* the co-clustering counts behind generate_psm (src/output_analysis/consensus_map.jl:50-56,
 * psm[k][i, j] = sum(output[:, i] .== output[:, j]) / n_iter) for the block of rows [row_lo, row_hi):
 *   counts[k][i - row_lo][j] = #{t : samples[t][k][i] == samples[t][k][j]},  full rows (the reference fills
 * i > j only; the caller masks and divides by S).  samples: device-resident uint8 [S][K][n] (the pooled,
 * all-gathered allocation samples of all chains); counts: device int32 [K][row_hi-row_lo][n].
 * n_labels: every label is < n_labels (the model's N; 0 = unknown).  With 1 <= n_labels <= 64 the counts are
 * computed as a one-hot int8 GEMM on the matrix cores, otherwise by byte compares; both are exact, but a label
 * >= a non-zero n_labels is a caller error (its matches are not counted).
 * Stateless (no handle): `device` is the HIP device ordinal.  Asynchronous on `stream`. */
int pmdi_psm_counts_device(int32_t device, const uint8_t *samples, int64_t S, int32_t K, int64_t n,
                           int64_t row_lo, int64_t row_hi, int32_t n_labels, int32_t *counts, void *stream);

/* ---- device-resident Gibbs chains -----------------------------------------
 * Everything pmdi() does per iteration AROUND the sweep, for every chain of the handle, without leaving the
 * device: shuffle!(order_obs) (src/pmdi.jl:172), update_M!, update_gamma!, Pi, update_Phi!, update_Z, update_v
 * (src/pmdi.jl:176-185, src/update_hypers.jl) evaluated WITHOUT the N^K tables of src/pmdi.jl:69-92, and
 * align_labels! (src/pmdi.jl:375, src/misc.jl:61-96) through N x N contingency tables.  The stale Gamma_c of
 * the reference (built once from the initial gamma, src/pmdi.jl:75-79) is kept.  Host-side draws of the
 * reference (Julia's global RNG) become counter-based Philox variates keyed on (seed + chain, iteration, site).
 * pmdi_gibbs_create replaces src/pmdi.jl:59-66, 95-96, 106-110, 160-161. */
typedef struct pmdi_gibbs pmdi_gibbs;
int pmdi_gibbs_create(pmdi_handle *h, double rho, int32_t feature_select, pmdi_gibbs **out);
int pmdi_gibbs_destroy(pmdi_gibbs *g);      /* before pmdi_destroy of its handle */

/* n_iter iterations of src/pmdi.jl:164-384 for every chain, asynchronously on `stream` (hipStream_t, verbatim).
 * samples: NULL, or device memory for n_iter x n_chains x K x n bytes: the allocations after each iteration
 * (0-based labels), i.e. the rows generate_psm reads back from the CSV (consensus_map.jl:31-46). */
int pmdi_gibbs_iterate(pmdi_gibbs *g, int64_t n_iter, uint8_t *samples, void *stream);

/* The current allocations of every chain as bytes (0-based labels), n_chains x K x n, into device memory `out`:
 * one retained sample in the layout pmdi_psm_counts_device reads.  Asynchronous on `stream`. */
int pmdi_gibbs_pack_samples(pmdi_gibbs *g, uint8_t *out, void *stream);

/* The pieces of one iteration as separate launches (tests; a host that wants to interleave its own work).
 * Order inside pmdi(): BEGIN (iteration counter += 1), HYPERS (:172-185), SWEEP (:165-171,188-350,373),
 * FEATSEL (:354-370, only with feature selection), ALIGN (:375). */
enum { PMDI_STEP_BEGIN = 0, PMDI_STEP_HYPERS = 1, PMDI_STEP_SWEEP = 2, PMDI_STEP_FEATSEL = 3, PMDI_STEP_ALIGN = 4 };
int pmdi_gibbs_step(pmdi_gibbs *g, int32_t what, void *stream);
int64_t pmdi_gibbs_iterations(const pmdi_gibbs *g);

/* Host copies of one chain's state in the reference's shapes (any pointer may be NULL): M[K], gamma and gamma0
 * N x K column-major (gamma0 = exp.(Gamma_c) rows, the initial gamma), Phi[max(1,K(K-1)/2)], vZ = (v, Z),
 * s n x K column-major Int64 labels 1..N, order_obs n Int64 1-based, feature_flag sum_k D_k bytes. */
int pmdi_gibbs_get(pmdi_gibbs *g, int32_t chain, double *M, double *gamma, double *gamma0, double *Phi, double *vZ,
                   int64_t *s, int64_t *order_obs, uint8_t *feature_flag);
int pmdi_gibbs_set(pmdi_gibbs *g, int32_t chain, const double *M, const double *gamma, const double *gamma0,
                   const double *Phi, const double *vZ, const int64_t *s, const int64_t *order_obs,
                   const uint8_t *feature_flag);
/* Counters and outputs of the last sweep for all chains (synchronises): stats n_chains x 8 Int64 (layout of
 * pmdi_sweep_stats), err per chain, p_star 1-based, logweight n_chains x P.  Fails if a chain reported an error. */
int pmdi_gibbs_results(pmdi_gibbs *g, int64_t *stats, int32_t *err, int64_t *p_star, double *logweight);

/* Device pointers of the resident state (zero-copy consumers; layouts of pmdi_sweep_device). */
typedef struct {
    double *M, *gamma, *gamma0, *Phi, *vZ, *Pi, *log1p_phi, *feature_prob, *logweight;
    int32_t *s, *order_obs, *p_star, *err;
    int64_t *stats;
    uint8_t *feature_flag;
    int64_t n1;
} pmdi_gibbs_view;
int pmdi_gibbs_device_view(pmdi_gibbs *g, pmdi_gibbs_view *v);

/* ---- pmdi()'s output files, byte-compatible with Julia's writedlm(io, row', ',') --------
 * pmdi_csv_open writes the header of src/pmdi.jl:147-156 (MassParameter_k..., phi_a_b... [phi_1_1 when K = 1],
 * ll, <name>_n<i>...; data_names NULL = "K1".."KK", :46-48); pmdi_csv_write_row one row [M; Phi; ll; s[1:n*K]]'
 * (:158, :379; every field printed as Julia prints a Float64: shortest round-trip digits, "3.0", "1.0e-5");
 * pmdi_csv_write_gibbs the same from a device-resident chain.  pmdi_csv_open_features / pmdi_csv_write_flags:
 * the feature-selection file (:111-116, :380-382): <name>_d<d> header, rows of true/false.  Host-side code. */
typedef struct pmdi_csv pmdi_csv;
int pmdi_csv_open(const char *path, int32_t K, int64_t n, const char *const *data_names, pmdi_csv **out);
int pmdi_csv_write_row(pmdi_csv *w, const double *M, const double *Phi, double ll, const int64_t *s);
int pmdi_csv_write_gibbs(pmdi_csv *w, pmdi_gibbs *g, int32_t chain, double ll);
int pmdi_csv_open_features(const char *path, int32_t K, const int32_t *D, const char *const *data_names, pmdi_csv **out);
int pmdi_csv_write_flags(pmdi_csv *w, const uint8_t *flags);
int pmdi_csv_close(pmdi_csv *w);
/* Reader side of the output file: what generate_psm, src/output_analysis/consensus_map.jl:32-47, takes
 * from it.  K = header names containing "MassParameter" (:34-36); the data rows after `burnin`, every `thin`-th of them
 * (:33,:38); the allocation columns from K + binomial(K, 2) + (K == 1) + 2 on (:38); n_obs = their number / K, an error if that
 * is not an integer (:40-41); names = the K distinct prefixes before the first '_' of those columns (:47), '\n'-separated.
 * labels (may be NULL: sizes only): bytes [row][k][i], the layout pmdi_psm_counts_device takes, labels_cap bytes available.
 * Host-only. */
int pmdi_csv_read_allocations(const char *path, int64_t burnin, int64_t thin, int32_t *K_out, int64_t *n_obs_out,
                              int64_t *n_iter_out, uint8_t *labels, int64_t labels_cap, char *names, int32_t names_cap);

/* Base.show(::Float64) of one number into out (NUL-terminated); returns its length or a negative error */
int pmdi_format_float64(double x, char *out, int32_t cap);

/* ---- the one exchange step of the multi-GPU path ------------------------------------------
 * Chains are independent (no data-path collective).  After sampling, the retained allocation samples of every
 * rank (uint8 labels, layout of pmdi_gibbs_iterate's `samples`) are all-gathered with RCCL over xGMI so that every
 * GPU can build its row block of the posterior-similarity matrix (pmdi_psm_counts_device; consumer generate_psm,
 * src/output_analysis/consensus_map.jl:31-65).  Two ways to form the communicator:
 *   one process per GPU:  rank 0 calls pmdi_comm_unique_id, the host distributes the 128 bytes (MPI, a file,
 *                         torch.distributed ...), every rank calls pmdi_comm_init_rank;
 *   one process, G GPUs:  pmdi_comm_init_all fills G communicators (devices NULL = 0..G-1).
 * pmdi_allgather_samples: entry i of comms/send/recv/streams belongs to local communicator i (n_local = 1 in the
 * one-process-per-GPU case); recv[i] receives n_ranks x bytes_per_rank bytes in rank order.  Asynchronous on the
 * given streams (hipStream_t, NULL entries / NULL array = default stream).  RCCL is loaded at first use. */
#define PMDI_COMM_ID_BYTES 128
typedef struct pmdi_comm pmdi_comm;
int pmdi_comm_unique_id(uint8_t id[PMDI_COMM_ID_BYTES]);
int pmdi_comm_init_rank(int32_t device, int32_t n_ranks, int32_t rank, const uint8_t id[PMDI_COMM_ID_BYTES], pmdi_comm **out);
int pmdi_comm_init_all(int32_t n_devices, const int32_t *devices, pmdi_comm **out);
int pmdi_comm_destroy(pmdi_comm *c);
int pmdi_comm_rank(const pmdi_comm *c);
int pmdi_comm_size(const pmdi_comm *c);
int pmdi_allgather_samples(pmdi_comm *const *comms, int32_t n_local, const uint8_t *const *send, uint8_t *const *recv,
                           int64_t bytes_per_rank, void *const *streams);

/* Debug: per-phase shader-clock totals of the last sweep (lane 0 of the chain's workgroup);
 * only when the environment variable PMDI_PHASE_TIMERS was set at pmdi_create. */
int pmdi_phase_timers(pmdi_handle *h, int32_t chain, int64_t *out16);

/* sizes a caller needs to allocate outputs */
/* Debug: shader cycles each chain's last sweep took (n_chains values); the library uses them to
 * launch the heaviest chains first. */
int pmdi_chain_costs(pmdi_handle *h, int64_t *out);

/* Work counters of the last sweep, n_chains x K x 8 Int64 per (chain, dataset): [0] clusters whose log-predictive
 * was evaluated (the kernel evaluates only clusters a particle-class leader can reach, src/pmdi.jl:232; the
 * reference's count of :218-220 is n_operations), [1] distinct clusters updated (cluster_add!, :300), [2] of which
 * cloned (:297), [3] cluster ids moved by the renumbering of resampling events (:336), [4] resampling events that
 * moved any, [5] distinct columns of particle[:, :, k] met by the resampling events, summed (the device stores the table by
 * distinct column: a settled chain holds about ten for 1 024 particles), [6] columns created by copy-on-write splits (:301-308).
 * bench.py builds its de-duplication-aware algorithmic byte count from these. */
int pmdi_work_counters(pmdi_handle *h, int64_t *out);

/* 1 when the handle's light chains (few live clusters per step: what a chain looks like after its first iterations) are swept by
 * the settled-chain kernel (csrc/pmdi_sweep2.hip: any mix of Gaussian / Categorical / NegBinom datasets, K <= 4, N <= 64, D <= 64,
 * P in {256, 512, 1024, 2048}, default quirk modes, one workgroup per chain; PMDI_SETTLED=0 switches it off), else 0.
 * given_back4 (optional, 4 Int64): chains that kernel has handed back to the general kernel so far because a step outgrew its
 * tables -- [0] unused (always 0: any number of reachable clusters is evaluated in place); [1] steps with more than 32 particle
 * classes in a dataset (a subset of [2]); [2] steps with more than 16 particle classes in a dataset, or cluster ids beyond 16 bits;
 * [3] in total.  A handed-back chain is swept by the general kernel inside the same call: allocations, picked particle and
 * counters never depend on which kernel ran (the traced ESS agrees to ~1e-13: tree-ordered sums). */
int pmdi_settled_kernel(pmdi_handle *h, int64_t *given_back4);

/* out[n_chains]: which kernel finished each chain's LAST sweep -- 0 the general kernel (csrc/pmdi_sweep.hip), 1 the settled-chain
 * kernel, 2 the general kernel after the settled-chain kernel had handed the chain back in that sweep.  Diagnostics: the parity
 * tests and bench.py's parity_check use it to say which kernel the compared chains ran on. */
int pmdi_chain_swept_by(pmdi_handle *h, int32_t *out);

int pmdi_sum_D(const pmdi_handle *h);
int pmdi_block_threads(const pmdi_handle *h);   /* threads per chain workgroup */
int pmdi_is_split(const pmdi_handle *h);        /* 1: K cooperating workgroups per chain (one per dataset) */
int64_t pmdi_shader_clock_hz(const pmdi_handle *h); /* the clock pmdi_chain_costs counts in (hipDeviceAttributeClockRate) */
int64_t pmdi_lds_bytes(const pmdi_handle *h);    /* LDS bytes per chain workgroup */
int64_t pmdi_pool_cap(const pmdi_handle *h);
int pmdi_categorical_L(const pmdi_handle *h, int32_t k);

#ifdef __cplusplus
}
#endif
#endif
