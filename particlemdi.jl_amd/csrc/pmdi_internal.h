// pmdi_internal.h -- structures shared by the host C-ABI (pmdi_api.cpp) and
// the device kernels (pmdi_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PMDI_KMAX_I 8
#define PMDI_INF_I 0x7fffffff

// Compile-time capacities of the per-chain LDS tables (a step whose working set exceeds them
// runs on the global-memory fallback).  Constants, so that the addresses of these tables fold
// into instruction immediates instead of occupying registers.
#define PMDI_ITEM_CAP 256   // (class, label) items of a fast-path step
#define PMDI_ITEM_CAP_BIGN 384   // ... when N > 32 (N = 50: seven classes instead of five); must stay below PMDI_HT_SIZE
#define PMDI_HT_SIZE 512    // entries per hash table (>= 2 * PMDI_ITEM_CAP, power of two)
#define PMDI_CLS_LDS 128    // class-list slots per dataset kept in LDS (64 measured 11 % slower at HL: chains with 64..128 classes set the sweep time)
#define PMDI_DL_LDS 128     // distinct-chosen-cluster entries kept in LDS (fallback path)

enum { K_GAUSSIAN = 0, K_CATEGORICAL = 1, K_NEGBINOM = 2 };
enum { SITE_DRAW = 0, SITE_RESAMPLE_U = 1, SITE_RESAMPLE_SLOT = 2, SITE_PSTAR = 3, SITE_FEATSEL = 4,
       // the per-iteration work around the sweep (pmdi_hypers.hip; same numbering as oracle/pmdi_oracle_hypers.c)
       SITE_SHUFFLE = 5, SITE_M_NORMAL = 6, SITE_M_ACCEPT = 7, SITE_GAMMA = 8, SITE_PHI_ALPHA = 9, SITE_PHI_GAMMA = 10,
       SITE_V = 11, SITE_ALIGN = 12, SITE_INIT_GAMMA = 13, SITE_INIT_PHI = 14, SITE_INIT_S = 15, SITE_INIT_FLAGS = 16 };
enum { ST_NOPS = 0, ST_NRESAMPLE = 1, ST_NCLONES = 2, ST_MAXID = 3, ST_SUMCLASSES = 4 };
// work counters per (chain, dataset), 8 slots each: clusters whose log-predictive was evaluated, distinct clusters
// updated (deepcopy + cluster_add!), of which cloned, cluster ids moved down by the renumbering of a resampling event,
// resampling events that moved any id, distinct columns of the particle -> cluster table met by the resampling events (summed),
// columns created by copy-on-write splits
enum { WK_EVAL = 0, WK_UPD = 1, WK_CLONE = 2, WK_MOVED = 3, WK_MOVE_EVENTS = 4, WK_COLS = 5, WK_SPLITS = 6 };

// One dataset as the kernels see it.  Data are row-major on the device so
// that an observation row is one contiguous, coalesced read.
struct DsetDev {
    int kind, D, L, flag_off;
    const double *xf;       // [n][D]   Gaussian
    const int *xi;          // [n][D]   Categorical (levels 1..L) / NegBinom (counts)
    // host-built tables (so that integer-data predictives are bit-identical
    // to a CPU evaluation with the same libm):
    const double *gtab;     // Gaussian: G[m] = log(1/sqrt(pi)) + lgamma(m/2+1) - lgamma(m/2+1/2), m = 0..n
    const double *lmtab;    // Gaussian logmarginal constant per cluster size m = 0..n
    const double *lhtab;    // Categorical: LH[j] = log(j/2),    j = 0..2n+L+2
    const double *lghtab;   // Categorical: LGH[j] = lgamma(j/2), j = 0..2(n+L)+2
    const int *maxcol;      // Categorical: column maxima (nlevels = maxcol/2)
    const double *lgtab;    // NegBinom: LG[m] = lgamma(m), m = 0..lgtab_len-1
    long long lgtab_len;
    // per-(chain, dataset) state arena: chain c lives at arena + c*stride
    char *arena;
    size_t stride;
    size_t o_particle[2];   // int  [P][N]  the DISTINCT columns of particle[:, p, k] (label -> cluster id, 1-based): column c at
                            //              [c*N, c*N + N); columns 0..ncol-1 are live; double buffered (a resampling event compacts)
    size_t o_col;           // int  [P]     column of each particle
    size_t o_cgrp;          // u64  [P][N]  scratch of the copy-on-write split, keyed (column, label): (step << 32) | group's column
    size_t o_pid;           // int  [P]     class of each particle (particle_id)
    size_t o_sid;           // int  [P]     sstar_id
    size_t o_kv;            // int  [P]     per-particle scratch (class-key value), used when not in LDS
    size_t o_newid;         // int  [P][N]  new_id[(class-1)*N + label]
    size_t o_counts;        // int  [cap+1]
    size_t o_ncop;          // int  [cap+1] scratch, 0 between steps
    size_t o_firstc;        // int  [cap+1] scratch, INF between steps
    size_t o_lp;            // double [cap+1] logprob table
    size_t o_cn;            // int  [cap+1] cl.n
    size_t o_ml;            // double2 [cap+1][D] (mu, lambda): stand-alone cluster batches only
    size_t o_sb;            // double2 [cap+1][D] (Sigma, beta)
    size_t o_cnt;           // int  [cap+1][D][L]
    size_t o_nbs;           // long long [cap+1][D]
    size_t o_sstar;         // uchar [n][P]  allocation history (0-based labels)
    size_t o_clslead;       // int  [P]   class slot -> leader particle
    size_t o_clsval;        // int  [P]   class slot -> class value
    size_t o_cdf;           // double [P][N+2]  per class slot: CDF, log-increment, one-hot label or -1
    size_t o_dl;            // int  [3][P]  distinct chosen ids this step: src, dst, new n
    size_t o_s2x;           // settled-chain kernel: log-predictives (double [2048]) and ids (int [2048]) of the reachable clusters beyond its LDS list
};

// LDS layout of the settled-chain kernel (pmdi_sweep2_body.h: make_layout fills it on the host, the kernel reads it from the
// argument block): byte offsets into the workgroup's LDS
struct S2Layout {
    int red, sc, stat, wk, ph, leaf_i1, leaf_n, leaf_tot, leaf_carry, leaf_prog, xfl;
    int tr, tr_stride, tr_tb, tr_cdf, tr_bytes;       // transient region: per dataset tb rows + CDF rows; aliased by the resampling scratch
    int rs_jtab, rs_raw, rs_anc, rs_hist;             // resampling scratch inside the transient region
    int ds0, ds_stride;                               // dataset blocks
    // offsets inside a dataset block
    int tab, cmask, wmask, cbi, counts, cn, ncop, firstp, tgt, slotmap, ta, tax, lp, slot_id, slot_cn, slot_g, clsval, clslead, leadcol,
        minp, nidv, knew, itemj, clist, klist, kval, krep, bmc, bmf, cbm, kbm, xid, dsc;
    int Dp, cols_l, idcap, total;
    int cls, kcap;          // particle classes per dataset the tables hold (16 .. 32); touched (class, label) keys per step the key lists hold
    int cdfl;               // class slots whose mutation-CDF rows live in LDS (the rest of the cls rows: the chain's arena)
};

struct SweepArgs {
    int K, N, P, cap;
    int Dmax, sumD, npairs, q1, q2, trace_on;
    int terms_cap;          // doubles in the LDS term buffer
    int pid_lds;            // 1: particle class ids [K][P] live in LDS
    int pp_lds;             // 1: per-particle step scratch (sid, kv) lives in LDS
    int col_lds;            // 1: the particles' column indices [K][P] live in LDS
    int two_per_cu;         // 1: register-capped build so that two chains co-reside on a CU
    int item_cap;           // (class, label) items a fast-path step may have: PMDI_ITEM_CAP, or PMDI_ITEM_CAP_BIGN when N > 32
    unsigned iter;
    long long n, n1;
    unsigned long long seed;
    double lw_init;
    DsetDev ds[PMDI_KMAX_I];
    const int *s_in;            // [chain][K][n]
    const int *order;           // [chain][n]
    const double *Pi;           // [chain][K][N]
    const double *logphi;       // [chain][npairs]
    const unsigned char *flags; // [chain][sumD] or null
    int *s_out;                 // [chain][K][n]
    double *lw_out;             // [chain][P]
    int *pstar;                 // [chain]
    long long *stats;           // [chain][8]
    int *err;                   // [chain]
    double *trace;              // [chain][n-n1+1][2+2K]
    // per-chain scratch
    double *uscratch;           // [chain][P]
    int *partstar;              // [chain][P]
    int *kstate;                // [chain][KMAX][2]  final (max id, particle buffer) per dataset
    long long *phase;           // [chain][16] per-phase shader-clock totals of lane 0, or null
    long long *work;            // [chain][KMAX][8] work counters (WK_*), or null
    int *anclog;                // q2_mode 1: [chain][n-n1+1][P] ancestor table of every resampling event
    int *evpos;                 // q2_mode 1: [chain][2][n-n1+1] position of every resampling event; the selected particle's lineage
    long long *cost;            // [chain] shader cycles this chain's sweep took (drives the next launch order)
    const int *chain_order;     // [n_chains] workgroup b sweeps chain chain_order[b] (heaviest first), or null
    const unsigned char *group_flag;  // [n_chains] 1 = heavy chain (many private clusters), 0 = light; or null
    // split mode (K > 1): the K datasets of a chain are swept by K cooperating workgroups that meet once per swept observation
    int ksplit;                 // 1: one dataset per workgroup, grid = chain slots (padded to 8) x K
    int n_slots;                // chain slots of this launch
    int *xcnt;                  // [chain][32] arrival counter of the chain's hand-off (first word; own 128-B line)
    double *xinc;               // [chain][2][K][P] log-weight increment per particle (by parity of the swept observation)
    int *xlab;                  // [chain][2][K][P] chosen label per particle
    unsigned long long *xhdr;   // [chain][2][K][2] per-step header: flags / labels, increment of class slot 0
    unsigned *start_sig;        // every workgroup of this launch adds 1 on entry (signal memory), or null: the other launches of a sweep
                                // wait for the heaviest chains' workgroups to have been placed (a whole CU each) before they start
    int group_sel;              // this launch sweeps the chains whose flag equals group_sel (when group_flag != null)
    int rank_lo, rank_hi;       // ... and whose position in the launch order is in [rank_lo, rank_hi)
    S2Layout s2;                // settled-chain kernel: its LDS layout (cols_l / idcap = columns / cluster ids kept in LDS)
    int *requeue;               // [chain] 1: the settled-chain kernel gave the chain back (it does not fit its tables): sweep it again
    int *handed;                // [chain] number of the sweep in which the chain was last given back (such a chain goes to the general
                                // kernel directly for the next few sweeps: it tends to be given back again), or null
    int sweep_no;               // this sweep's number (per handle)
    long long *requeue_total;   // [4] chains given back so far, by reason (reachable clusters, chosen clusters, classes), and in total
    int requeue_only;           // 1: this launch of the general kernel sweeps exactly those chains
    int slot_base;              // split mode launched in residency-sized batches: first chain slot of this launch
    int err_keep;               // 1: a successful sweep leaves err[chain] as it is (device-resident chains: the first error sticks)
    int *resume;                // [chain][16] hand-over record of the settled-chain kernel: [0] position of the observation whose step did
                                // not fit its tables, [1] mask of the datasets whose step of that observation is done, [2 + k] live columns
                                // of dataset k; null: no continuation (a given-back chain is swept again from the start)
    int resume_mode;            // 1: this launch of the general kernel carries on the given-back chains from their hand-over records
    int *ticket;                // settled-chain launch: [0] a counter zeroed before the launch, [1 + b] the position in the launch order
                                // workgroup b drew from it.  The hardware deals workgroup indices to 32 dispatch queues (8 XCDs x 4 shader
                                // engines, 16 workgroup slots each) statically, so `chain_order[blockIdx.x]` is 32 separate greedy schedules;
                                // with a ticket the next chain of the order goes to whichever slot of the whole GPU frees first.  Null:
                                // position = blockIdx.x
    int *swept_by;              // [chain] which kernel finished the chain's last sweep: 0 general kernel, 1 settled-chain kernel,
                                // 2 general kernel after the settled-chain kernel gave the chain back; or null
};

struct ClusterBatchArgs {
    DsetDev ds;        // arena/offset fields describe the batch's own storage (chain 0)
    int B;
    const int *rows;   // [B] 0-based
    const unsigned char *flags; // [D] or null
    double *out;
};

struct FeatSelArgs {
    int K, N, sumD;
    int ltile;              // categorical levels counted per pass of the per-lane LDS histogram (set by the launcher)
    long long n;
    unsigned iter;
    unsigned long long seed;
    DsetDev ds[PMDI_KMAX_I];
    const int *traj;        // [chain][K][n] 0-based labels
    double *lm;             // [chain][sumD][N]  per-label log marginals
    int *firstpos;          // [chain][K][N]     first position of each label (INF if absent)
    const double *fnull;    // [sumD] -(log marginal of the all-in-one cluster)
    unsigned char *flags_out; // [chain][sumD]
    double *prob_out;         // [chain][sumD]
};

// Device-resident Gibbs state of every chain of a handle (src/pmdi.jl:59-96) and the scratch of the
// hyper-parameter / label-alignment kernels (pmdi_hypers.hip).
struct GibbsArgs {
    int K, N, npairs, n_chains;
    long long n;
    unsigned iter;
    unsigned long long seed;
    double *M;          // [chain][K]
    double *gamma;      // [chain][K][N]   gamma_c
    double *gamma0;     // [chain][K][N]   the initial gamma_c: exp(Gamma_c), never refreshed
    double *Phi;        // [chain][npairs]
    double *vZ;         // [chain][2]      (v, Z)
    int *s;             // [chain][K][n]   allocations, 0-based labels
    int *order;         // [chain][n]      order_obs, 0-based
    double *Pi;         // [chain][K][N]   gamma ./ sum(gamma)  (src/pmdi.jl:179)
    double *logphi;     // [chain][npairs] log(1 + Phi)         (src/misc.jl:53)
    int *ctab;          // [chain][K][K][N][N] contingency tables of align_labels (diagonal blocks unused)
    double *wscr;       // [chain][n + 1]  weights of the alpha* draw in update_Phi!
    const double *lgtab;// [n + 3]         lgamma(m), host-built (same libm as the oracle)
    int ctab_lds;       // 1: the contingency tables fit LDS
    int order_lds;      // 1: order_obs fits LDS during the shuffle
};

size_t pmdi_hypers_lds_bytes(const GibbsArgs &a);
size_t pmdi_align_lds_bytes(const GibbsArgs &a);
hipError_t pmdi_launch_gibbs_init(const GibbsArgs &a, hipStream_t stream);
hipError_t pmdi_launch_hypers(const GibbsArgs &a, int do_shuffle, hipStream_t stream);
hipError_t pmdi_launch_align(const GibbsArgs &a, hipStream_t stream);
hipError_t pmdi_launch_pack_samples(const int *s, unsigned char *out, long long count, hipStream_t stream);

size_t pmdi_sweep_lds_bytes(const SweepArgs &a, int T);
// `staging`: a pinned host slot the argument block is copied through (truly asynchronous), or null (pageable copy: host-synchronous)
hipError_t pmdi_launch_sweep(const SweepArgs &a, SweepArgs *d_args, int n_chains, int T, hipStream_t stream, SweepArgs *staging = nullptr);
// workgroups of the sweep kernel build (T threads, the argument block's LDS layout) that one CU holds at once
hipError_t pmdi_sweep_blocks_per_cu(const SweepArgs &a, int T, int *blocks);
// the settled-chain kernel (pmdi_sweep2.hip)
void pmdi_sweep2_layout(int K, int N, int P, int Dmax, int cols_l, int idcap, int cls, int cdfl, S2Layout *L);
bool pmdi_sweep2_supports(int K, int N, int P, int Dmax, long long cap);
int pmdi_sweep2_threads(int K, int P);
bool pmdi_sweep2_cdf_arena(int K, int P);     // that shape's build takes S2Layout::cdfl < cls
int pmdi_sweep2_max_classes(int K, int P);  // particle classes per dataset the class slots of that shape can name (16 or 32)      // threads of the workgroup that sweeps a chain of P particles (0: no build for it)
hipError_t pmdi_sweep2_blocks_per_cu(const SweepArgs &a, int *blocks);
hipError_t pmdi_launch_sweep2(const SweepArgs &a, SweepArgs *d_args, int n_chains, hipStream_t stream, SweepArgs *staging = nullptr);
hipError_t pmdi_launch_cluster_add(const ClusterBatchArgs &a, hipStream_t stream);
hipError_t pmdi_launch_cluster_logprob(const ClusterBatchArgs &a, hipStream_t stream);
hipError_t pmdi_launch_cluster_logmarginal(const ClusterBatchArgs &a, hipStream_t stream);
hipError_t pmdi_launch_chain_order(const long long *cost, int *order, const long long *stats, unsigned char *group_flag,
                                   long long light_ops_max, int n_chains, hipStream_t stream, const int *handed = nullptr, int sweep_no = 0);
hipError_t pmdi_launch_featsel(const FeatSelArgs &a, int n_chains, hipStream_t stream);
hipError_t pmdi_launch_psm_counts(const unsigned char *samples, long long S, int K, long long n, long long row_lo, long long row_hi,
                                  int *counts, hipStream_t stream);
hipError_t pmdi_launch_psm_counts_mfma(const unsigned char *samples, long long S, int K, long long n, long long row_lo, long long row_hi,
                                       int n_labels, int *counts, hipStream_t stream);
hipError_t pmdi_launch_label_counts(const int *s, int *counts, int n_rows, long long n, int N, hipStream_t stream);
