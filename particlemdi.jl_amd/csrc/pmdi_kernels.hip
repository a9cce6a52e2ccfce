// pmdi_kernels.hip -- unit-level kernels of the cluster plugin protocol (calc_logprob /
// cluster_add! / calc_logmarginal on stand-alone clusters) and the feature-selection pass.
// The sweep itself is in pmdi_sweep.hip.  Compile with -ffp-contract=off.
#include "pmdi_device.h"

using namespace pmdi_dev;

namespace {

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
            double2 ml = ld2(s.ml, (size_t)id * D + q), sb = ld2(s.sb, (size_t)id * D + q);
            gauss_add(glob(d.xf)[(size_t)row * D + q], nnew, ml, sb);
            st2(s.ml, (size_t)id * D + q, ml); st2(s.sb, (size_t)id * D + q, sb);
        } else if (d.kind == K_CATEGORICAL) {
            s.cnt[((size_t)id * D + q) * d.L + (glob(d.xi)[(size_t)row * D + q] - 1)] += 1;
        } else {
            s.nbs[(size_t)id * D + q] += glob(d.xi)[(size_t)row * D + q];
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
            out = (double)nflag * glob(d.gtab)[cn];
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                double ta, tb;
                gauss_terms(glob(d.xf)[(size_t)row * D + q], (double)cn, ld2(s.ml, (size_t)id * D + q), ta, tb);
                out += ta; out -= tb;
            }
        } else if (d.kind == K_CATEGORICAL) {
            double acc = 0.0;
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                acc += glob(d.lhtab)[glob(d.maxcol)[q] + 2 * cn];
            }
            out = -acc;
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                const int c = s.cnt[((size_t)id * D + q) * d.L + (glob(d.xi)[(size_t)row * D + q] - 1)];
                out += (cn == 0) ? glob(d.lhtab)[1] : glob(d.lhtab)[2 * c + 1];
            }
        } else {
            out = 0.0;
            for (int q = 0; q < D; ++q) {
                if (a.flags && !a.flags[q]) continue;
                out += negbin_term(glob(d.lgtab), cn, glob(d.xi)[(size_t)row * D + q], s.nbs[(size_t)id * D + q]);
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
        return (-a_n) * log(beta) + glob(d.lmtab)[cn];
    } else if (d.kind == K_CATEGORICAL) {
        const int mc = glob(d.maxcol)[q];                       // 2*nlevels_q
        double v = 0.0;
        v += glob(d.lghtab)[2 * mc] - glob(d.lghtab)[2 * (mc + cn)];
        for (int r = 0; r < mc; ++r) v += glob(d.lghtab)[2 * cnt_q[r] + 1];
        return v;
    } else {
        return glob(d.lgtab)[S + 1] - glob(d.lgtab)[S + (cn + 1 + 1)] + glob(d.lgtab)[1 + cn];
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
        const double beta = d.kind == K_GAUSSIAN ? ld2(s.sb, (size_t)id * D + q).y : 0.0;
        const int *cq = d.kind == K_CATEGORICAL ? gen(s.cnt + ((size_t)id * D + q) * d.L) : nullptr;
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
    int *lcnt = (int *)smem;                  // [lanes][ltile] categorical level counts
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
                gauss_add(glob(d.xf)[(size_t)i * D + q], c, ml, sb);
            }
            val = logmarginal_one(d, cn, sb.y, nullptr, 0, q);
        } else if (d.kind == K_CATEGORICAL) {
            // calc_logmarginal(::CategoricalCluster) (categorical_cluster.jl:53-66) with the level counts held in LDS a tile of
            // levels at a time (any number of levels fits; the members are re-read once per tile), terms added in level order
            const int Lt = a.ltile;
            int *mine = lcnt + (size_t)tid * Lt;
            const int mc = glob(d.maxcol)[q];                       // 2 * nlevels_q
            val = 0.0;
            val += glob(d.lghtab)[2 * mc] - glob(d.lghtab)[2 * (mc + cn)];
            for (int l0 = 0; l0 < mc; l0 += Lt) {
                for (int l = 0; l < Lt; ++l) mine[l] = 0;
                for (long long i = first; i < n; ++i) {
                    if (traj[i] != u) continue;
                    const int x = glob(d.xi)[(size_t)i * D + q] - 1 - l0;
                    if (x >= 0 && x < Lt) mine[x] += 1;
                }
                const int hi = min(Lt, mc - l0);
                for (int r = 0; r < hi; ++r) val += glob(d.lghtab)[2 * mine[r] + 1];
            }
        } else {
            long long S = 0;
            for (long long i = first; i < n; ++i) {
                if (traj[i] != u) continue;
                S += glob(d.xi)[(size_t)i * D + q];
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

// countn (src/misc.jl) for all labels of one (chain, dataset) row of allocations: LDS histogram
__global__ void __launch_bounds__(256) label_count_kernel(const int *__restrict__ s, int *__restrict__ counts, long long n, int N)
{
    __shared__ int hist[256];  // N <= 255 (pmdi_create)
    const int row = blockIdx.x;
    for (int l = threadIdx.x; l < N; l += blockDim.x) hist[l] = 0;
    __syncthreads();
    const int *sr = s + (size_t)row * n;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = sr[i];
        if (v >= 0 && v < N) atomicAdd(&hist[v], 1);
    }
    __syncthreads();
    for (int l = threadIdx.x; l < N; l += blockDim.x) counts[(size_t)row * N + l] = hist[l];
}

// generate_psm's co-clustering counts (consensus_map.jl:50-56).  One workgroup = a 64 x 64 tile of
// (row i, column j) pairs of one dataset; 256 lanes, 4 x 4 pairs each.  The labels of 64 samples for the
// tile's 64 rows and 64 columns are staged in LDS (sample-major, so a lane reads its 4 row labels and 4
// column labels as one dword each); integer compares and adds only -- bit-exact by construction.
#define PSM_TT 64
__global__ void __launch_bounds__(256) psm_count_kernel(const unsigned char *__restrict__ samples, long long S, int K, long long n,
                                                        long long row_lo, long long row_hi, int *__restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) unsigned char As[PSM_TT][64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[PSM_TT][64];
    const int k = blockIdx.z;
    const long long i0 = row_lo + (long long)blockIdx.y * 64, j0 = (long long)blockIdx.x * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    int acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0;
    const int lc = tid & 63, lt = tid >> 6;                 // staging: column of the tile, sample row mod 4
    for (long long t0 = 0; t0 < S; t0 += PSM_TT) {
#pragma unroll 4
        for (int tt = lt; tt < PSM_TT; tt += 4) {
            const long long t = t0 + tt;
            unsigned char av = 255, bv = 254;                // out of range: never equal to anything
            if (t < S) {
                const unsigned char *row = samples + ((size_t)t * K + k) * n;
                if (i0 + lc < row_hi) av = row[i0 + lc];
                if (j0 + lc < n) bv = row[j0 + lc];
            }
            As[tt][lc] = av; Bs[tt][lc] = bv;
        }
        __syncthreads();
#pragma unroll 8
        for (int tt = 0; tt < PSM_TT; ++tt) {
            const unsigned a4 = *(const unsigned *)&As[tt][ty * 4];
            const unsigned b4 = *(const unsigned *)&Bs[tt][tx * 4];
            unsigned a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (a4 >> (8 * r)) & 0xffu; b[r] = (b4 >> (8 * r)) & 0xffu; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += (a[r] == b[c]) ? 1 : 0;
        }
        __syncthreads();
    }
    const long long rows = row_hi - row_lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long i = i0 + ty * 4 + r;
        if (i >= row_hi) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long j = j0 + tx * 4 + c;
            if (j < n) counts[((size_t)k * rows + (i - row_lo)) * n + j] = acc[r][c];
        }
    }
}

// The same counts on the matrix cores, for labels known to be < 32 * NKB: with one-hot rows
// A[i][(t, l)] = [samples[t][i] == l] the counts are A * A^T, an int8 GEMM whose K dimension is
// (sample, label).  One v_mfma_i32_32x32x32_i8 covers one sample x 32 labels for a 32 x 32 tile of pairs.
// Workgroup = 4 waves = a 128 x 128 tile; each wave a 64 x 64 quadrant (4 accumulator tiles).  The one-hot
// fragments are built in registers from the staged label bytes: lane (r = l & 31, h = l >> 5) holds 16 of
// the 32 k-values of row r; which 16 does not matter as long as A and B use the same rule, because the
// sum over k is permutation-invariant and the hardware pairs A's and B's k by the same (h, byte) position.
// C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (cdna_hip_programming.md).
typedef int psm_v4i __attribute__((ext_vector_type(4)));
typedef int psm_v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ psm_v4i psm_onehot(int label, int kb, int h)
{
    const unsigned x = (unsigned)(label - 32 * kb - 16 * h);      // byte position among this lane's 16 k-values
    const unsigned bit = (x < 16u) ? (1u << ((x & 3u) * 8u)) : 0u;
    const unsigned dw = x >> 2;
    psm_v4i f;
    f.x = (dw == 0u) ? (int)bit : 0; f.y = (dw == 1u) ? (int)bit : 0; f.z = (dw == 2u) ? (int)bit : 0; f.w = (dw == 3u) ? (int)bit : 0;
    return f;
}

#define PSM_MT 32       // samples staged per round
template <int NKB>
__global__ void __launch_bounds__(256) psm_count_mfma_kernel(const unsigned char *__restrict__ samples, long long S, int K, long long n,
                                                             long long row_lo, long long row_hi, int *__restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) unsigned char As[PSM_MT][128];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[PSM_MT][128];
    const int k = blockIdx.z;
    const long long i0 = row_lo + (long long)blockIdx.y * 128, j0 = (long long)blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;                   // this wave's 64 x 64 quadrant
    const int r = lane & 31, h = lane >> 5;
    psm_v16i acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
    const int lc = tid & 127, lt = tid >> 7;                   // staging: column of the tile, sample row mod 2
    for (long long t0 = 0; t0 < S; t0 += PSM_MT) {
#pragma unroll 4
        for (int tt = lt; tt < PSM_MT; tt += 2) {
            const long long t = t0 + tt;
            unsigned char av = 255, bv = 254;                  // out of range: outside every 32-label block used
            if (t < S) {
                const unsigned char *row = samples + ((size_t)t * K + k) * n;
                if (i0 + lc < row_hi) av = row[i0 + lc];
                if (j0 + lc < n) bv = row[j0 + lc];
            }
            As[tt][lc] = av; Bs[tt][lc] = bv;
        }
        __syncthreads();
#pragma unroll 2
        for (int tt = 0; tt < PSM_MT; ++tt) {
            const int a0 = As[tt][wy * 64 + r], a1 = As[tt][wy * 64 + 32 + r];
            const int b0 = Bs[tt][wx * 64 + r], b1 = Bs[tt][wx * 64 + 32 + r];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const psm_v4i fa0 = psm_onehot(a0, kb, h), fa1 = psm_onehot(a1, kb, h);
                const psm_v4i fb0 = psm_onehot(b0, kb, h), fb1 = psm_onehot(b1, kb, h);
                acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa0, fb0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa0, fb1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa1, fb0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa1, fb1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const long long rows = row_hi - row_lo;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long i = i0 + wy * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const long long j = j0 + wx * 64 + b * 32 + r;
                if (i < row_hi && j < n) counts[((size_t)k * rows + (i - row_lo)) * n + j] = acc[a][b][e];
            }
}

hipError_t pmdi_launch_psm_counts_mfma(const unsigned char *samples, long long S, int K, long long n, long long row_lo, long long row_hi,
                                       int n_labels, int *counts, hipStream_t stream)
{
    const long long rows = row_hi - row_lo;
    if (rows <= 0 || n <= 0 || K <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + 127) / 128), (unsigned)((rows + 127) / 128), (unsigned)K);
    if (n_labels <= 32) hipLaunchKernelGGL(psm_count_mfma_kernel<1>, grid, dim3(256), 0, stream, samples, S, K, n, row_lo, row_hi, counts);
    else if (n_labels <= 64) hipLaunchKernelGGL(psm_count_mfma_kernel<2>, grid, dim3(256), 0, stream, samples, S, K, n, row_lo, row_hi, counts);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t pmdi_launch_psm_counts(const unsigned char *samples, long long S, int K, long long n, long long row_lo, long long row_hi,
                                  int *counts, hipStream_t stream)
{
    const long long rows = row_hi - row_lo;
    if (rows <= 0 || n <= 0 || K <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)((rows + 63) / 64), (unsigned)K);
    hipLaunchKernelGGL(psm_count_kernel, grid, dim3(256), 0, stream, samples, S, K, n, row_lo, row_hi, counts);
    return hipGetLastError();
}

hipError_t pmdi_launch_label_counts(const int *s, int *counts, int n_rows, long long n, int N, hipStream_t stream)
{
    hipLaunchKernelGGL(label_count_kernel, dim3(n_rows), dim3(256), 0, stream, s, counts, n, N);
    return hipGetLastError();
}

hipError_t pmdi_launch_featsel(const FeatSelArgs &a, int n_chains, hipStream_t stream)
{
    int Lmax = 1, lanes = 1;
    for (int k = 0; k < a.K; ++k)
        if (a.ds[k].kind == K_CATEGORICAL) {
            if (a.ds[k].L > Lmax) Lmax = a.ds[k].L;
            if (a.ds[k].D > lanes) lanes = a.ds[k].D;
        }
    if (lanes > 256) lanes = 256;
    FeatSelArgs b = a;
    b.ltile = Lmax < 12288 / lanes ? Lmax : 12288 / lanes;       // <= 48 KiB of level counts per workgroup
    const size_t lds = (size_t)lanes * b.ltile * 4;              // lanes beyond min(D, 256) never touch their slice
    hipLaunchKernelGGL(featsel_label_kernel, dim3(n_chains * a.K * a.N), dim3(256), lds, stream, b);
    hipLaunchKernelGGL(featsel_combine_kernel, dim3((a.sumD + 255) / 256, n_chains), dim3(256), 0, stream, a);
    return hipGetLastError();
}
