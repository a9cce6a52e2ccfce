// pmdi_sweep.hip -- the conditional-SMC sweep of one Gibbs iteration as ONE persistent
// kernel for gfx950 (CDNA4).  Compile with -ffp-contract=off: the floating-point expression
// order is the reference's and must not be contracted into FMAs.
//
// Mapping (DESIGN.md has the full account):
//   * one workgroup = one Gibbs chain, resident for the whole sweep (known-prefix build, the
//     n_s x K strictly sequential steps, resampling, particle pick): a kernel boundary per
//     step (>= 1.5 us) would cost more than a step's work;
//   * one particle per lane for everything per particle (allocation draw, weight update,
//     class and copy-on-write bookkeeping);
//   * the reference's de-duplication is kept and tightened: log-predictives are evaluated only
//     for the clusters a particle-class leader can reach (lanes = cluster x feature), mutation
//     CDFs once per class (lanes = class x label inside a wave, shuffles);
//   * the per-step working set lives in LDS: observation row, Pi, log-weights, class ids and
//     class lists, the needed-cluster logprob table, per-class CDFs, a hash table for the
//     chosen-cluster census, scan/reduction scratch.  HBM/L2 holds the bulk state only: the
//     cluster-statistics pool, the label->cluster table and the allocation history.  When a
//     step's working set outgrows the LDS structures (burn-in: hundreds of classes) the step
//     falls back to per-id tables in global memory -- same results, slower.
//
// Reference lines are cited as file:line relative to /root/reference.
#include "pmdi_sweep_body.h"

namespace {

// WPS = minimum waves per SIMD the register allocation must allow (2 co-resident chains per CU at T = 512 need 4); K1: the build
// for single-dataset models; MANY: the build for more than 64 labels (pmdi_sweep_body.h)
template <int T, int WPS, bool K1, bool MANY>
__global__ void __launch_bounds__(T, WPS) pmdi_sweep_kernel(const SweepArgs *__restrict__ ap)
{
    pmdi_sweep_body<T, WPS, K1, MANY, false>(ap);
}

// Launch order for the next sweep: chains sorted by the cycles their last sweep took, heaviest
// first (longest-processing-time order: the few slow chains start at once and the many fast ones
// fill in behind them, instead of a slow chain starting last and leaving the GPU idle).
__global__ void chain_order_kernel(const long long *cost, int *order, const long long *stats, unsigned char *group_flag,
                                   long long light_ops_max, int n_chains, const int *handed, int sweep_no)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    // group of the next sweep: a chain is light when its sweep met few live clusters per step
    // (stats[0] = sum over steps of the live-cluster count, the reference's n_operations)
    // ... and was not given back by the settled-chain kernel in one of the last three sweeps (a chain in a state with many
    // particle classes tends to stay there for a while: it goes to the general kernel directly instead of being swept twice)
    if (group_flag) group_flag[c] = (stats[(size_t)c * 8] > light_ops_max || (handed && sweep_no - handed[c] < 3)) ? 1 : 0;
    const long long mine = cost[c];
    int rank = 0;
    for (int j = 0; j < n_chains; ++j) {
        const long long o = cost[j];
        rank += (o > mine || (o == mine && j < c)) ? 1 : 0;
    }
    order[rank] = c;
}

}  // namespace

hipError_t pmdi_launch_chain_order(const long long *cost, int *order, const long long *stats, unsigned char *group_flag,
                                   long long light_ops_max, int n_chains, hipStream_t stream, const int *handed, int sweep_no)
{
    hipLaunchKernelGGL(chain_order_kernel, dim3((n_chains + 255) / 256), dim3(256), 0, stream, cost, order, stats, group_flag,
                       light_ops_max, n_chains, handed, sweep_no);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
size_t pmdi_sweep_lds_bytes(const SweepArgs &a, int T)
{
    (void)T;
    Carve c;
    carve_lds(a, c);
    return c.total;
}

static const void *sweep_kernel_for(const SweepArgs &a, int T)
{
    const bool two = a.two_per_cu != 0;
    const bool k1 = a.K == 1 || a.ksplit;      // one dataset per workgroup
    const bool many = a.N > 64;
#define PMDI_PICK(T_, W_) (k1 ? (many ? (const void *)pmdi_sweep_kernel<T_, W_, true, true> : (const void *)pmdi_sweep_kernel<T_, W_, true, false>) \
                              : (many ? (const void *)pmdi_sweep_kernel<T_, W_, false, true> : (const void *)pmdi_sweep_kernel<T_, W_, false, false>))
    if (T == 1024) return PMDI_PICK(1024, 4);
    if (T == 512 && two) return PMDI_PICK(512, 4);
    if (T == 512) return PMDI_PICK(512, 2);
    if (T == 128) return PMDI_PICK(128, 2);
    if (T == 256) return PMDI_PICK(256, PMDI_LIGHT_WPS);
#undef PMDI_PICK
    return nullptr;
}

hipError_t pmdi_sweep_blocks_per_cu(const SweepArgs &a, int T, int *blocks)
{
    const void *fn = sweep_kernel_for(a, T);
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = pmdi_sweep_lds_bytes(a, T);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, fn, T, lds);
}

hipError_t pmdi_launch_sweep(const SweepArgs &a_in, SweepArgs *d_args, int n_chains, int T, hipStream_t stream, SweepArgs *staging)
{
    SweepArgs a = a_in;
    if (!a.ksplit || a.n_slots <= 0) { a.n_slots = n_chains; a.slot_base = 0; }   // (a split launch in batches says which slots are its own)
    const size_t lds = pmdi_sweep_lds_bytes(a, T);
    const void *fn = sweep_kernel_for(a, T);
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    // the argument block lives in device memory (stream-ordered copy), the kernel gets a pointer:
    // cold-path device functions then read what they need instead of holding it in registers.  Through a pinned staging
    // slot the copy is asynchronous for the host too (from a stack object it is not: the runtime copies pageable memory
    // before it returns).
    const SweepArgs *src = &a;
    if (staging) { *staging = a; src = staging; }
    e = hipMemcpyAsync(d_args, src, sizeof(SweepArgs), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    const SweepArgs *ap = d_args;
    void *args[] = {(void *)&ap};
    // split mode: K workgroups per chain slot, slots dealt in groups of eight so that blocks b, b + 8, ... (one XCD
    // under round-robin placement) belong to one chain
    const unsigned grid = a.ksplit ? (unsigned)((n_chains + 7) / 8) * 8u * (unsigned)a.K : (unsigned)n_chains;
    e = hipLaunchKernel(fn, dim3(grid), dim3(T), args, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
