// pmdi_sweep2.hip -- the conditional-SMC sweep for settled chains as a gfx950 kernel: the device code is pmdi_sweep2_body.h (design
// notes there and in DESIGN.md); this file instantiates it per (datasets, particles per lane) and launches it.
// Compile with -ffp-contract=off.
// The general sweep kernel's device code: a chain this kernel's tables no longer hold is carried on by it IN PLACE -- same workgroup,
// same dynamic LDS (the launch asks for the larger of the two layouts), from the observation of the hand-over -- instead of
// waiting for another launch behind this one.
// (-DPM2_NO_RESUME_GENERAL: tests/test_build_budget.py measures this kernel's own spills -- the compiler's figure folds callees in)
#ifndef PM2_NO_RESUME_GENERAL
#define PMDI_BSLOT_FROM_TICKET 1
#include "pmdi_sweep_body.h"
template <int K, int NW>
__device__ __noinline__ void pmdi_resume_general(const SweepArgs *ap)
{
    pmdi_sweep_body<64 * NW, 2, K == 1, false, true>(ap);
}
#define PM2_RESUME_GENERAL(K_, NW_, ap_) pmdi_resume_general<K_, NW_>(ap_)
#endif
#include "pmdi_sweep2_body.h"

namespace {

// Two waves per SIMD whatever the workgroup's width (64 * NW threads: two 256-thread workgroups per CU, or one of 512) -- the second
// launch bound is hipcc's waves-per-SIMD floor -- so a wave may use up to 256 VGPRs.  The particle state and the cluster cache of a wave live in registers for the whole sweep; the builds' spill
// counts are pinned by tests/test_build_budget.py.
template <int K, int PPL, int NW, bool GO>
__global__ void __launch_bounds__(64 * NW, 2) pmdi_sweep2_kernel(const SweepArgs *__restrict__ ap)
{
    const SweepArgs &a = *ap;
    int bslot = (int)blockIdx.x;
    if (a.ticket) {
        // the workgroup's position in the launch order is drawn, not blockIdx.x (SweepArgs::ticket): one atomic per workgroup; the
        // general kernel's code finds it under 1 + blockIdx.x if this workgroup has to carry its chain on
        int *tk = (int *)pm2_smem_;
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ticket + 1 + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *tk = t;
        }
        __syncthreads();
        bslot = *tk;
        __syncthreads();
        if (bslot < 0 || bslot >= a.n_slots) return;
    }
    const int chain = a.chain_order ? a.chain_order[bslot] : bslot;
    // the chains of a sweep are shared out between launches by what their previous sweep looked like (pmdi_api.cpp)
    if (a.group_flag && ((int)a.group_flag[chain] != a.group_sel || bslot < a.rank_lo || bslot >= a.rank_hi)) return;
    pmdi_s2::Sweep2<K, PPL, NW, GO> s;
    s.run(ap, chain);
}

// the instantiations: P = 256 / 512 / 1024 particles on four waves (1, 2, 4 particles per lane), P = 2048 on eight waves (4 per lane)
// (gauss_only: every dataset of the handle is Gaussian -- the build without the integer cluster types' code; 4-wave shapes only)
template <int K>
const void *kernel_for_k(int P, int *nw, bool gauss_only)
{
    *nw = 4;
    if (gauss_only) {
        if (P == 256) return (const void *)pmdi_sweep2_kernel<K, 1, 4, true>;
        if (P == 512) return (const void *)pmdi_sweep2_kernel<K, 2, 4, true>;
        if (P == 1024) return (const void *)pmdi_sweep2_kernel<K, 4, 4, true>;
    }
    if (P == 256) return (const void *)pmdi_sweep2_kernel<K, 1, 4, false>;
    if (P == 512) return (const void *)pmdi_sweep2_kernel<K, 2, 4, false>;
    if (P == 1024) return (const void *)pmdi_sweep2_kernel<K, 4, 4, false>;
    *nw = 8;
    if (P == 2048) return (const void *)pmdi_sweep2_kernel<K, 4, 8, false>;
    return nullptr;
}

const void *kernel_for(int K, int P, int *nw, bool gauss_only = false)
{
    switch (K) {
    case 1: return kernel_for_k<1>(P, nw, gauss_only);
    case 2: return kernel_for_k<2>(P, nw, gauss_only);
    case 3: return kernel_for_k<3>(P, nw, gauss_only);
    case 4: return kernel_for_k<4>(P, nw, gauss_only);
    }
    return nullptr;
}

bool all_gaussian(const SweepArgs &a)
{
    for (int k = 0; k < a.K; ++k) if (a.ds[k].kind != K_GAUSSIAN) return false;
    return true;
}

}  // namespace

void pmdi_sweep2_layout(int K, int N, int P, int Dmax, int cols_l, int idcap, int cls, int cdfl, S2Layout *L)
{
    pmdi_s2::make_layout(K, N, P, Dmax, cols_l, idcap, cls, cdfl, *L);
}

// particle classes per dataset the class slots of that shape's lanes can name (the LDS tables may hold fewer: S2Layout::cls)
int pmdi_sweep2_max_classes(int K, int P)
{
    int nw = 0;
    if (!kernel_for(K, P, &nw)) return 0;
    return pmdi_s2::class_slots_max(K, P / (64 * nw), nw);
}

// whether that shape's build lets the mutation-CDF rows of the class slots beyond S2Layout::cdfl continue in the chain's arena
bool pmdi_sweep2_cdf_arena(int K, int P)
{
    int nw = 0;
    return kernel_for(K, P, &nw) && pmdi_s2::cdf_rows_in_arena(nw);
}

int pmdi_sweep2_threads(int K, int P)
{
    int nw = 0;
    return kernel_for(K, P, &nw) ? 64 * nw : 0;
}

// shapes the kernel is built for (the rest stays with pmdi_sweep.hip)
bool pmdi_sweep2_supports(int K, int N, int P, int Dmax, long long cap)
{
    (void)cap;      // (cluster ids travel as 16-bit values in the LDS tables: a chain whose ids would outgrow them is given back, pmdi_sweep2_body.h)
    int nw;
    return K >= 1 && K <= pmdi_s2::KMAX2 && N >= 2 && N <= 64 && Dmax <= 64 && kernel_for(K, P, &nw) != nullptr;
}

// dynamic LDS of a launch: this kernel's layout, or the general kernel's for the same workgroup width when a handed-over chain is
// carried on in place and that layout is the larger one
static size_t launch_lds(const SweepArgs &a, int nw)
{
    size_t lds = (size_t)a.s2.total;
    if (a.resume) { const size_t g = pmdi_sweep_lds_bytes(a, 64 * nw); if (g > lds) lds = g; }
    return lds;
}

hipError_t pmdi_sweep2_blocks_per_cu(const SweepArgs &a, int *blocks)
{
    int nw = 0;
    const void *fn = kernel_for(a.K, a.P, &nw, all_gaussian(a));
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = launch_lds(a, nw);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, fn, 64 * nw, lds);
}

hipError_t pmdi_launch_sweep2(const SweepArgs &a_in, SweepArgs *d_args, int n_chains, hipStream_t stream, SweepArgs *staging)
{
    SweepArgs a = a_in;
    a.n_slots = n_chains;
    int nw = 0;
    const void *fn = kernel_for(a.K, a.P, &nw, all_gaussian(a));
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = launch_lds(a, nw);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const SweepArgs *src = &a;
    if (staging) { *staging = a; src = staging; }
    e = hipMemcpyAsync(d_args, src, sizeof(SweepArgs), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    if (a.ticket) {
        e = hipMemsetAsync(a.ticket, 0, sizeof(int), stream);
        if (e != hipSuccess) return e;
    }
    const SweepArgs *ap = d_args;
    void *args[] = {(void *)&ap};
    e = hipLaunchKernel(fn, dim3((unsigned)n_chains), dim3(64u * (unsigned)nw), args, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
