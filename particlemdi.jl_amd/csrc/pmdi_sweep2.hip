// pmdi_sweep2.hip -- the conditional-SMC sweep for settled chains as a gfx950 kernel: the device code is pmdi_sweep2_body.h (design
// notes there and in DESIGN.md); this file instantiates it per (datasets, particles per lane) and launches it.
// Compile with -ffp-contract=off.
#include "pmdi_sweep2_body.h"

namespace {

// Two workgroups per CU: 256 threads = one wave per SIMD each, so a SIMD hosts two waves and each may use up to 256 VGPRs.  The
// particle state and the cluster cache of a wave live in registers for the whole sweep; nothing is spilled (tests/test_build_budget.py).
template <int K, int PPL>
__global__ void __launch_bounds__(256, 2) pmdi_sweep2_kernel(const SweepArgs *__restrict__ ap)
{
    const SweepArgs &a = *ap;
    const int bslot = (int)blockIdx.x;
    const int chain = a.chain_order ? a.chain_order[bslot] : bslot;
    // the chains of a sweep are shared out between launches by what their previous sweep looked like (pmdi_api.cpp)
    if (a.group_flag && ((int)a.group_flag[chain] != a.group_sel || bslot < a.rank_lo || bslot >= a.rank_hi)) return;
    pmdi_s2::Sweep2<K, PPL> s;
    s.run(ap, chain);
}

template <int K>
const void *kernel_for_ppl(int ppl)
{
    if (ppl == 1) return (const void *)pmdi_sweep2_kernel<K, 1>;
    if (ppl == 2) return (const void *)pmdi_sweep2_kernel<K, 2>;
    if (ppl == 4) return (const void *)pmdi_sweep2_kernel<K, 4>;
    return nullptr;
}

const void *kernel_for(int K, int ppl)
{
    switch (K) {
    case 1: return kernel_for_ppl<1>(ppl);
    case 2: return kernel_for_ppl<2>(ppl);
    case 3: return kernel_for_ppl<3>(ppl);
    case 4: return kernel_for_ppl<4>(ppl);
    }
    return nullptr;
}

}  // namespace

void pmdi_sweep2_layout(int K, int N, int P, int Dmax, int cols_l, int idcap, S2Layout *L)
{
    pmdi_s2::make_layout(K, N, P, Dmax, cols_l, idcap, *L);
}

// shapes the kernel is built for (the rest stays with pmdi_sweep.hip)
bool pmdi_sweep2_supports(int K, int N, int P, int Dmax, long long cap)
{
    return K >= 1 && K <= pmdi_s2::KMAX2 && N >= 2 && N <= 64 && Dmax <= 64 && (P == 256 || P == 512 || P == 1024) && cap <= 65535;
}

hipError_t pmdi_sweep2_blocks_per_cu(const SweepArgs &a, int *blocks)
{
    const void *fn = kernel_for(a.K, a.P / 256);
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.s2.total);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, fn, 256, (size_t)a.s2.total);
}

hipError_t pmdi_launch_sweep2(const SweepArgs &a_in, SweepArgs *d_args, int n_chains, hipStream_t stream, SweepArgs *staging)
{
    SweepArgs a = a_in;
    a.n_slots = n_chains;
    const void *fn = kernel_for(a.K, a.P / 256);
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.s2.total);
    if (e != hipSuccess) return e;
    const SweepArgs *src = &a;
    if (staging) { *staging = a; src = staging; }
    e = hipMemcpyAsync(d_args, src, sizeof(SweepArgs), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    const SweepArgs *ap = d_args;
    void *args[] = {(void *)&ap};
    e = hipLaunchKernel(fn, dim3((unsigned)n_chains), dim3(256), args, (size_t)a.s2.total, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
