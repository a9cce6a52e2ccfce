// Latency micro-benchmarks (development aid, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chase(const int* next, int start, int iters, long long* out, int* sink) {
    int p = start; long long t0 = clock64();
    for (int i = 0; i < iters; ++i) p = next[p];
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = p; }
}
__global__ void chase_vec(const int* next, int iters, long long* out, int* sink) {   // per-lane (vector) loads
    int p = threadIdx.x; long long t0 = clock64();
    for (int i = 0; i < iters; ++i) p = next[p];
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0; sink[threadIdx.x] = p;
}
__global__ void lds_chase(int iters, long long* out, int* sink) {
    __shared__ int a[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) a[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    int p = threadIdx.x; long long t0 = clock64();
    for (int i = 0; i < iters; ++i) p = a[p];
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0; sink[threadIdx.x] = p;
}
__global__ void barrier_cost(int iters, long long* out) {
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void store_barrier(int* buf, int iters, long long* out) {   // store + barrier (vmcnt(0) wait)
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) { buf[threadIdx.x + 1024 * (i & 7)] = i; __syncthreads(); }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void fp64_chain(int iters, double x, long long* out, double* sink) {
    double a = x; long long t0 = clock64();
    for (int i = 0; i < iters; ++i) a = log(a + 1.5);
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0; sink[threadIdx.x] = a;
}
__global__ void div_chain(int iters, double x, long long* out, double* sink) {
    double a = x; long long t0 = clock64();
    for (int i = 0; i < iters; ++i) a = 1.0 + 3.0 / a;
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0; sink[threadIdx.x] = a;
}
__global__ void add_chain(int iters, double x, long long* out, double* sink) {
    double a = x; long long t0 = clock64();
    for (int i = 0; i < iters; ++i) a = a + x;
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0; sink[threadIdx.x] = a;
}
int main() {
    const int N = 1 << 22;   // 16 MB of ints: beyond L2 (4 MB), inside Infinity Cache
    std::vector<int> h(N);
    for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 1048583 + 12345) % N);
    int *d, *sink; long long* out; double* dsink;
    hipMalloc(&d, N * 4); hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, 4096 * 4); hipMalloc(&out, 64); hipMalloc(&dsink, 4096 * 8);
    long long r;
    auto get = [&]() { hipDeviceSynchronize(); hipMemcpy(&r, out, 8, hipMemcpyDeviceToHost); return r; };
    const int IT = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0, IT, out, sink); printf("scalar-path dependent load (16MB footprint): %.0f cyc\n", (double)get() / IT);
        hipLaunchKernelGGL(chase_vec, dim3(1), dim3(64), 0, 0, d, IT, out, sink); printf("vector dependent load, 64 lanes scattered (16MB): %.0f cyc\n", (double)get() / IT);
    }
    // small footprint (fits L2 / L1)
    std::vector<int> h2(4096); for (int i = 0; i < 4096; ++i) h2[i] = (i * 37 + 11) & 4095;
    hipMemcpy(d, h2.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chase_vec, dim3(1), dim3(64), 0, 0, d, IT, out, sink); printf("vector dependent load, 16KB footprint (L1/L2 hit): %.0f cyc\n", (double)get() / IT);
    hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0, IT, out, sink); printf("scalar dependent load, 16KB footprint: %.0f cyc\n", (double)get() / IT);
    hipLaunchKernelGGL(lds_chase, dim3(1), dim3(64), 0, 0, IT, out, sink); printf("LDS dependent read: %.0f cyc\n", (double)get() / IT);
    for (int T : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(barrier_cost, dim3(1), dim3(T), 0, 0, IT, out); printf("__syncthreads T=%d: %.0f cyc\n", T, (double)get() / IT);
        hipLaunchKernelGGL(store_barrier, dim3(1), dim3(T), 0, 0, sink, IT, out); printf("global store + __syncthreads T=%d: %.0f cyc\n", T, (double)get() / IT);
    }
    hipLaunchKernelGGL(fp64_chain, dim3(1), dim3(64), 0, 0, IT, 2.0, out, dsink); printf("fp64 log chain: %.0f cyc\n", (double)get() / IT);
    hipLaunchKernelGGL(div_chain, dim3(1), dim3(64), 0, 0, IT, 2.0, out, dsink); printf("fp64 div chain: %.0f cyc\n", (double)get() / IT);
    hipLaunchKernelGGL(add_chain, dim3(1), dim3(64), 0, 0, IT, 2.0, out, dsink); printf("fp64 add chain: %.0f cyc\n", (double)get() / IT);
    return 0;
}
