// FETCH_SIZE / WRITE_SIZE calibration on the sweep kernel's own access patterns (development aid, not part of the product).
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads; other widths
// are uncalibrated -- "calibrate on a known byte count in your own access pattern".  Each kernel below moves a known number of
// bytes out of a 2 GiB table (far beyond L2 and the 256 MiB Infinity Cache); run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- ./traffic     (and once more with WRITE_SIZE)
// and compare the per-kernel counter with the byte counts this program prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double dbl2v __attribute__((ext_vector_type(2)));

// (a) the pool gather: every lane reads one 16-byte (Sigma, beta) pair of a random cluster row, consecutive lanes consecutive features
__global__ void gather16(const dbl2v *tab, const unsigned *row, int D, long long items, double *sink)
{
    double acc = 0.0;
    for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
        const long long j = it / D; const int q = (int)(it - j * D);
        const dbl2v v = tab[(size_t)row[j] * D + q];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) sink[0] = acc;
}
// (b) the resampling gather: lane p reads 4 bytes of row nn at column anc[p] (sorted ancestors with repeats), writes column p
__global__ void gather4(const int *src, int *dst, const int *anc, int N, int P, int chains, int *sink)
{
    for (int c = blockIdx.x; c < chains; c += gridDim.x) {
        const int *s = src + (size_t)c * N * P; int *d = dst + (size_t)c * N * P; const int *a = anc + (size_t)c * P;
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            const int an = a[p];
            for (int nn = 0; nn < N; ++nn) d[nn * P + p] = s[nn * P + an];
        }
    }
    if (sink[0] == 123456789) sink[1] = 1;
}
// (c) the pool update: read-modify-write of one 16-byte pair per lane of a random row
__global__ void rmw16(dbl2v *tab, const unsigned *row, int D, long long items)
{
    for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
        const long long j = it / D; const int q = (int)(it - j * D);
        dbl2v v = tab[(size_t)row[j] * D + q];
        v.x += 1.0; v.y += 0.5;
        tab[(size_t)row[j] * D + q] = v;
    }
}
// (d) the history byte: lane p writes one byte of a fresh row (P consecutive bytes per step)
__global__ void bytes1(unsigned char *hist, int P, long long steps)
{
    for (long long st = blockIdx.x; st < steps; st += gridDim.x)
        for (int p = threadIdx.x; p < P; p += blockDim.x) hist[(size_t)st * P + p] = (unsigned char)(p + st);
}

int main()
{
    const int D = 50, N = 20, P = 1024, chains = 2048;
    const size_t rows = 2684354;                          // x 50 features x 16 B = 2 GiB
    dbl2v *tab; unsigned *row; double *sink; int *src, *dst, *anc, *isink; unsigned char *hist;
    const long long nrows_touched = 4000000;              // random rows gathered
    CHECK(hipMalloc(&tab, rows * D * 16)); CHECK(hipMemset(tab, 0, rows * D * 16));
    CHECK(hipMalloc(&row, nrows_touched * 4)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&isink, 64)); CHECK(hipMemset(isink, 0, 64));
    unsigned *hrow = (unsigned *)malloc(nrows_touched * 4);
    unsigned long long x = 88172645463325252ull;
    for (long long i = 0; i < nrows_touched; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hrow[i] = (unsigned)(x % rows); }
    CHECK(hipMemcpy(row, hrow, nrows_touched * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&src, (size_t)chains * N * P * 4)); CHECK(hipMalloc(&dst, (size_t)chains * N * P * 4)); CHECK(hipMalloc(&anc, (size_t)chains * P * 4));
    CHECK(hipMemset(src, 0, (size_t)chains * N * P * 4));
    int *hanc = (int *)malloc((size_t)chains * P * 4);
    for (int c = 0; c < chains; ++c) { int a = 0; for (int p = 0; p < P; ++p) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; if (p && (x & 3)) a += (int)(x >> 60) % 3; if (a >= P) a = P - 1; hanc[(size_t)c * P + p] = p ? a : 0; } }
    CHECK(hipMemcpy(anc, hanc, (size_t)chains * P * 4, hipMemcpyHostToDevice));
    const long long steps = 2000000;
    CHECK(hipMalloc(&hist, (size_t)steps * P));
    CHECK(hipDeviceSynchronize());
    const long long items = nrows_touched * D;
    hipLaunchKernelGGL(gather16, dim3(4096), dim3(256), 0, 0, tab, row, D, items, sink);
    hipLaunchKernelGGL(gather4, dim3(2048), dim3(256), 0, 0, src, dst, anc, N, P, chains, isink);
    hipLaunchKernelGGL(rmw16, dim3(4096), dim3(256), 0, 0, tab, row, D, items);
    hipLaunchKernelGGL(bytes1, dim3(4096), dim3(256), 0, 0, hist, P, steps);
    CHECK(hipDeviceSynchronize());
    printf("gather16: %lld bytes read (rows of %d B = 6.25 lines of 128 B)\n", items * 16, D * 16);
    printf("gather4 : %lld bytes read, %lld bytes written (+ %lld ancestors)\n", (long long)chains * N * P * 4, (long long)chains * N * P * 4, (long long)chains * P * 4);
    printf("rmw16   : %lld bytes read, %lld bytes written\n", items * 16, items * 16);
    printf("bytes1  : %lld bytes written\n", steps * P);
    return 0;
}
