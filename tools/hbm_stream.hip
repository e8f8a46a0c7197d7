// HBM stream rates on one MI355X, the yardstick beside the nominal 8 TB/s ("to be re-measured on the
// box"): f64 copy, triad, write-only and read-only sums over 2 GiB arrays, plus the access pattern of the
// ensemble kernels -- every wavefront writing one 512-byte segment of each of R rows per "model step".
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_stream.hip -o tools/hbm_stream && tools/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const double* __restrict__ a, double* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_triad(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, double s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i] + s * b[i];
}
__global__ void k_write(double* __restrict__ a, double v, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v;
}
__global__ void k_read(const double* __restrict__ a, double* __restrict__ out, size_t n)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 12345.678) out[0] = s;  // keeps the loads
}
// one thread per member, `steps` rows of `vars` series each: series[v][t][N] (the two-layer kernel's stores, no arithmetic)
__global__ void k_rows(double* __restrict__ series, size_t N, int steps, int vars)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double x = (double)i;
    for (int t = 0; t < steps; ++t) {
        for (int v = 0; v < vars; ++v) series[((size_t)v * steps + t) * N + i] = x;
        x += 1.0;
    }
}

template <class F>
static double timed(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e-3;
}

int main()
{
    const size_t n = (size_t)1 << 28;  // 2 GiB per array
    double *a, *b, *c;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8));
    const dim3 block(256), grid(256 * 32);
    hipLaunchKernelGGL(k_write, grid, block, 0, 0, a, 1.0, n);
    hipLaunchKernelGGL(k_write, grid, block, 0, 0, b, 2.0, n);
    CK(hipDeviceSynchronize());
    double t;
    t = timed([&] { hipLaunchKernelGGL(k_copy, grid, block, 0, 0, a, c, n); }, 10);
    printf("copy   (8 B read + 8 B written per element, 2 GiB arrays): %7.1f GB/s\n", 16.0 * n / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_triad, grid, block, 0, 0, a, b, c, 3.0, n); }, 10);
    printf("triad  (16 B read + 8 B written):                          %7.1f GB/s\n", 24.0 * n / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_write, grid, block, 0, 0, c, 3.0, n); }, 10);
    printf("write  (8 B written):                                      %7.1f GB/s\n", 8.0 * n / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_read, grid, block, 0, 0, a, c, n); }, 10);
    printf("read   (8 B read):                                         %7.1f GB/s\n", 8.0 * n / t / 1e9);
    for (size_t N : {(size_t)100000, (size_t)1000000}) {
        for (int vars : {2, 7}) {
            const int steps = 750;
            double* rows;
            CK(hipMalloc(&rows, (size_t)vars * steps * N * 8));
            t = timed([&] { hipLaunchKernelGGL(k_rows, dim3((unsigned)((N + 255) / 256)), block, 0, 0, rows, N, steps, vars); }, 5);
            CK(hipFree(rows));
            printf("rows   (%d series x %d rows x %zu members, one thread per member, store only): %7.1f GB/s  (%.3f ms)\n", vars, steps, N,
                   8.0 * vars * steps * N / t / 1e9, t * 1e3);
        }
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c));
    return 0;
}
