// Does gfx950 skip VALU passes for a wave64 whose upper lanes are all inactive?
// Same f64 FMA chain, 8 waves/SIMD, with 64 / 32 / 16 active lanes per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, CHAINS = 8;
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int active)
{
    if ((threadIdx.x & 63) >= active) return;
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = a + threadIdx.x * 1e-9 + c;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], b, a);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    double* d; CK(hipMalloc(&d, (size_t)256 * 8 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps : {1, 2, 8}) for (int active : {64, 48, 32, 16}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0, 0.999999, active);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0, 0.999999, active);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("waves/SIMD=%d active lanes=%2d: %.3f ms per launch\n", wps, active, ms / 10);
    }
    return 0;
}
