// Measures what one MI355X actually sustains for the instruction mix of the two-layer kernel:
// f64 FMA / ADD / MUL issue rates, 32-bit integer VALU, and the shader clock under that load.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int CHAINS = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, unsigned long long* clk)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = a + threadIdx.x * 1e-9 + c;
    unsigned u[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) u[c] = threadIdx.x + c;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) x[c] = __builtin_fma(x[c], b, a);
            if (OP == 1) x[c] = x[c] + a;
            if (OP == 2) x[c] = x[c] * b;
            if (OP == 3) { x[c] = x[c] * b; x[c] = x[c] + a; }  // unfused mul+add pair
            if (OP == 4) { u[c] = (u[c] & 0x7ff00000u) + 0xf0000000u; u[c] ^= i; }
            if (OP == 5) x[c] = x[c] / b;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0; unsigned v = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { s += x[c]; v += u[c]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + v;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
int run(const char* name, int ops_per_iter, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves)
    double* d; unsigned long long* c;
    CK(hipMalloc(&d, (size_t)blocks * 256 * 8)); CK(hipMalloc(&c, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 0.999999, c);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 0.999999, c);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    unsigned long long hc[2]; CK(hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost));
    const double instr = (double)blocks * 4 /*waves*/ * ITER * CHAINS * ops_per_iter;  // wave-instructions
    const double ghz = (double)hc[0] / ((double)hc[1] / 100e6) / 1e9;
    const double cyc_per_instr_per_simd = (ms * 1e-3 * ghz * 1e9) / (instr / 1024.0);
    printf("%-22s waves/SIMD=%d  %.3f ms  clock %.2f GHz  %.2f cyc/wave-instr/SIMD  %.2f T lane-ops/s\n", name,
           waves_per_simd, ms, ghz, cyc_per_instr_per_simd, instr * 64 / (ms * 1e-3) / 1e12);
    CK(hipFree(d)); CK(hipFree(c));
    return 0;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f64", 1, w);
        run<1>("v_add_f64", 1, w);
        run<2>("v_mul_f64", 1, w);
        run<3>("v_mul_f64+v_add_f64", 2, w);
        run<4>("int and+add+xor", 3, w);
        run<5>("f64 divide (11 instr)", 1, w);
    }
    return 0;
}
