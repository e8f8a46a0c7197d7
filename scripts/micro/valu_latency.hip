// Micro-benchmark (gfx950, run on the GPU box): what one wavefront per SIMD pays for DEPENDENT f64 instructions.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_latency scripts/micro/valu_latency.hip && /tmp/valu_latency
// Prints shader cycles (s_memtime runs at 100 MHz: wall time x the measured clock is used instead) per instruction for
// chains of 1, 2 and 4 independent streams of v_fma_f64, for v_rcp_f64, for an LDS write + read of a double, and per row of a
// Thomas-sweep c' chain with 0-18 independent instructions beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int STREAMS>
__global__ __launch_bounds__(64) void fma_chain(double* out, int iters, double a, double b)
{
    double x[STREAMS];
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) x[s] = (double)threadIdx.x + s;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / STREAMS; ++r)
#pragma unroll
            for (int s = 0; s < STREAMS; ++s) x[s] = __builtin_fma(x[s], a, b);
    }
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) acc += x[s];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int STREAMS>
__global__ __launch_bounds__(64) void rcp_chain(double* out, int iters)
{
    double x[STREAMS];
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) x[s] = 1.5 + (double)threadIdx.x + s;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / STREAMS; ++r)
#pragma unroll
            for (int s = 0; s < STREAMS; ++s) x[s] = __builtin_amdgcn_rcp(x[s]);
    }
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) acc += x[s];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// rcp followed by independent FMAs: does the quarter-rate instruction block the FMAs behind it?
__global__ __launch_bounds__(64) void rcp_mixed(double* out, int iters, double a, double b)
{
    double x = 1.5 + threadIdx.x, y0 = 1.0, y1 = 2.0, y2 = 3.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x = __builtin_amdgcn_rcp(x);
            y0 = __builtin_fma(y0, a, b);
            y1 = __builtin_fma(y1, a, b);
            y2 = __builtin_fma(y2, a, b);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + y0 + y1 + y2;
}

// A Thomas-sweep row in miniature: the c' chain (fma -> v_rcp_f64 -> fma -> fma -> fma, with one multiply beside it) and INDEP
// independent fmas around it.  If the chain's latency is covered, a row costs the sum of its issue slots.
template <int INDEP>
__global__ __launch_bounds__(64) void thomas_row(double* out, int iters, double a, double b, double k)
{
    double c = 0.25 + 1e-3 * threadIdx.x;
    double y[INDEP > 0 ? INDEP : 1];
#pragma unroll
    for (int j = 0; j < INDEP; ++j) y[j] = 1.0 + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double denom = __builtin_fma(-a, c, b);
            const double r0 = __builtin_amdgcn_rcp(denom);
            const double e = __builtin_fma(-denom, r0, 1.0);
            const double u = __builtin_fma(e, e, e);
            const double t = k * r0;
            c = __builtin_fma(t, u, t);
#pragma unroll
            for (int j = 0; j < INDEP; ++j) y[j] = __builtin_fma(y[j], 0.999, 0.001);
        }
    }
    double acc = c;
#pragma unroll
    for (int j = 0; j < INDEP; ++j) acc += y[j];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void lds_swap(double* out, int iters)
{
    __shared__ double park[32][64];
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { v[k] = threadIdx.x + k; park[k][threadIdx.x] = v[k] * 2.0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const double t = park[k][threadIdx.x];
            park[k][threadIdx.x] = v[k];
            v[k] = t + 1.0;
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += v[k];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <class F>
static double time_ms(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int blocks = 1024, iters = 20000;   // one wavefront per SIMD
    double* out;
    hipMalloc(&out, 2 * blocks * 64 * sizeof(double));   // the last line launches 2 x blocks
    const double ghz = 2.4;   // nominal; the ratios between the lines are what matters
    auto report = [&](const char* what, double ms, double n_inst) {
        printf("%-46s %8.3f ms  %6.2f ns/inst  ~%5.1f cycles/inst at %.1f GHz\n", what, ms, ms * 1e6 / n_inst, ms * 1e6 / n_inst * ghz, ghz);
    };
    report("v_fma_f64, 1 dependent chain", time_ms([&] { hipLaunchKernelGGL(fma_chain<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999, 0.001); }), 64.0 * iters);
    report("v_fma_f64, 2 independent chains", time_ms([&] { hipLaunchKernelGGL(fma_chain<2>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999, 0.001); }), 64.0 * iters);
    report("v_fma_f64, 4 independent chains", time_ms([&] { hipLaunchKernelGGL(fma_chain<4>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999, 0.001); }), 64.0 * iters);
    report("v_rcp_f64, 1 dependent chain", time_ms([&] { hipLaunchKernelGGL(rcp_chain<1>, dim3(blocks), dim3(64), 0, 0, out, iters); }), 64.0 * iters);
    report("v_rcp_f64, 4 independent chains", time_ms([&] { hipLaunchKernelGGL(rcp_chain<4>, dim3(blocks), dim3(64), 0, 0, out, iters); }), 64.0 * iters);
    report("1 v_rcp_f64 + 3 v_fma_f64 (per group of 4)", time_ms([&] { hipLaunchKernelGGL(rcp_mixed, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999, 0.001); }), 16.0 * iters);
    report("LDS read + write of a double (per pair)", time_ms([&] { hipLaunchKernelGGL(lds_swap, dim3(blocks), dim3(64), 0, 0, out, iters); }), 32.0 * iters);
    report("Thomas row: c' chain alone (5 VALU + rcp)", time_ms([&] { hipLaunchKernelGGL(thomas_row<0>, dim3(blocks), dim3(64), 0, 0, out, iters / 4, 0.3, 1.5, 0.4); }), 16.0 * (iters / 4));
    report("Thomas row: chain + 6 independent fma", time_ms([&] { hipLaunchKernelGGL(thomas_row<6>, dim3(blocks), dim3(64), 0, 0, out, iters / 4, 0.3, 1.5, 0.4); }), 16.0 * (iters / 4));
    report("Thomas row: chain + 12 independent fma", time_ms([&] { hipLaunchKernelGGL(thomas_row<12>, dim3(blocks), dim3(64), 0, 0, out, iters / 4, 0.3, 1.5, 0.4); }), 16.0 * (iters / 4));
    report("Thomas row: chain + 18 independent fma", time_ms([&] { hipLaunchKernelGGL(thomas_row<18>, dim3(blocks), dim3(64), 0, 0, out, iters / 4, 0.3, 1.5, 0.4); }), 16.0 * (iters / 4));
    // the same with two wavefronts per SIMD
    report("v_fma_f64, 1 chain, 2 wavefronts per SIMD", time_ms([&] { hipLaunchKernelGGL(fma_chain<1>, dim3(2 * blocks), dim3(64), 0, 0, out, iters, 0.999, 0.001); }), 2 * 64.0 * iters);
    hipFree(out);
    return 0;
}
