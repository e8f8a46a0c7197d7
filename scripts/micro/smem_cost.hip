// Micro-benchmark (gfx950, run on the GPU box): what a wave-uniform table read with scalar loads costs one wavefront per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/smem_cost scripts/micro/smem_cost.hip && /tmp/smem_cost
// Each iteration walks a table of ROWS rows of W doubles (uniform address: s_load_dwordx{2W}), requests row r + 1 before it uses row r
// (one s_waitcnt lgkmcnt(0) per row) and spends FMAS dependent v_fma_f64 (2.07 ns each) per row on the values.  Printed: ns per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W, int FMAS>
__global__ __launch_bounds__(64) void walk(const double* __restrict__ table, double* out, int iters, int)
{
    constexpr int rows = 48;   // unrolled, as the column sweep is: no loop counter, no register copies between the rows
    double x = 1.0 + threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
        int32_t opaque = 0;
        asm volatile("" : "+s"(opaque));
        const double* __restrict__ tab = table + opaque;
        double cur[W], nxt[W];
#pragma unroll
        for (int k = 0; k < W; ++k) cur[k] = tab[k];
#pragma unroll
        for (int r = 0; r < rows; ++r) {
#pragma unroll
            for (int k = 0; k < W; ++k) asm volatile("" ::"s"(cur[k]));
#pragma unroll
            for (int k = 0; k < W; ++k) nxt[k] = tab[(size_t)(r + 1) * W + k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < FMAS; ++f) x = __builtin_fma(x, cur[f % W], 1e-9);
#pragma unroll
            for (int k = 0; k < W; ++k) cur[k] = nxt[k];
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
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
    const int blocks = 1024, iters = 2000, rows = 48;
    std::vector<double> h(64 * 16, 0.999);
    double *table, *out;
    hipMalloc(&table, h.size() * sizeof(double));
    hipMemcpy(table, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMalloc(&out, blocks * 64 * sizeof(double));
    auto report = [&](const char* what, double ms, int fmas) {
        const double per_row = ms * 1e6 / ((double)iters * rows);
        printf("%-44s %8.3f ms  %7.2f ns per row\n", what, ms, per_row);
    };
#define RUN(W, F) report("s_load of " #W " doubles per row + " #F " fma", time_ms([&] { hipLaunchKernelGGL((walk<W, F>), dim3(blocks), dim3(64), 0, 0, table, out, iters, rows); }), F)
    RUN(1, 20); RUN(2, 20); RUN(4, 20); RUN(6, 20); RUN(8, 20);
    RUN(2, 40); RUN(6, 40); RUN(8, 40);
    RUN(2, 80); RUN(6, 80); RUN(8, 80);
    RUN(2, 160); RUN(6, 160); RUN(8, 160);
    return 0;
}
