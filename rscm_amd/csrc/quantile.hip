// Ensemble quantiles per time index for gfx950 (MI355X): the plume of a variable (median, 5-95 %
// band ...) without moving the series to the host.  Not part of the reference (which has no ensemble
// statistics); the definition is numpy's: nanquantile(series[t, :], q, method="linear") --
// NaN members are left out, virtual index (n - 1) q, linear interpolation between the two
// neighbouring order statistics with numpy's _lerp (function_base.py), so the results carry
// numpy's bits.
//
// Rows [T][N] are sorted segment-wise with rocPRIM's segmented radix sort (one segment per time
// index) after NaNs of either sign are made +NaN, which the radix order places after +inf; each
// (row, q) thread then finds the number of non-NaN entries by bisection and interpolates.
// HBM-bound: a few passes over 8 B per member per row; rows are processed in chunks so that the
// scratch stays below ~4 GiB whatever the ensemble size.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "rscm_device.hpp"

namespace rscm {

namespace {

__global__ __launch_bounds__(kBlock) void canon_nan_kernel(const double* __restrict__ in, double* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    out[i] = x != x ? __builtin_nan("") : x;
}

__global__ void offsets_kernel(unsigned* offs, int32_t n_rows, int64_t N)
{
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r <= n_rows) offs[r] = (unsigned)((int64_t)r * N);
}

// out[r][0] = number of non-NaN members, out[r][1 + k] = quantile q[k] of them (NaN if none)
__global__ void quantile_kernel(const double* __restrict__ sorted, int64_t N, int32_t n_rows, const double* __restrict__ q, int32_t n_q,
                                double* __restrict__ out)
{
    const int32_t idx = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= n_rows * (n_q + 1)) return;
    const int32_t r = idx / (n_q + 1), k = idx % (n_q + 1);
    const double* row = sorted + (size_t)r * N;
    int64_t lo = 0, hi = N;  // first NaN (they sort last)
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (row[mid] != row[mid]) hi = mid;
        else lo = mid + 1;
    }
    const int64_t n = lo;
    if (k == 0) {
        out[(size_t)r * (n_q + 1)] = (double)n;
        return;
    }
    if (n == 0) {
        out[(size_t)r * (n_q + 1) + k] = __builtin_nan("");
        return;
    }
    const double t = q[k - 1];
    const double vi = (double)(n - 1) * t;       // numpy _compute_virtual_index(n, q, 1, 1)
    double prev = floor(vi);
    if (prev < 0.0) prev = 0.0;
    if (prev > (double)(n - 1)) prev = (double)(n - 1);
    const int64_t ip = (int64_t)prev;
    const int64_t in_ = ip + 1 < n ? ip + 1 : n - 1;
    const double g = vi - prev;
    const double a = row[ip], b = row[in_];
    const double d = b - a;
    double v = a + d * g;                          // numpy _lerp
    if (g >= 0.5) v = b - d * (1.0 - g);
    if (d == 0.0) v = a;                           // ... where(diff_b_a == 0, a, lerp): also keeps inf - inf out
    out[(size_t)r * (n_q + 1) + k] = v;
}

}  // namespace

// rows: [n_rows][N] on the device; d_q: [n_q]; d_out: [n_rows][n_q + 1]
hipError_t launch_quantile_rows(const double* rows, int64_t N, int32_t n_rows, const double* d_q, int32_t n_q, double* d_out,
                                hipStream_t s)
{
    if (n_rows <= 0 || N <= 0) return hipSuccess;
    int64_t per_chunk = ((int64_t)1 << 28) / N;  // 2^28 doubles = 2 GiB per scratch buffer
    if (per_chunk < 1) per_chunk = 1;
    if (per_chunk > n_rows) per_chunk = n_rows;
    if ((uint64_t)per_chunk * (uint64_t)N > 0xFFFFFFF0ull) return hipErrorInvalidValue;  // rocPRIM sizes are 32-bit
    double *canon = nullptr, *sorted = nullptr;
    unsigned* offs = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t rc) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(canon);
        (void)hipFree(sorted);
        (void)hipFree(offs);
        (void)hipFree(tmp);
        return rc;
    };
    const size_t chunk_elems = (size_t)per_chunk * (size_t)N;
    if ((e = hipMalloc(&canon, chunk_elems * sizeof(double))) != hipSuccess) return done(e);
    if ((e = hipMalloc(&sorted, chunk_elems * sizeof(double))) != hipSuccess) return done(e);
    if ((e = hipMalloc(&offs, (size_t)(per_chunk + 1) * sizeof(unsigned))) != hipSuccess) return done(e);
    hipLaunchKernelGGL(offsets_kernel, dim3((unsigned)((per_chunk + 1 + 255) / 256)), dim3(256), 0, s, offs, (int32_t)per_chunk, N);
    e = rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, canon, sorted, (unsigned)chunk_elems, (unsigned)per_chunk, offs, offs + 1, 0,
                                           64, s);
    if (e != hipSuccess) return done(e);
    if ((e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8)) != hipSuccess) return done(e);
    for (int64_t r0 = 0; r0 < n_rows; r0 += per_chunk) {
        const int64_t nr = r0 + per_chunk <= n_rows ? per_chunk : n_rows - r0;
        const int64_t elems = nr * N;
        hipLaunchKernelGGL(canon_nan_kernel, dim3((unsigned)((elems + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           rows + (size_t)r0 * N, canon, elems);
        e = rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, canon, sorted, (unsigned)elems, (unsigned)nr, offs, offs + 1, 0, 64, s);
        if (e != hipSuccess) return done(e);
        const int64_t threads = nr * (n_q + 1);
        hipLaunchKernelGGL(quantile_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, sorted, N, (int32_t)nr, d_q, n_q,
                           d_out + (size_t)r0 * (n_q + 1));
        if ((e = hipGetLastError()) != hipSuccess) return done(e);
    }
    return done(hipSuccess);
}

}  // namespace rscm
