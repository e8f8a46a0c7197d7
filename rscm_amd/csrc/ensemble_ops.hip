// Ensemble-side device operations around the stepper kernels (gfx950):
//   * Gaussian log-likelihood per member    crates/rscm-calibrate/src/likelihood.rs:186-250
//   * ensemble summaries (count/sum/min/max) with wavefront (64-lane) shuffle reductions
//   * counter-based Latin hypercube          crates/rscm-calibrate/src/parameter_set.rs:207-233
//   * fills / row broadcast for collection initialisation (builder.rs:772-780)
//   * the division self-test behind rscm_gpu_selftest_div
#include <algorithm>

#include "philox.hpp"
#include "rk4_device.hpp"
#include "rscm_device.hpp"

namespace rscm {

namespace {

__global__ __launch_bounds__(kBlock) void fill_kernel(double* p, int64_t n, double v)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ __launch_bounds__(kBlock) void broadcast_row_kernel(double* row, int64_t n,
                                                               const double* src, int64_t n_src)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        row[i] = src[n_src == 1 ? 0 : i];
}

// ---- windowed series (RSCM_FLAG_WINDOWED) -------------------------------------------------------
// buf is [n_vars][R][N].  Slide: rows [shift, shift + keep) of every variable move to rows [0, keep)
// (shift >= keep: source and destination rows are disjoint).
__global__ __launch_bounds__(kBlock) void slide_rows_kernel(double* buf, int64_t N, int32_t R, int32_t n_vars, int32_t shift,
                                                            int32_t keep)
{
    const int64_t per_var = (int64_t)keep * N, total = per_var * n_vars;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t x = (int64_t)blockIdx.x * kBlock + threadIdx.x; x < total; x += stride) {
        const int64_t v = x / per_var, o = x - v * per_var;
        double* var = buf + (size_t)v * R * N;
        var[o] = var[(size_t)shift * N + o];
    }
}

// rows [row_begin, R) of every variable = value
__global__ __launch_bounds__(kBlock) void fill_rows_kernel(double* buf, int64_t N, int32_t R, int32_t n_vars, int32_t row_begin,
                                                           double value)
{
    const int64_t per_var = (int64_t)(R - row_begin) * N, total = per_var * n_vars;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t x = (int64_t)blockIdx.x * kBlock + threadIdx.x; x < total; x += stride) {
        const int64_t v = x / per_var, o = x - v * per_var;
        buf[(size_t)v * R * N + (size_t)row_begin * N + o] = value;
    }
}

// dst[k][dst_row][:] = src[vars[k] - 1][src_row][:] for k < n_out (or vars == nullptr: k-th variable)
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const double* src, int64_t N, int32_t src_rows, int32_t src_row,
                                                             const int32_t* vars, int32_t n_out, double* dst, int32_t dst_rows,
                                                             int32_t dst_row)
{
    const int64_t total = (int64_t)n_out * N;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t x = (int64_t)blockIdx.x * kBlock + threadIdx.x; x < total; x += stride) {
        const int64_t k = x / N, i = x - k * N;
        const int64_t v = vars ? vars[k] - 1 : k;
        dst[((size_t)k * dst_rows + dst_row) * N + i] = src[((size_t)v * src_rows + src_row) * N + i];
    }
}

// the reverse: src[vars[k] - 1][src_row][:] = dst[k][dst_row][:]
__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(double* src, int64_t N, int32_t src_rows, int32_t src_row,
                                                              const int32_t* vars, int32_t n_out, const double* dst, int32_t dst_rows,
                                                              int32_t dst_row)
{
    const int64_t total = (int64_t)n_out * N;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t x = (int64_t)blockIdx.x * kBlock + threadIdx.x; x < total; x += stride) {
        const int64_t k = x / N, i = x - k * N;
        const int64_t v = vars ? vars[k] - 1 : k;
        src[((size_t)v * src_rows + src_row) * N + i] = dst[((size_t)k * dst_rows + dst_row) * N + i];
    }
}

// many of the three above in one launch (rscm_device.hpp, WindowBatch): blockIdx.y picks the descriptor
__global__ __launch_bounds__(kBlock) void window_batch_kernel(const WindowBatch batch)
{
    const WindowOp& op = batch.ops[blockIdx.y];
    const int64_t N = op.N;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t x0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (op.kind == 0) {
        const int64_t per_var = (int64_t)op.keep * N, total = per_var * op.n_vars;
        for (int64_t x = x0; x < total; x += stride) {
            const int64_t v = x / per_var, o = x - v * per_var;
            double* var = op.buf + (size_t)v * op.R * N;
            var[o] = var[(size_t)op.shift * N + o];
        }
    } else if (op.kind == 1) {
        const int64_t total = (int64_t)op.n_out * N;
        for (int64_t x = x0; x < total; x += stride) {
            const int64_t k = x / N, i = x - k * N;
            const int64_t v = op.vars ? op.vars[k] - 1 : k;
            op.dst[((size_t)k * op.dst_rows + op.dst_row) * N + i] = op.buf[((size_t)v * op.R + op.src_row) * N + i];
        }
    } else {
        const int64_t per_var = (int64_t)(op.R - op.fill_from) * N, total = per_var * op.n_vars;
        const double nan = __builtin_nan("");
        for (int64_t x = x0; x < total; x += stride) {
            const int64_t v = x / per_var, o = x - v * per_var;
            op.buf[(size_t)v * op.R * N + (size_t)op.fill_from * N + o] = nan;
        }
    }
}

// ---- Gaussian log-likelihood ----------------------------------------------------------------
// per observation: residual = obs - model; chi = (residual*residual)/(sigma*sigma); l = -0.5*chi;
// normalised: l -= 0.5*ln(2*pi); l -= ln(sigma).  Per-variable partial sums, then their total
// (likelihood.rs:206-226, 238-248).  A non-finite model value is a member failure -> -inf
// (likelihood.rs:216-221, sampler/ensemble.rs:163-172).
__global__ __launch_bounds__(kBlock) void loglik_kernel(LoglikArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    const double ln_2pi = 1.8378770664093453;  // ln(2*pi) rounded to f64
    double total = 0.0, partial = 0.0;
    bool bad = false;
    for (int32_t j = 0; j < a.n_obs; ++j) {
        if (j > 0 && a.obs_group[j] != a.obs_group[j - 1]) {
            total += partial;
            partial = 0.0;
        }
        const double m = a.obs_series[j][i];
        if (!is_finite(m)) {
            bad = true;
            break;
        }
        const double sigma = a.obs_sigma[j];
        const double residual = a.obs_value[j] - m;
        const double chi = (residual * residual) / (sigma * sigma);
        double l = -0.5 * chi;
        if (a.normalize) {
            l -= 0.5 * ln_2pi;
            l -= log(sigma);
        }
        partial += l;
    }
    total += partial;
    a.out[i] = bad ? -__builtin_inf() : total;
}

// ---- summaries: count/sum/min/max over finite members ---------------------------------------
struct Stat4 {
    double cnt, sum, mn, mx;
};

__device__ __forceinline__ Stat4 stat_merge(Stat4 x, Stat4 y)
{
    return {x.cnt + y.cnt, x.sum + y.sum, fmin(x.mn, y.mn), fmax(x.mx, y.mx)};
}

__device__ __forceinline__ Stat4 wave_reduce(Stat4 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Stat4 o = {__shfl_down(v.cnt, off, 64), __shfl_down(v.sum, off, 64),
                   __shfl_down(v.mn, off, 64), __shfl_down(v.mx, off, 64)};
        v = stat_merge(v, o);
    }
    return v;
}

__device__ __forceinline__ Stat4 block_reduce(Stat4 v)
{
    __shared__ Stat4 wave_part[kBlock / 64];
    v = wave_reduce(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) v = stat_merge(v, wave_part[w]);
    }
    return v;  // valid in thread 0
}

__global__ __launch_bounds__(kBlock) void summary_partial_kernel(const double* row, int64_t n,
                                                                 double* partial)
{
    Stat4 v = {0.0, 0.0, __builtin_inf(), -__builtin_inf()};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double x = row[i];
        if (is_finite(x)) v = stat_merge(v, {1.0, x, x, x});
    }
    v = block_reduce(v);
    if (threadIdx.x == 0) {
        partial[4 * blockIdx.x + 0] = v.cnt;
        partial[4 * blockIdx.x + 1] = v.sum;
        partial[4 * blockIdx.x + 2] = v.mn;
        partial[4 * blockIdx.x + 3] = v.mx;
    }
}

__global__ __launch_bounds__(kBlock) void summary_final_kernel(const double* partial,
                                                               int32_t n_blocks, double* out)
{
    Stat4 v = {0.0, 0.0, __builtin_inf(), -__builtin_inf()};
    for (int32_t b = threadIdx.x; b < n_blocks; b += kBlock)
        v = stat_merge(v, {partial[4 * b], partial[4 * b + 1], partial[4 * b + 2], partial[4 * b + 3]});
    v = block_reduce(v);
    if (threadIdx.x == 0) {
        out[0] = v.cnt;
        out[1] = v.sum;
        out[2] = v.mn;
        out[3] = v.mx;
    }
}

// The same reduction for many time rows in one launch: blockIdx.y selects the row, the block
// layout along x is the one launch_summary uses, so every row gets the bits a single-row call gives.
__global__ __launch_bounds__(kBlock) void summary_rows_partial_kernel(const double* rows, int64_t n, double* partial)
{
    const double* row = rows + (size_t)blockIdx.y * n;
    Stat4 v = {0.0, 0.0, __builtin_inf(), -__builtin_inf()};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double x = row[i];
        if (is_finite(x)) v = stat_merge(v, {1.0, x, x, x});
    }
    v = block_reduce(v);
    if (threadIdx.x == 0) {
        double* p = partial + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        p[0] = v.cnt; p[1] = v.sum; p[2] = v.mn; p[3] = v.mx;
    }
}

__global__ __launch_bounds__(kBlock) void summary_rows_final_kernel(const double* partial, int32_t n_blocks, double* out)
{
    const double* part = partial + 4 * (size_t)blockIdx.x * n_blocks;
    Stat4 v = {0.0, 0.0, __builtin_inf(), -__builtin_inf()};
    for (int32_t b = threadIdx.x; b < n_blocks; b += kBlock)
        v = stat_merge(v, {part[4 * b], part[4 * b + 1], part[4 * b + 2], part[4 * b + 3]});
    v = block_reduce(v);
    if (threadIdx.x == 0) {
        double* o = out + 4 * (size_t)blockIdx.x;
        o[0] = v.cnt; o[1] = v.sum; o[2] = v.mn; o[3] = v.mx;
    }
}

// ---- counter-based Latin hypercube ----------------------------------------------------------
// Philox4x32-10 (philox.hpp) keyed by (seed, dimension), counter = global member id.
// Keyed bijection of [0, n): 4-round Feistel network on the next even-bit power-of-two domain,
// cycle-walked back into range (expected < 4 walks).  Replaces the serial Fisher-Yates shuffle
// of parameter_set.rs:224-226 with something every member can evaluate independently.
__device__ __forceinline__ uint64_t feistel_perm(uint64_t x, uint64_t n, uint32_t half_bits,
                                                 uint32_t k0, uint32_t k1)
{
    const uint64_t mask = (1ull << half_bits) - 1ull;
    do {
        uint64_t l = x >> half_bits, r = x & mask;
#pragma unroll
        for (uint32_t round = 0; round < 4; ++round) {
            uint32_t c[4] = {(uint32_t)r, (uint32_t)(r >> 32), round, 0x5EEDu};
            philox4x32_10(c, k0, k1);
            const uint64_t f = (((uint64_t)c[1] << 32) | c[0]) & mask;
            const uint64_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << half_bits) | r;
    } while (x >= n);
    return x;
}

__global__ __launch_bounds__(kBlock) void lhs_kernel(double* params, int32_t n_params,
                                                     int64_t n_local, uint64_t seed,
                                                     const double* low, const double* high,
                                                     int64_t member_offset, int64_t n_total,
                                                     uint32_t half_bits)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_local) return;
    const uint64_t g = (uint64_t)(member_offset + i);
    const double interval_size = 1.0 / (double)n_total;  // parameter_set.rs:216
    for (int32_t j = 0; j < n_params; ++j) {
        const uint32_t k0 = (uint32_t)seed ^ (0x9E3779B9u * (uint32_t)(j + 1));
        const uint32_t k1 = (uint32_t)(seed >> 32) + (uint32_t)j;
        const uint64_t stratum = feistel_perm(g, (uint64_t)n_total, half_bits, k0, k1);
        uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), 0xA5A5u, (uint32_t)j};
        philox4x32_10(c, k0, ~k1);
        const double u01 = u01_from_bits(c[0], c[1]);
        // interval_start + U*interval_size (parameter_set.rs:217-218), then the inverse CDF of a
        // bounded constant-pdf prior, low + u*(high-low) (:331-334)
        const double u = (double)stratum * interval_size + u01 * interval_size;
        params[(size_t)j * n_local + i] = low[j] + u * (high[j] - low[j]);
    }
}

__global__ __launch_bounds__(kBlock) void divtest_kernel(const double* num, const double* den,
                                                         double* out_ref, double* out_fast,
                                                         uint8_t* used_fast, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double a = num[i], b = den[i];
    out_ref[i] = a / b;
    const ConstDiv c = make_const_div(b);
    out_fast[i] = spec_div(a, c.d, c.r);  // the raw three-instruction quotient, no fallback
    used_fast[i] = const_div_fast_ok(a, c) ? 1 : 0;
}

inline unsigned grid_for(int64_t n, int64_t cap = 2048)
{
    int64_t b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace

static unsigned stream_grid(int64_t total)
{
    return (unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 8192);
}

hipError_t launch_slide_rows(double* buf, int64_t N, int32_t R, int32_t n_vars, int32_t shift, int32_t keep, hipStream_t s)
{
    if (n_vars <= 0 || keep <= 0 || shift <= 0) return hipSuccess;
    if (shift >= keep) {
        hipLaunchKernelGGL(slide_rows_kernel, dim3(stream_grid((int64_t)keep * N * n_vars)), dim3(kBlock), 0, s, buf, N, R, n_vars, shift, keep);
        return hipGetLastError();
    }
    // overlapping move: row by row in ascending order, each launch ordered after the one before
    for (int32_t r = 0; r < keep; ++r) {
        hipLaunchKernelGGL(slide_rows_kernel, dim3(stream_grid(N * n_vars)), dim3(kBlock), 0, s, buf + (size_t)r * N, N, R, n_vars, shift, 1);
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_fill_rows(double* buf, int64_t N, int32_t R, int32_t n_vars, int32_t row_begin, double value, hipStream_t s)
{
    if (n_vars <= 0 || row_begin >= R) return hipSuccess;
    hipLaunchKernelGGL(fill_rows_kernel, dim3(stream_grid((int64_t)(R - row_begin) * N * n_vars)), dim3(kBlock), 0, s, buf, N, R, n_vars,
                       row_begin, value);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const double* src, int64_t N, int32_t src_rows, int32_t src_row, const int32_t* vars, int32_t n_out,
                              double* dst, int32_t dst_rows, int32_t dst_row, hipStream_t s)
{
    if (n_out <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(stream_grid((int64_t)n_out * N)), dim3(kBlock), 0, s, src, N, src_rows, src_row, vars, n_out,
                       dst, dst_rows, dst_row);
    return hipGetLastError();
}

hipError_t launch_scatter_rows(double* src, int64_t N, int32_t src_rows, int32_t src_row, const int32_t* vars, int32_t n_out,
                               const double* dst, int32_t dst_rows, int32_t dst_row, hipStream_t s)
{
    if (n_out <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(stream_grid((int64_t)n_out * N)), dim3(kBlock), 0, s, src, N, src_rows, src_row, vars, n_out,
                       dst, dst_rows, dst_row);
    return hipGetLastError();
}

hipError_t launch_window_batch(const WindowBatch& batch, int32_t n_ops, hipStream_t s)
{
    if (n_ops <= 0) return hipSuccess;
    if (n_ops > kMaxWindowOps) return hipErrorInvalidValue;
    int64_t most = 1;
    for (int32_t k = 0; k < n_ops; ++k) {
        const WindowOp& op = batch.ops[k];
        const int64_t total = op.kind == 0 ? (int64_t)op.keep * op.N * op.n_vars
                            : op.kind == 1 ? (int64_t)op.n_out * op.N : (int64_t)(op.R - op.fill_from) * op.N * op.n_vars;
        most = std::max(most, total);
    }
    const unsigned gx = (unsigned)std::min<int64_t>((most + kBlock - 1) / kBlock, 2048);
    hipLaunchKernelGGL(window_batch_kernel, dim3(gx, (unsigned)n_ops), dim3(kBlock), 0, s, batch);
    return hipGetLastError();
}

hipError_t launch_fill(double* p, int64_t n, double v, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, p, n, v);
    return hipGetLastError();
}

hipError_t launch_broadcast_row(double* row, int64_t n, const double* src, int64_t n_src,
                                hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(broadcast_row_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, row, n, src, n_src);
    return hipGetLastError();
}

hipError_t launch_loglik(const LoglikArgs& a, hipStream_t s)
{
    if (a.n_members <= 0) return hipSuccess;
    hipLaunchKernelGGL(loglik_kernel, dim3((unsigned)((a.n_members + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

int32_t summary_blocks(int64_t n) { return (int32_t)grid_for(n, 1024); }

hipError_t launch_summary(const double* row, int64_t n, double* partial, int32_t n_blocks,
                          double* out, hipStream_t s)
{
    hipLaunchKernelGGL(summary_partial_kernel, dim3(n_blocks), dim3(kBlock), 0, s, row, n, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(summary_final_kernel, dim3(1), dim3(kBlock), 0, s, partial, n_blocks, out);
    return hipGetLastError();
}

// rows: n_rows consecutive [n] rows; partial: [n_rows][n_blocks][4]; out: [n_rows][4]
hipError_t launch_summary_rows(const double* rows, int64_t n, int32_t n_rows, double* partial, int32_t n_blocks,
                               double* out, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(summary_rows_partial_kernel, dim3(n_blocks, n_rows), dim3(kBlock), 0, s, rows, n, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(summary_rows_final_kernel, dim3(n_rows), dim3(kBlock), 0, s, partial, n_blocks, out);
    return hipGetLastError();
}

hipError_t launch_lhs(double* params, int32_t n_params, int64_t n_local, uint64_t seed,
                      const double* low, const double* high, int64_t member_offset,
                      int64_t n_total, hipStream_t s)
{
    if (n_local <= 0) return hipSuccess;
    uint32_t bits = 2;
    while ((1ull << bits) < (uint64_t)n_total) ++bits;
    if (bits & 1u) ++bits;  // balanced Feistel halves
    hipLaunchKernelGGL(lhs_kernel, dim3((unsigned)((n_local + kBlock - 1) / kBlock)), dim3(kBlock),
                       0, s, params, n_params, n_local, seed, low, high, member_offset, n_total,
                       bits / 2);
    return hipGetLastError();
}

hipError_t launch_divtest(const double* num, const double* den, double* out_ref, double* out_fast,
                          uint8_t* used_fast, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(divtest_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       s, num, den, out_ref, out_fast, used_fast, n);
    return hipGetLastError();
}

}  // namespace rscm
