// OceanCarbon ensemble kernel for gfx950 (MI355X), one thread per member.
//
// What it replaces, per model step n (reference file:line):
//   OceanCarbon::solve_impl / solve_ocean / calculate_delta_dic / calculate_flux
//                                    crates/rscm-magicc/src/carbon/ocean.rs:73-215
//   OceanCarbonParameters::{irf, scale_irf, delta_pco2_from_dic, ocean_pco2, ...}
//                                    crates/rscm-magicc/src/parameters/ocean_carbon.rs:198-250
// under the stepper conventions of crates/rscm-core/src/model/runtime.rs: CO2 and the SST anomaly
// are exogenous series shared per scenario (index n), pCO2 and the cumulative uptake are the
// component's own state (index n), the flux history is its internal state; outputs at n+1.
//
// The reference convolves, at every monthly sub-step, the whole flux history (up to
// max_history_months = 6000 pulses) with the mixed-layer impulse response, evaluating the response
// function for every pair.  Here:
//   * the scaled response depends only on the lag and on uniform parameters: the host tabulates it
//     once (rscm_gpu.cpp, ocean_irf_table -- same expressions, so the same bits as per-call
//     evaluation), and the kernel reads it with wave-uniform indices (scalar loads);
//   * the flux history lives in HBM as hist[month][N], member fastest;
//   * the convolutions of two consecutive years share their old pulses: each old pulse is loaded
//     ONCE per two years and multiplied into 24 running sums (one per sub-step), so two years cost
//     one pass over the history instead of 24.  Every sum still adds its terms oldest-to-newest, one
//     rounded multiply and one rounded add per term like the reference (the build uses
//     -ffp-contract=off), so the sums carry the same bits;
//   * this year's own pulses stay in registers.
// Per member-year: half a coalesced pass over the history (4 B per pulse) and 24 f64 operations
// per pulse: with the full 6000-month window 24 KB and 144 k operations (DESIGN.md has the
// roofline).  exp() of the temperature factor comes from the device
// math library: results agree with the CPU oracle to its last-place error, and bit for bit when
// the temperature feedback is off (tests/test_gpu_ocean.py).
#include "ocean_body.hpp"

namespace rscm {

namespace {

using namespace ocean;

// acc + f*r: one rounded multiply and one rounded add like the reference (EXACT), or fused
// (RSCM_MODE_FAST: half the instructions, results differ by rounding only)
template <bool FUSED>
__device__ __forceinline__ double mac(double acc, double f, double r)
{
    return FUSED ? __builtin_fma(f, r, acc) : acc + f * r;
}

// YEARS consecutive model steps starting at n: one pass over the old pulses feeds the
// STEPS*YEARS running sums of all their sub-steps.
// PART splits a tile over YEARS launches, for callers that advance one step at a time (linked
// graphs: next year's CO2 depends on this year's flux): part 0 = the pass over the old pulses and
// the first year, the later years' running sums parked in a.partial; part p > 0 = year p, resumed
// from its parked sums and the tile's earlier pulses.  The same sums in the same order as a whole
// tile (PART -1), a YEARS-th of the history traffic of one-year tiles.
template <int STEPS, int YEARS, bool FUSED, int PART, class Inputs>
__device__ __forceinline__ void ocean_tile(const OceanArgs& a, const OceanMember& m, const Inputs& in,
                                           const double* __restrict__ irf, double* __restrict__ hist, int64_t i, int32_t n)
{
    constexpr int K = STEPS * YEARS;
    const int64_t N = a.n_members;
    const int64_t H = a.max_hist;
    const size_t vs = (size_t)a.rows * N;
    const int64_t m0 = (int64_t)n * STEPS;  // months already in the history
    // sub-step k of the tile convolves the pulses j in [lo(k), m0 + k]
    auto lo = [&](int k) -> int64_t { const int64_t v = m0 + k + 1 - H; return v > 0 ? v : 0; };
    double A[K];
#pragma unroll
    for (int k = 0; k < K; ++k) A[k] = 0.0;
    // ---- the old pulses, oldest first.  Head: the first K-1 of them are still outside the
    // window of the later sub-steps (bounded history), so each term is predicated.
    const int32_t R = a.hist_rows;
    int64_t j = lo(0);
    if constexpr (PART > 0) {
        j = m0;  // the old pulses were summed by the launch of part 0
#pragma unroll
        for (int k = PART * STEPS; k < (PART + 1) * STEPS; ++k) A[k] = a.partial[(size_t)(k - STEPS) * N + i];
    }
    const int32_t mr0 = (int32_t)(m0 % R);     // ring row of the tile's first own pulse
    const int64_t head_end = (j + K - 1 < m0) ? j + K - 1 : m0;
    // The old pulses [j, m0) occupy at most two runs of consecutive ring rows (the ring wraps at most once
    // inside a window): within a run the loops below walk a plain pointer, as they did over the unwrapped history.
    const int32_t jr0 = (int32_t)(j % R);
    const int64_t j_wrap = j + (R - jr0);   // the first pulse at or after j that sits in ring row 0
    int64_t jb = j;
    int32_t row = jr0;
#pragma unroll 1
    for (int seg = 0; seg < 2 && jb < m0; ++seg) {   // one copy of the loops serves both runs
        const int64_t je = (seg == 0 && j_wrap < m0) ? j_wrap : m0;
        const double* __restrict__ hp = hist + (size_t)row * N;
        int64_t jj = jb;
        // Head: the first K-1 old pulses are still outside the window of the later sub-steps (bounded
        // history), so each term is predicated.
        for (; jj < je && jj < head_end; ++jj, hp += N) {
            const double f = *hp;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int64_t lag = m0 + k - jj;
                const double next = mac<FUSED>(A[k], f, irf[lag < H ? lag : 0]);
                if (jj >= lo(k)) A[k] = next;
            }
        }
        // Bulk: groups of STEPS pulses share a K+STEPS-1 entry window of the response table
        for (; jj + STEPS <= je; jj += STEPS, hp += (size_t)STEPS * N) {
            const int64_t base = m0 - jj - (STEPS - 1);  // lag of (last pulse of the group, k = 0), >= 1
            double w[K + STEPS - 1], f[STEPS];
#pragma unroll
            for (int t = 0; t < K + STEPS - 1; ++t) w[t] = irf[base + t];
#pragma unroll
            for (int u = 0; u < STEPS; ++u) f[u] = hp[(size_t)u * N];
#pragma unroll
            for (int u = 0; u < STEPS; ++u) {
#pragma unroll
                for (int k = 0; k < K; ++k) A[k] = mac<FUSED>(A[k], f[u], w[STEPS - 1 - u + k]);
            }
        }
        for (; jj < je; ++jj, hp += N) {  // tail
            const double f = *hp;
#pragma unroll
            for (int k = 0; k < K; ++k) A[k] = mac<FUSED>(A[k], f, irf[m0 + k - jj]);
        }
        jb = je;
        row = 0;
    }
    // ---- the tile's own sub-steps (solve_ocean, carbon/ocean.rs:116-160), its pulses in registers
    double fy[K];
    const size_t r0 = (size_t)n * N + i;
    constexpr int Y0 = PART > 0 ? PART : 0, Y1 = PART >= 0 ? PART + 1 : YEARS;
    double pco2 = a.series[r0 + (size_t)Y0 * N], cumulative = a.series[vs + r0 + (size_t)Y0 * N];
    if constexpr (PART > 0) {
#pragma unroll
        for (int q = 0; q < PART * STEPS; ++q) fy[q] = hist[(size_t)ring_add(mr0, q, R) * N];  // the tile's earlier pulses
    }
#pragma unroll
    for (int y = Y0; y < Y1; ++y) {
        const double co2 = in.at(0, n + y), delta_sst = in.at(1, n + y);
        const double dt = a.bounds[n + y + 1] - a.bounds[n + y];
        const double dt_month = dt / (double)STEPS;
        const double temp_factor = m.temp_on ? exp(m.temp_sens * delta_sst) : 1.0;
        double total = 0.0;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int k = y * STEPS + s;
            const double flux_ppm = m.k_gas * (co2 - pco2);
            fy[k] = flux_ppm;
            hist[(size_t)ring_add(mr0, k, R) * N] = flux_ppm;
            const double flux_gtc_yr = flux_ppm * 12.0 * kPpmToGtc;
            total += flux_gtc_yr / (double)STEPS;
            cumulative += flux_gtc_yr * dt_month;
            double integral = A[k];
#pragma unroll
            for (int q = 0; q <= k; ++q)
                if (m0 + q >= lo(k)) integral = mac<FUSED>(integral, fy[q], irf[k - q < H ? k - q : 0]);
            const double delta_dic = H > 0 ? integral * m.dic_conv : 0.0;
            pco2 = pco2_from_dic(m, delta_dic, temp_factor);
        }
        const size_t r1 = r0 + (size_t)(y + 1) * N;
        a.series[r1] = pco2;
        a.series[vs + r1] = cumulative;
        a.series[2 * vs + r1] = total;
    }
    if constexpr (PART == 0 && YEARS > 1) {
#pragma unroll
        for (int k = STEPS; k < K; ++k) a.partial[(size_t)(k - STEPS) * N + i] = A[k];
    }
}

template <int STEPS, int SRC, bool FUSED>
__global__ __launch_bounds__(kBlock) void ocean_kernel(OceanArgs a, const double* __restrict__ irf_table)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    OceanMember m;
    m.pco2_pi = P(2);
    m.k_gas = P(3) / (P(4) * 12.0);  // gas_exchange_rate()
    m.temp_sens = P(5);
    m.dic_conv = kMicromolPerPpmM3PerKg / (P(7) * P(8));  // dic_conversion_factor()
    const double sst_pi = P(9);
#pragma unroll
    for (int q = 0; q < 5; ++q) m.coef[q] = P(13 + q) + P(18 + q) * sst_pi;
    m.temp_on = P(23) != 0.0;
    const MemberInputsEager<SRC, 2> in(a.inputs, a.scen, a.links, T, N, i);
    const double* __restrict__ irf = irf_table;  // [H], wave-uniform indices: a read-only kernel argument -> scalar loads
    double* __restrict__ hist = a.hist + i;   // [months][N]
    a.status[i] = 0;
    int32_t n = a.step_begin;
    // Several model steps per pass over the history: a fraction of the HBM traffic, the same sums in
    // the same order.  Four-year tiles pay off where the multiply-adds are fused (FAST: 570 -> 450 ms
    // at 262144 members) and cost more than they save where each is two instructions (EXACT: 887 ms,
    // 48 running sums and their response window crowd the registers); three-year tiles are the
    // best EXACT gets (750 -> 720 ms).
    constexpr int YS = kOceanTileYears<FUSED>;
    if (a.part >= 0) {
        // one step of a tile split over kOceanSplitYears launches (rscm_gpu.cpp groups the steps).
        // Two years: split tiles of three or four measured no faster in a lock-step graph (the
        // launch of part 0 is issue-bound, not traffic-bound: 343 vs 352 us per step, EXACT).
        static_assert(kOceanSplitYears == 2, "the dispatch below lists the parts of a two-year tile");
        if (a.part == 0) ocean_tile<STEPS, kOceanSplitYears, FUSED, 0>(a, m, in, irf, hist, i, n);
        else ocean_tile<STEPS, kOceanSplitYears, FUSED, 1>(a, m, in, irf, hist, i, n - 1);
        return;
    }
    for (; n + YS <= a.step_end; n += YS) ocean_tile<STEPS, YS, FUSED, -1>(a, m, in, irf, hist, i, n);
    for (; n + 2 <= a.step_end; n += 2) ocean_tile<STEPS, 2, FUSED, -1>(a, m, in, irf, hist, i, n);
    if (n < a.step_end) ocean_tile<STEPS, 1, FUSED, -1>(a, m, in, irf, hist, i, n);
}

// ---------------------------------------------------------------------------------------------
// RSCM_MODE_FAST: O(T) instead of O(T^2).  The convolution of sub-step m splits at lag NEAR:
//   lags 0 .. NEAR-1   explicit, with the tabulated response (whatever its form: the first-year
//                      polynomial of 3D-GFDL, the early exponential sums of 2D-BERN / HILDA), the last
//                      NEAR pulses kept in registers;
//   lags NEAR .. H-1   through the host-fitted decaying modes (OceanModes), one running sum S_q per
//                      mode and member: S_q <- d_q S_q + f(m - NEAR) - e_q f(m - H).
// The scaled response is s raw / (s raw + 1 - raw) with raw a sum of six exponentials: not itself a
// finite exponential sum, but its expansion in powers of (1 - s) raw converges fast; rates up to
// pairwise sums, amplitudes by least squares over the whole window, reproduce the table to a few 1e-12
// (checked by the host at configuration, which otherwise keeps the tiled kernel above).  The results
// differ from the EXACT mode by that fit and by the summation order: tests/test_gpu_ocean.py states
// the tolerance.  Per member-year: 12 x (NEAR + 2 x 21 + exits + ~25) f64 operations and 216 B of HBM
// traffic (12 pulses written, 12 leaving pulses read, 3 output rows) against 144 k operations and 24 KB.
// One wavefront per SIMD: the 72 pulses, 21 mode sums, 12 partial convolutions and 12 leaving pulses of a step are
// ~250 registers; cut to 256 for two wavefronts per SIMD the compiler spills 122 of them (17.3 ms against 16.2 ms at
// 262 144 members x 750 years), and the twelve independent sums of a step hide their own latencies.  The same cut for the
// one-step launches of a lock-step graph alone (125 000 members: 1954 wavefronts on 1024 SIMDs, three quarters of a
// wavefront's time spent waiting on memory): 34-38 us per launch either way, configs[3]'s share 2.65 s against 2.64 s.
template <int NEAR, int SRC>
__global__ __launch_bounds__(kBlock, 1) void ocean_recur_kernel(OceanArgs a, const double* __restrict__ irf_table,
                                                             const double* __restrict__ mode_table)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    ocean_recur_run<NEAR, SRC>(a, irf_table, mode_table, i, a.step_begin, a.step_end, a.rebuild != 0);
}

template <int NEAR>
static hipError_t launch_recur(const OceanArgs& a, hipStream_t s)
{
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    const int src = a.linked ? 2 : a.scen ? 1 : 0;
    void (*kern)(OceanArgs, const double*, const double*) =
        src == 2 ? ocean_recur_kernel<NEAR, 2> : src == 1 ? ocean_recur_kernel<NEAR, 1> : ocean_recur_kernel<NEAR, 0>;
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, s, a, a.irf, a.mode_table);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_ocean(const OceanArgs& a, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    if (a.steps != 12) return hipErrorInvalidValue;  // the sub-step loop is unrolled for monthly steps
    if (a.recur) {
        if (a.near == 60) return launch_recur<60>(a, s);
        if (a.near == 120) return launch_recur<120>(a, s);
        return hipErrorInvalidValue;
    }
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    const int src = a.linked ? 2 : a.scen ? 1 : 0;
    void (*kern)(OceanArgs, const double*) =
        a.fused ? (src == 2 ? ocean_kernel<12, 2, true> : src == 1 ? ocean_kernel<12, 1, true> : ocean_kernel<12, 0, true>)
                : (src == 2 ? ocean_kernel<12, 2, false> : src == 1 ? ocean_kernel<12, 1, false> : ocean_kernel<12, 0, false>);
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, s, a, a.irf);
    return hipGetLastError();
}

}  // namespace rscm
