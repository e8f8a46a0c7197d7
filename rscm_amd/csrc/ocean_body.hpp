// OceanCarbon per-member arithmetic (csrc/ocean.hip launches it): parameters/ocean_carbon.rs:198-250, carbon/ocean.rs:73-215.
#pragma once

#include <type_traits>

#include "rscm_device.hpp"

namespace rscm {
namespace ocean {

constexpr double kPpmToGtc = 2.124;                 // carbon/ocean.rs:26
constexpr double kMicromolPerPpmM3PerKg = 1.72e17;  // parameters/ocean_carbon.rs:4

struct OceanMember {
    double pco2_pi, k_gas, temp_sens, dic_conv, coef[5];
    bool temp_on;
};

// delta_pco2_from_dic + ocean_pco2 (parameters/ocean_carbon.rs:218-245); powi(k) as LLVM expands it
__device__ __forceinline__ double pco2_from_dic(const OceanMember& m, double d, double temp_factor)
{
    const double d2 = d * d, d3 = d * d2, d4 = d2 * d2, d5 = d * d4;
    const double g[5] = {d, d2 * 1e-3, -d3 * 1e-5, d4 * 1e-7, -d5 * 1e-10};
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 5; ++q) s += m.coef[q] * g[q];
    return (m.pco2_pi + s) * temp_factor;
}

// The flux history is a ring of a.hist_rows pulses (>= max_history_months + the pulses of one tile; the
// whole run's pulses when that is shorter): pulse j lives in row j mod hist_rows.  All pulse indices are
// wave-uniform, so the ring arithmetic stays on the scalar unit: one modulo per tile, then increments.
__device__ __forceinline__ int32_t ring_add(int32_t r, int32_t k, int32_t R)
{
    const int32_t x = r + k;
    return x >= R ? x - R : x;
}

// The O(T) recurrence of RSCM_MODE_FAST (csrc/ocean.hip has the derivation) over model steps [step_begin, step_end)
// for member i: the running mode sums are loaded at the start (or re-formed from the flux history: `rebuild`) and
// stored at the end, the last NEAR pulses come out of the history ring -- so a call per model step (the one-step launches
// of a graph) carries the same bits as one call over many steps.
template <int NEAR, int SRC>
__device__ __forceinline__ void ocean_recur_run(const OceanArgs& a, const double* __restrict__ irf_table, const double* __restrict__ mode_table,
                                                int64_t i, int32_t step_begin, int32_t step_end, bool rebuild)
{
    constexpr int STEPS = 12, M = kOceanModes;
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    OceanMember m;
    m.pco2_pi = P(2);
    m.k_gas = P(3) / (P(4) * 12.0);
    m.temp_sens = P(5);
    m.dic_conv = kMicromolPerPpmM3PerKg / (P(7) * P(8));
    const double sst_pi = P(9);
#pragma unroll
    for (int q = 0; q < 5; ++q) m.coef[q] = P(13 + q) + P(18 + q) * sst_pi;
    m.temp_on = P(23) != 0.0;
    const MemberInputsEager<SRC, 2> in(a.inputs, a.scen, a.links, T, N, i);
    const double* __restrict__ irf = irf_table;
    double* __restrict__ hist = a.hist + i;
    a.status[i] = 0;
    const int64_t H = a.max_hist;
    const size_t vs = (size_t)a.rows * N;
    // ---- state at the start of the launch
    int64_t m0 = (int64_t)step_begin * STEPS;
    double S[M];
    if (rebuild) {
        // S_q(m0 - 1) = sum over the pulses j with lag m0 - 1 - j in [NEAR, H): Horner in d_q, oldest pulse first
#pragma unroll
        for (int q = 0; q < M; ++q) S[q] = 0.0;
        const int64_t j_lo = m0 - H > 0 ? m0 - H : 0;
        int32_t jr = (int32_t)(j_lo % a.hist_rows);
        for (int64_t j = j_lo; j <= m0 - 1 - NEAR; ++j, jr = ring_add(jr, 1, a.hist_rows)) {
            const double f = hist[(size_t)jr * N];
#pragma unroll
            for (int q = 0; q < M; ++q) S[q] = __builtin_fma(S[q], mode_table[q], f);
        }
    } else {
#pragma unroll
        for (int q = 0; q < M; ++q) S[q] = a.mode_state[(size_t)q * N + i];
    }
    double w[NEAR + STEPS];  // w[x] = f(m0 - NEAR + x): the last NEAR pulses, then this step's
    {   // (one modulo for the first ring row, then increments: the indices are wave-uniform, but a 64-bit modulo per
        // pulse is ~60 scalar instructions, and a one-step launch pays this for every model step)
        const int64_t j0 = m0 - NEAR;
        int32_t jr = (int32_t)((j0 > 0 ? j0 : 0) % a.hist_rows);
#pragma unroll
        for (int x = 0; x < NEAR; ++x) {
            const int64_t j = j0 + x;
            if (j >= 0) {
                w[x] = hist[(size_t)jr * N];
                jr = ring_add(jr, 1, a.hist_rows);
            } else {
                w[x] = 0.0;
            }
        }
    }
    const size_t r0 = (size_t)step_begin * N + i;
    double pco2 = a.series[r0], cumulative = a.series[vs + r0];
    const int32_t R = a.hist_rows;
    int32_t mr = (int32_t)(m0 % R);                                  // ring row of this sub-step's pulse
    int32_t mr_out = m0 >= H ? (int32_t)((m0 - H) % R) : 0;          // ... of the pulse that leaves the window (used once m >= H)
    double* __restrict__ out = a.series + r0;
    for (int32_t n = step_begin; n < step_end; ++n, m0 += STEPS) {
        const double co2 = in.at(0, n), delta_sst = in.at(1, n);
        const double dt = a.bounds[n + 1] - a.bounds[n];
        const double dt_month = dt / (double)STEPS;
        const double temp_factor = m.temp_on ? exp(m.temp_sens * delta_sst) : 1.0;
        const bool leaving = m0 >= H;   // sub-steps of a step that straddles m = H take the general path below
        const bool straddle = !leaving && m0 + STEPS > H;
        // The response values and mode constants of a step are wave-uniform and read with scalar loads.  They do not
        // fit the 102 SGPRs of a wave at once: an opaque zero offset per step keeps the compiler from hoisting them out
        // of the step loop (and then spilling them into VGPR lanes, one v_readlane pair per use).
        int32_t opaque = 0;
        asm volatile("" : "+s"(opaque));
        const double* __restrict__ rt = irf + opaque;
        const double* __restrict__ md = mode_table + opaque;           // d_q
        const double* __restrict__ mc = mode_table + M + opaque;       // c_q
        const double* __restrict__ me = mode_table + 2 * M + opaque;   // e_q
        // The twelve convolutions of the step, sub-step s: integral_s = sum_q c_q S_q(s) (fastest-decaying mode first)
        // + sum over the lags NEAR-1 .. 0 of f(m0 + s - lag) r(lag) (oldest pulse first).  Everything that does not
        // involve this step's own pulses -- the mode sums (their inputs are the pulses that are NEAR months old and the
        // ones leaving the window) and the lags >= 12 -- is formed for all twelve sub-steps at once, mode by mode and
        // lag by lag: every table value is loaded once per step instead of once per sub-step, and the twelve sums are
        // twelve independent chains.  Each sum still receives its terms in the order written above, so the bits are
        // those of the sub-step-by-sub-step form.
        double f_out[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            f_out[s] = 0.0;
            if (leaving || (straddle && m0 + s >= H)) {  // wave-uniform
                f_out[s] = hist[(size_t)mr_out * N];
                mr_out = ring_add(mr_out, 1, R);
            }
        }
        double acc[STEPS];
        // (one uniform branch per step on the window's state, not one per mode and sub-step)
        auto mode_sums = [&](auto exit_kind) {
            constexpr int EXIT = decltype(exit_kind)::value;   // 0: no pulse leaves in this step, 1: one per sub-step, 2: from some sub-step on
#pragma unroll
            for (int q = M - 1; q >= 0; --q) {
                const double d = md[q], c = mc[q], e = me[q];
                double Sq = S[q];
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    Sq = __builtin_fma(Sq, d, w[s]);   // the pulse that is NEAR months old enters
                    if (EXIT == 1 || (EXIT == 2 && m0 + s >= H)) Sq = __builtin_fma(-e, f_out[s], Sq);   // the one H months old leaves (e_q = 0 beyond n_exit)
                    acc[s] = __builtin_fma(c, Sq, q == M - 1 ? 0.0 : acc[s]);
                }
                S[q] = Sq;
            }
        };
        if (leaving) mode_sums(std::integral_constant<int, 1>());
        else if (straddle) mode_sums(std::integral_constant<int, 2>());
        else mode_sums(std::integral_constant<int, 0>());
#pragma unroll
        for (int lag = NEAR - 1; lag >= STEPS; --lag) {
            const double r = rt[lag];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) acc[s] = __builtin_fma(w[NEAR + s - lag], r, acc[s]);
        }
        double total = 0.0;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const double flux_ppm = m.k_gas * (co2 - pco2);
            w[NEAR + s] = flux_ppm;
            hist[(size_t)mr * N] = flux_ppm;
            mr = ring_add(mr, 1, R);
            const double flux_gtc_yr = flux_ppm * 12.0 * kPpmToGtc;
            total += flux_gtc_yr / (double)STEPS;
            cumulative += flux_gtc_yr * dt_month;
            double integral = acc[s];
#pragma unroll
            for (int lag = STEPS - 1; lag >= 0; --lag) integral = __builtin_fma(w[NEAR + s - lag], rt[lag], integral);
            const double delta_dic = integral * m.dic_conv;
            pco2 = pco2_from_dic(m, delta_dic, temp_factor);
        }
#pragma unroll
        for (int x = 0; x < NEAR; ++x) w[x] = w[x + STEPS];
        out += N;
        out[0] = pco2;
        out[vs] = cumulative;
        out[2 * vs] = total;
    }
#pragma unroll
    for (int q = 0; q < M; ++q) a.mode_state[(size_t)q * N + i] = S[q];
}


}  // namespace ocean
}  // namespace rscm
