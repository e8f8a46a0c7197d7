// Device code of the two-layer stepping kernel, shared by two_layer.hip (whole-axis launches) and
// group.hip (several linked components of one model step in one launch).  See two_layer.hip for what it
// replaces in the reference and why it is written this way.
#pragma once

#include "rk4_device.hpp"
#include "rscm_device.hpp"

namespace rscm {
namespace tl {

struct TLConst {
    double lambda0, a, eff_eta, eta, cs, cd, rcs, rcd;
};

// EXACT arithmetic, the reference's expression order, every product and sum rounded separately:
//   temperature_difference = ts - td
//   lambda_eff = lambda0 - a*ts
//   heat_exchange_surface = efficacy*eta*temperature_difference      ((efficacy*eta) first)
//   dts = (erf - lambda_eff*ts - heat_exchange_surface) / heat_capacity_surface
//   dtd = (eta*temperature_difference) / heat_capacity_deep
template <bool SPEC>
__device__ __forceinline__ void rhs_exact(const TLConst& p, double erf, double ts, double td,
                                          double& dts, double& dtd, int32_t& acc)
{
    const double diff = ts - td;
    const double lambda_eff = p.lambda0 - p.a * ts;
    const double hx_s = p.eff_eta * diff;
    const double num_s = erf - lambda_eff * ts - hx_s;
    const double num_d = p.eta * diff;
    if constexpr (SPEC) {
        dts = spec_div(num_s, p.cs, p.rcs);
        dtd = spec_div(num_d, p.cd, p.rcd);
        acc = max3_i32(acc, window_tag(num_s), window_tag(num_d));
    } else {
        dts = num_s / p.cs;
        dtd = num_d / p.cd;
    }
}

template <bool SPEC>
__device__ __forceinline__ void rk4_step_exact(const TLConst& p, double erf, double h,
                                               double half_step, double sixth, double& ts,
                                               double& td, int32_t& acc)
{
    double k1s, k1d, k2s, k2d, k3s, k3d, k4s, k4d;
    rhs_exact<SPEC>(p, erf, ts, td, k1s, k1d, acc);
    rhs_exact<SPEC>(p, erf, ts + k1s * half_step, td + k1d * half_step, k2s, k2d, acc);
    rhs_exact<SPEC>(p, erf, ts + k2s * half_step, td + k2d * half_step, k3s, k3d, acc);
    rhs_exact<SPEC>(p, erf, ts + k3s * h, td + k3d * h, k4s, k4d, acc);
    if constexpr (SPEC) {
        // (k1 + k2*2) + k3*2: the doubling is exact, so the fused form rounds identically
        ts = rk4_combine_fused2(ts, k1s, k2s, k3s, k4s, sixth);
        td = rk4_combine_fused2(td, k1d, k2d, k3d, k4d, sixth);
    } else {
        ts = rk4_combine(ts, k1s, k2s, k3s, k4s, sixth);
        td = rk4_combine(td, k1d, k2d, k3d, k4d, sixth);
    }
}

struct TLFast {
    double l0, a, ee, ed;  // lambda0/Cs, a/Cs, efficacy*eta/Cs, eta/Cd
};

// FAST: the same algebra with the heat capacities folded into the coefficients, FMAs, and the temperature DIFFERENCE
// w = Ts - Td as the second unknown instead of Td:
//     Ts' = F/Cs - (l0 - a Ts) Ts - ee w          w' = Ts' - ed w
// Four fused operations per stage instead of five (the difference need not be formed), 30 per RK4 step instead of 34; the
// deep-ocean temperature is Ts - w at the end of the model step.  Within 1e-11 of the oracle on bounded members
// (tests/test_gpu_parity.py; measured 1e-14).
__device__ __forceinline__ void rhs_fast(const TLFast& p, double erf_cs, double ts, double w, double& dts, double& dw)
{
    const double q = __builtin_fma(p.a, ts, -p.l0);   // -(lambda0 - a Ts)/Cs
    const double t = __builtin_fma(q, ts, erf_cs);
    dts = __builtin_fma(-p.ee, w, t);
    dw = __builtin_fma(-p.ed, w, dts);
}

// The RK4 sub-steps of one model step: stages as FMAs, combination y + h/6*(k1+k4) + h/3*(k2+k3).  Shared by the two-layer kind
// and the FAST coupled chain (coupled.hip), so the chain assembled from linked components carries the fused kernel's bits in this
// mode too.
__device__ __forceinline__ void rk4_year_fast(const TLFast& p, double erf_cs, int32_t m, double h, double half_step, double third,
                                              double sixth, double& ts, double& td)
{
    double w = ts - td;
    for (int32_t s = 0; s < m; ++s) {
        double k1s, k1w, k2s, k2w, k3s, k3w, k4s, k4w;
        rhs_fast(p, erf_cs, ts, w, k1s, k1w);
        rhs_fast(p, erf_cs, __builtin_fma(k1s, half_step, ts), __builtin_fma(k1w, half_step, w), k2s, k2w);
        rhs_fast(p, erf_cs, __builtin_fma(k2s, half_step, ts), __builtin_fma(k2w, half_step, w), k3s, k3w);
        rhs_fast(p, erf_cs, __builtin_fma(k3s, h, ts), __builtin_fma(k3w, h, w), k4s, k4w);
        ts = __builtin_fma(k2s + k3s, third, __builtin_fma(k1s + k4s, sixth, ts));
        w = __builtin_fma(k2w + k3w, third, __builtin_fma(k1w + k4w, sixth, w));
    }
    td = ts - w;
}

// the folded coefficients of a member, formed the same way wherever FAST two-layer arithmetic runs
__device__ __forceinline__ TLFast make_fast(double lambda0, double a, double efficacy, double eta, double cs, double cd, double& inv_cs)
{
    inv_cs = 1.0 / cs;
    TLFast p;
    p.l0 = lambda0 * inv_cs;
    p.a = a * inv_cs;
    p.ee = efficacy * eta * inv_cs;
    p.ed = eta / cd;
    return p;
}

// Per-member Gaussian log-likelihood accumulated while stepping (STORE == false): same
// expression and summation order as loglik_kernel in ensemble_ops.hip -- per-variable partial
// sums in time order, then the total in the caller's group order (likelihood.rs:186-250).
struct LikAcc {
    double part_s = 0.0, part_d = 0.0;
    bool bad = false;
    int32_t oi = 0;
};

__device__ __forceinline__ void lik_consume(const TwoLayerArgs& a, LikAcc& L, int32_t row, double ts,
                                            double td)
{
    while (L.oi < a.n_obs && a.obs_tidx[L.oi] == row) {  // wave-uniform
        const bool deep = a.obs_is_deep[L.oi] != 0;
        const double m = deep ? td : ts;
        if (!is_finite(m)) L.bad = true;
        const double sigma = a.obs_sigma[L.oi];
        const double residual = a.obs_value[L.oi] - m;
        const double chi = (residual * residual) / (sigma * sigma);
        double l = -0.5 * chi;
        if (a.normalize) {
            l -= 0.5 * 1.8378770664093453;  // ln(2*pi)
            l -= log(sigma);
        }
        if (deep) L.part_d += l;
        else L.part_s += l;
        ++L.oi;
    }
}

// Member i over the steps [a.step_begin, a.step_end).  LDS: the forcing slice staged by the caller in
// lds_forcing ([n_scen][len]); otherwise read through L2 (table or linked series).
// Cache: NoCache for the stand-alone kernels; the fused multi-step launch (group.hip) keeps parameters, the
// state and the linked forcing of the current step in LDS between its steps (rscm_device.hpp, LdsCache).
template <int MODE, bool LDS, bool STORE, class Cache = NoCache>
__device__ __forceinline__ void two_layer_body(const TwoLayerArgs& a, const double* lds_forcing, int64_t i, int32_t step_begin,
                                               int32_t step_end, const Cache& cache = Cache())
{
    const int32_t len = step_end - step_begin;
    const int64_t N = a.row_stride;   // the rows' stride (the caller has checked i against a.n_members)

    const double lambda0 = cache.param(a.params, a.uniform_rows, 0, N, i);
    const double pa = cache.param(a.params, a.uniform_rows, 1, N, i);
    const double efficacy = cache.param(a.params, a.uniform_rows, 2, N, i);
    const double eta = cache.param(a.params, a.uniform_rows, 3, N, i);
    const double cs = cache.param(a.params, a.uniform_rows, 4, N, i);
    const double cd = cache.param(a.params, a.uniform_rows, 5, N, i);
    const int32_t scen = a.scen ? a.scen[i] : 0;
    // a linked forcing (rscm_ens_link_input, always the non-LDS variant) is another ensemble's
    // [T][N] series: coalesced, one stride of N per year
    const double* fglob = a.link ? a.link + (size_t)a.src_off * N + i : a.forcing + (size_t)scen * a.n_times + a.src_off;
    const size_t fstride = a.link ? (size_t)N : (size_t)1;
    const int32_t fl0 = scen * len - step_begin;  // lds_forcing[fl0 + n], n >= step_begin
    auto forcing_at = [&](int32_t n) -> double {
        if constexpr (LDS) return lds_forcing[fl0 + n];
        else return fglob[(size_t)n * fstride];
    };
    // the year a fused launch is at: the linked forcing from the producer's LDS slot if it is kept there
    auto forcing_first = [&]() -> double {
        if constexpr (Cache::kOn) {
            if (a.link && cache.has_link(0)) return cache.link(0);
        }
        return forcing_at(step_begin);
    };
    // next year's forcing: a fused launch calls per year (n == last), there is no next year to fetch
    auto forcing_ahead = [&](int32_t n, int32_t np, double current) -> double {
        if constexpr (Cache::kOn) return n < np ? forcing_at(np) : current;
        else return forcing_at(np);
    };

    double ts = cache.state(0, a.ts + (size_t)step_begin * N + i);
    double td = cache.state(1, a.td + (size_t)step_begin * N + i);
    double* out_ts = a.ts + (size_t)(step_begin + 1) * N + i;
    double* out_td = a.td + (size_t)(step_begin + 1) * N + i;

    const double h = a.h;
    const double half_step = a.h_half;
    const double sixth = a.h_sixth;
    const int32_t last = step_end - 1;

    LikAcc lik;
    if constexpr (!STORE) lik_consume(a, lik, step_begin, ts, td);  // observations of the start row

    // next year's forcing and sub-step count are fetched a year ahead of their use
    double erf_next = forcing_first();
    int32_t m_next = a.nsub[step_begin];

    if constexpr (MODE == 0) {
        TLConst p;
        p.lambda0 = lambda0;
        p.a = pa;
        p.eff_eta = efficacy * eta;
        p.eta = eta;
        p.cs = cs;
        p.cd = cd;
        const ConstDiv dcs = make_const_div(cs), dcd = make_const_div(cd);
        p.rcs = dcs.r;
        p.rcd = dcd.r;
        // 0 (never "all inside") when a heat capacity is outside the divisor window
        const int32_t acc0 = (dcs.ok && dcd.ok) ? (int32_t)0x80000000 : 0;
        for (int32_t n = step_begin; n < step_end; ++n) {
            const double erf = erf_next;
            const int32_t m = m_next;
            const int32_t np = n < last ? n + 1 : n;
            erf_next = forcing_ahead(n, np, erf_next);
            m_next = a.nsub[np];
            const double ts0 = ts, td0 = td;
            int32_t acc = acc0;
            for (int32_t s = 0; s < m; ++s) rk4_step_exact<true>(p, erf, h, half_step, sixth, ts, td, acc);
            // A NaN state at the start of the year makes every value of the year NaN on either
            // path; everything else must have stayed inside the window.
            const bool settled = (ts0 != ts0) || (td0 != td0);
            if (__builtin_expect(acc >= 0 && !settled, 0)) {
                ts = ts0;
                td = td0;
                int32_t unused = 0;
                for (int32_t s = 0; s < m; ++s) rk4_step_exact<false>(p, erf, h, half_step, sixth, ts, td, unused);
            }
            if constexpr (STORE) {
                *out_ts = ts;
                *out_td = td;
                cache.put(0, ts);
                cache.put(1, td);
                out_ts += N;
                out_td += N;
            } else {
                lik_consume(a, lik, n + 1, ts, td);
            }
        }
    } else {
        double inv_cs;
        const TLFast p = make_fast(lambda0, pa, efficacy, eta, cs, cd, inv_cs);
        const double third = h / 3.0;
        for (int32_t n = step_begin; n < step_end; ++n) {
            const double erf = erf_next * inv_cs;
            const int32_t m = m_next;
            const int32_t np = n < last ? n + 1 : n;
            erf_next = forcing_ahead(n, np, erf_next);
            m_next = a.nsub[np];
            rk4_year_fast(p, erf, m, h, half_step, third, sixth, ts, td);
            if constexpr (STORE) {
                *out_ts = ts;
                *out_td = td;
                cache.put(0, ts);
                cache.put(1, td);
                out_ts += N;
                out_td += N;
            } else {
                lik_consume(a, lik, n + 1, ts, td);
            }
        }
    }
    if (cache.last_step()) a.status[i] = (is_finite(ts) && is_finite(td)) ? 0 : 1;
    if constexpr (!STORE) {
        // observations whose row is never reached were never computed -> member failure
        if (lik.oi < a.n_obs) lik.bad = true;
        const double total = a.first_is_deep ? (0.0 + lik.part_d) + lik.part_s
                                             : (0.0 + lik.part_s) + lik.part_d;
        a.loglik[i] = lik.bad ? -__builtin_inf() : total;
    }
}

}  // namespace tl
}  // namespace rscm
