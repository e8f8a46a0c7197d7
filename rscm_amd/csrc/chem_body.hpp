// Device code of CH4Chemistry and N2OChemistry, shared by chem.hip (one launch per component) and group.hip (several linked
// components of one model step in one launch).  See chem.hip for what it replaces in the reference.
#pragma once

#include "rk4_device.hpp"
#include "rscm_device.hpp"

namespace rscm {
namespace chem {

constexpr int kPratherIterations = 4;

// x^y for x >= 1 as exp(y ln x): the lifetime ratios are in [1, ~10] and |y| < 1, so the product
// y ln x is O(1) and carries ~1 ulp of ln's error into an exponent of that size -- a relative
// error of ~1e-16 in the power, at a third of the instructions of the general pow().
__device__ __forceinline__ double pow_ratio(double x, double y) { return exp(y * log_f64(x)); }

template <int SRC>
__device__ __forceinline__ void ch4_body(const ChemArgs& a, int64_t i, int32_t step_begin, int32_t step_end)
{
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    // state and the first step's rows first: in flight together with the parameters (rscm_device.hpp, StepRows)
    const MemberInputs<SRC, 5> in(a.inputs, a.scen, a.links, T, N, i);
    double cur = a.conc[(size_t)step_begin * N + i];
    double prev = step_begin > 0 ? a.conc[(size_t)(step_begin - 1) * N + i] : cur;  // previous().unwrap_or(current)
    StepRows<5> ahead = rows_at(in, step_begin);
    const double ch4_pi = P(0), natural = P(1), tau_oh0 = P(2);
    const double tau_other = 1.0 / (1.0 / P(3) + 1.0 / P(4) + 1.0 / P(5));  // parameters/ch4_chemistry.rs tau_other
    const double self_fb = P(6), gamma = P(7), s_nox = P(8), s_co = P(9), s_nmvoc = P(10), temp_sens = P(11);
    const bool incl_temp = P(12) != 0.0, incl_emis = P(13) != 0.0;
    const double ppb_to_tg = P(14), nox_ref = P(15), co_ref = P(16), nmvoc_ref = P(17);
    const double burden_reference = ch4_pi * ppb_to_tg;
    const double x = -gamma * self_fb;
    const double r_ref = guarded_rcp(burden_reference), r_other = guarded_rcp(tau_other), r_ppb = guarded_rcp(ppb_to_tg);
    const double r_tau0 = guarded_rcp(tau_oh0);
    for (int32_t n = step_begin; n < step_end; ++n) {
        const StepRows<5> now = ahead;
        if (n + 1 < step_end) ahead = rows_at(in, n + 1);
        const double emissions = now.v[0], temperature = now.v[1];
        const double delta_nox = now.v[2] - nox_ref, delta_co = now.v[3] - co_ref;
        const double delta_nmvoc = now.v[4] - nmvoc_ref;
        const double total_emissions = emissions + natural;
        const double burden_prev = prev * ppb_to_tg;
        double base = tau_oh0;
        if (incl_emis) base = tau_oh0 * exp(-gamma * (s_nox * delta_nox + s_co * delta_co + s_nmvoc * delta_nmvoc));
        double burden = cur * ppb_to_tg;
        // The ~27 quotients of a member-year are products with refined reciprocals (a tolerance-parity
        // kind: pow / exp already come from the device library); divisors that do not change are
        // inverted once per member (r_ref, r_other, r_ppb, r_tau0) or per year (r_prev), and the
        // temperature feedback tau0 / (tau0 / tau + s dT) is carried as its reciprocal.
        double delta_burden = 0.0, r_tau = r_tau0;
        const bool prev_ok = !(fabs(burden_prev) < 1e-10);
        const double r_prev = prev_ok ? guarded_rcp(burden_prev) : 0.0;
        const bool temp_on = incl_temp && !(fabs(temperature) < 1e-10);
        const double ts_dt = temp_sens * fmax(temperature, 0.0);
#pragma unroll
        for (int it = 0; it < kPratherIterations; ++it) {
            const double burden_mean = (burden + burden_prev) / 2.0;
            const double ratio = fmax(burden_mean * r_ref, 1.0);
            double tau_oh = base * pow_ratio(ratio, x);
            if (it > 0 && prev_ok) tau_oh = tau_oh * (1.0 - 0.5 * x * delta_burden * r_prev);
            r_tau = guarded_rcp(tau_oh);
            if (temp_on) r_tau = __builtin_fma(tau_oh0, r_tau, ts_dt) * r_tau0;  // 1 / (tau0 / (tau0 / tau + s dT))
            delta_burden = total_emissions - burden_mean * r_tau - burden_mean * r_other;
            burden = burden_prev + delta_burden;
        }
        const double next = burden * r_ppb;
        const size_t r = (size_t)(n + 1) * N + i;
        a.conc[r] = next;
        a.lifetime[r] = guarded_rcp(r_tau + r_other);
        prev = cur;
        cur = next;
    }
    a.status[i] = 0;
}

template <int SRC>
__device__ __forceinline__ void n2o_body(const ChemArgs& a, int64_t i, int32_t step_begin, int32_t step_end)
{
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    // state and the first step's row first: in flight together with the parameters (the delayed rows cannot be: their
    // address is a parameter)
    const MemberInputs<SRC, 1> in(a.inputs, a.scen, a.links, T, N, i);
    auto C = [&](int64_t k) -> double { return a.conc[(size_t)k * N + i]; };
    double cur = C(step_begin);
    double prev = step_begin > 0 ? C(step_begin - 1) : cur;
    StepRows<1> ahead = rows_at(in, step_begin);
    const double n2o_pi = P(0), natural = P(1), tau0 = P(2), lifetime_fb = P(3), ppb_to_tg = P(5);
    int64_t delay = (int64_t)P(4);
    if (delay < 1) delay = 1;  // strat_delay.max(1)
    const double burden_reference = n2o_pi * ppb_to_tg;
    const double r_ref = guarded_rcp(burden_reference), r_ppb = guarded_rcp(ppb_to_tg);
    for (int32_t n = step_begin; n < step_end; ++n) {
        const double dt = a.bounds[n + 1] - a.bounds[n];
        const StepRows<1> now = ahead;
        if (n + 1 < step_end) ahead = rows_at(in, n + 1);
        // n2o.rs:203-218: at_offset(-delay) else previous; at_offset(-(delay+1)) else the former
        double t_delay = prev;  // delay == 1: C(n-1) is `prev` already (and the fall-back for n == 0)
        if (delay > 1 && (int64_t)n - delay >= 0) t_delay = C((int64_t)n - delay);
        double t_delay_m1 = t_delay;
        if ((int64_t)n - delay - 1 >= 0) t_delay_m1 = C((int64_t)n - delay - 1);
        const double lagged = (t_delay + t_delay_m1) / 2.0;
        const double total_emissions = now.v[0] + natural;
        const double burden_prev = prev * ppb_to_tg, burden_lagged = lagged * ppb_to_tg;
        double burden = cur * ppb_to_tg, tau_eff = tau0;
#pragma unroll
        for (int it = 0; it < kPratherIterations; ++it) {
            const double burden_mid = (burden_prev + burden) / 2.0;
            const double ratio = fmax(burden_mid * r_ref, 1.0);   // quotients as products with refined reciprocals, as in ch4_kernel
            tau_eff = tau0 * pow_ratio(ratio, lifetime_fb);
            const double rate = total_emissions - burden_lagged * guarded_rcp(tau_eff);
            burden = burden_prev + rate * dt;
        }
        const double next = burden * r_ppb;
        const size_t r = (size_t)(n + 1) * N + i;
        a.conc[r] = next;
        a.lifetime[r] = tau_eff;
        prev = cur;
        cur = next;
    }
    a.status[i] = 0;
}


}  // namespace chem
}  // namespace rscm
