// Device code of CarbonCycle, CO2Budget and TerrestrialCarbon, shared by carbon.hip (one launch per component) and group.hip (several linked
// components of one model step in one launch).  See carbon.hip for what it replaces in the reference.
#pragma once

#include "rk4_device.hpp"
#include "rscm_device.hpp"

namespace rscm {
namespace carbon {

constexpr double kGtcPerPpm = 2.13;  // crates/rscm-components/src/constants.rs:37

// rscm-components' CarbonCycle on its own (carbon_cycle.rs:102-159): y = (C, cumulative uptake,
// cumulative emissions) integrated with RK4 over the model step, emissions and temperature
// constant over it (get() ignores t).  The arithmetic is that of the fused coupled chain
// (csrc/coupled.hip, year<false>), so a graph assembled from linked ensembles reproduces the
// fused kind bit for bit.  in = {Emissions|CO2|Anthropogenic, Surface Temperature}.
// The RK4 sub-steps of one model step.  SPEC: the forty divisions by the year's lifetime as the
// three-instruction quotient with the reciprocal hoisted (rk4_device.hpp), the numerators' exponent-window
// tags folded into `acc` -- the year is replayed with the compiler's divisions if any tag falls outside,
// as in the fused chain (coupled.hip, year<SPEC>), instead of a branch per division.
template <bool SPEC>
__device__ __forceinline__ int32_t carbon_cycle_year(double lifetime, double rlife, double conc_pi, double emis, double e_ppm, int32_t m, double hc,
                                                     double half_c, double sixth_c, double& conc, double& cum_u, double& cum_e, int32_t acc)
{
    auto quot = [&](double n, int32_t& tag) -> double {
        if constexpr (SPEC) {
            tag = window_tag(n);
            return spec_div(n, lifetime, rlife);
        } else {
            return n / lifetime;
        }
    };
    for (int32_t s = 0; s < m; ++s) {
        int32_t t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const double up1 = quot(conc - conc_pi, t1);
        const double k1c = e_ppm - up1, k1u = up1 * kGtcPerPpm;
        const double up2 = quot((conc + k1c * half_c) - conc_pi, t2);
        const double k2c = e_ppm - up2, k2u = up2 * kGtcPerPpm;
        const double up3 = quot((conc + k2c * half_c) - conc_pi, t3);
        const double k3c = e_ppm - up3, k3u = up3 * kGtcPerPpm;
        const double up4 = quot((conc + k3c * hc) - conc_pi, t4);
        const double k4c = e_ppm - up4, k4u = up4 * kGtcPerPpm;
        if constexpr (SPEC) {
            acc = max3_i32(max3_i32(acc, t1, t2), t3, t4);
            // inside the windows 2*k cannot overflow: the doubling fused into the addition rounds the same
            conc = rk4_combine_fused2(conc, k1c, k2c, k3c, k4c, sixth_c);
            cum_u = rk4_combine_fused2(cum_u, k1u, k2u, k3u, k4u, sixth_c);
        } else {
            conc = rk4_combine(conc, k1c, k2c, k3c, k4c, sixth_c);
            cum_u = rk4_combine(cum_u, k1u, k2u, k3u, k4u, sixth_c);
        }
        cum_e = rk4_combine(cum_e, emis, emis, emis, emis, sixth_c);
    }
    return acc;
}

// RSCM_MODE_FAST.  Over one model step emissions and temperature are constants (get() ignores t), so the
// concentration obeys the LINEAR equation  C' = A - r C  with  r = 1/lifetime = exp(-alpha T) / tau  and
// A = E/2.13 + r C_pi, and the four RK4 stages of such an equation collapse algebraically:
//     k2 = k1 (1 - z/2),  k3 = k1 (1 - z/2 + z^2/4),  k4 = k1 (1 - z + z^2/2 - z^3/4),   z = h r
//     C <- C + h phi(z) k1,   phi(z) = 1 - z/2 + z^2/6 - z^3/24,   k1 = A - r C
// which is the value the reference's forty separately rounded stage evaluations per step approximate
// (carbon_cycle.rs:102-159 under ivp/mod.rs:245-253).  In every stage dC + dU/2.13 = E/2.13, so the uptake
// integral is  U <- U + 2.13 (m h E/2.13 - sum of the C increments); the increments are summed on their own
// (relative to their size, no cancellation against C).  Cumulative emissions keep the reference's association
// -- their RK4 increment (((E + 2E) + 2E) + E) h/6 depends on E alone, so the EXACT bits cost one addition per
// sub-step.  No division: 1/tau is formed once per member, 1/2.13 is a constant.  Three FMAs and an addition per
// sub-step against ~60 instructions of the EXACT body; within 3e-13 of the oracle on all three states
// (tests/test_gpu_parity.py), shared by the fused chain (coupled.hip) and the linked component (below).
constexpr double kPpmPerGtc = 1.0 / 2.13;

__device__ __forceinline__ void carbon_cycle_year_fast(double rtau, double alpha, double conc_pi, double emis, double temperature, int32_t m,
                                                       double hc, double sixth_c, double& conc, double& cum_u, double& cum_e)
{
    const double r = rtau * exp(-(alpha * temperature));
    const double e_ppm = emis * kPpmPerGtc;
    const double A = __builtin_fma(conc_pi, r, e_ppm);
    const double z = hc * r;
    const double phi = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, -1.0 / 24.0, 1.0 / 6.0), -0.5), 1.0);
    const double hphi = hc * phi;
    const double inc_e = (((emis + emis * 2.0) + emis * 2.0) + emis) * sixth_c;
    double dsum = 0.0;
    for (int32_t s = 0; s < m; ++s) {
        const double k1 = __builtin_fma(-r, conc, A);
        conc = __builtin_fma(k1, hphi, conc);
        dsum = __builtin_fma(k1, hphi, dsum);
        cum_e = cum_e + inc_e;
    }
    cum_u = __builtin_fma(kGtcPerPpm, __builtin_fma((double)m * hc, e_ppm, -dsum), cum_u);
}

template <int MODE, int SRC, class Cache = NoCache>
__device__ __forceinline__ void carbon_cycle_body(const CarbonArgs& a, int64_t i, int32_t step_begin, int32_t step_end, const Cache& cache = Cache())
{
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    const double tau = cache.param(a.params, a.uniform_rows, 0, N, i), conc_pi = cache.param(a.params, a.uniform_rows, 1, N, i),
                 alpha = cache.param(a.params, a.uniform_rows, 2, N, i);
    const MemberInputs<SRC, 2> in(a.inputs, a.scen, a.links, T, N, i);
    const size_t vs = (size_t)a.rows * N;
    const size_t r0 = (size_t)step_begin * N + i;
    double conc = cache.state(0, a.series + r0), cum_u = cache.state(1, a.series + vs + r0), cum_e = cache.state(2, a.series + 2 * vs + r0);
    const double hc = a.h, half_c = a.h_half, sixth_c = a.h_sixth;
    if constexpr (MODE != 0) {
        const double rtau = 1.0 / tau;
        for (int32_t n = step_begin; n < step_end; ++n) {
            const double emis = in.at(0, n, cache), temperature = in.at(1, n, cache);
            carbon_cycle_year_fast(rtau, alpha, conc_pi, emis, temperature, a.nsub[n], hc, sixth_c, conc, cum_u, cum_e);
            const size_t r = (size_t)(n + 1) * N + i;
            a.series[r] = conc;
            a.series[vs + r] = cum_u;
            a.series[2 * vs + r] = cum_e;
            cache.put(0, conc);
            cache.put(1, cum_u);
            cache.put(2, cum_e);
        }
        if (cache.last_step()) a.status[i] = (is_finite(conc) && is_finite(cum_u) && is_finite(cum_e)) ? 0 : 1;
        return;
    }
    for (int32_t n = step_begin; n < step_end; ++n) {
        // (a fused launch calls per model step: n is the step it is at, which is what the cache holds)
        const double emis = in.at(0, n, cache), temperature = in.at(1, n, cache);
        const double lifetime = tau * exp(alpha * temperature);
        const double rlife = refined_rcp(lifetime);
        const int32_t acc0 = divisor_in_window(lifetime) ? (int32_t)0x80000000 : 0;  // 0: never "all inside"
        const double e_ppm = emis / kGtcPerPpm;
        const int32_t m = a.nsub[n];
        const double conc0 = conc, cum_u0 = cum_u, cum_e0 = cum_e;
        const int32_t acc = carbon_cycle_year<true>(lifetime, rlife, conc_pi, emis, e_ppm, m, hc, half_c, sixth_c, conc, cum_u, cum_e, acc0);
        // A NaN lifetime makes the concentration and the uptake NaN on either path (cumulative emissions
        // involve no division); everything else must have stayed inside the windows.
        if (__builtin_expect(acc >= 0 && lifetime == lifetime, 0)) {
            conc = conc0;
            cum_u = cum_u0;
            cum_e = cum_e0;
            carbon_cycle_year<false>(lifetime, rlife, conc_pi, emis, e_ppm, m, hc, half_c, sixth_c, conc, cum_u, cum_e, 0);
        }
        const size_t r = (size_t)(n + 1) * N + i;
        a.series[r] = conc;
        a.series[vs + r] = cum_u;
        a.series[2 * vs + r] = cum_e;
        cache.put(0, conc);
        cache.put(1, cum_u);
        cache.put(2, cum_e);
    }
    if (cache.last_step()) a.status[i] = (is_finite(conc) && is_finite(cum_u) && is_finite(cum_e)) ? 0 : 1;
}

template <int SRC, class Cache = NoCache>
__device__ __forceinline__ void co2_budget_body(const CarbonArgs& a, int64_t i, int32_t step_begin, int32_t step_end, const Cache& cache = Cache())
{
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    const double gtc_per_ppm = cache.param(a.params, a.uniform_rows, 0, N, i);
    const MemberInputs<SRC, 4> in(a.inputs, a.scen, a.links, T, N, i);
    const size_t vs = (size_t)a.rows * N;
    double co2 = cache.state(0, a.series + (size_t)step_begin * N + i);
    for (int32_t n = step_begin; n < step_end; ++n) {
        const double dt = a.bounds[n + 1] - a.bounds[n];
        const double total_emissions = in.at(0, n, cache) + in.at(1, n, cache);
        const double total_uptake = in.at(2, n, cache) + in.at(3, n, cache);
        const double net_to_atm = total_emissions - total_uptake;
        co2 = co2 + (net_to_atm * dt) / gtc_per_ppm;
        const double airborne = total_emissions > 0.0 ? net_to_atm / total_emissions : 0.0;
        const size_t r = (size_t)(n + 1) * N + i;
        a.series[r] = co2;
        a.series[vs + r] = net_to_atm;
        a.series[2 * vs + r] = airborne;
        cache.put(0, co2);
        cache.put(1, net_to_atm);
        cache.put(2, airborne);
    }
    if (cache.last_step()) a.status[i] = 0;   // (last: the byte's store may alias anything, no load moves across it)
}

// carbon/terrestrial.rs:82-100
// r_tau = 1 / tau (inverted once per member); the quotient by 1 + half_k is a product with a refined
// reciprocal: TerrestrialCarbon is a tolerance-parity kind (its log / exp come from the device library)
__device__ __forceinline__ void implicit_pool_step(double pool, double r_tau, double flux_in, double temp_factor, double dt,
                                                   double& new_pool, double& turnover)
{
    const double k_eff = temp_factor * r_tau;
    const double half_k = 0.5 * k_eff * dt;
    double np = ((1.0 - half_k) * pool + flux_in * dt) * guarded_rcp(1.0 + half_k);
    np = fmax(np, 0.0);
    new_pool = np;
    turnover = 0.5 * k_eff * (pool + np);
}

// parameters/terrestrial_carbon.rs:103-168: the four turnover times follow from the pre-industrial pools and fluxes alone.  Formed
// once per parameter set (launch_terrestrial_derive) and read back by the body:
//   [0] f_npp_soil   [1..4] 1 / tau of plant, detritus, soil, humus
__device__ __forceinline__ void terrestrial_member_constants(double npp_pi, double plant_pi, double det_pi, double soil_pi, double hum_pi, double resp_pi,
                                                             double f_npp_plant, double f_npp_det, double f_plant_det, double f_det_soil,
                                                             double f_soil_hum, double (&d)[kDerivedRows])
{
    const double f_npp_soil = fmax(1.0 - f_npp_plant - f_npp_det, 0.0);
    const double net_plant = f_npp_plant * npp_pi - resp_pi;
    const double tau_plant = net_plant > 1e-10 ? plant_pi / net_plant : 100.0;
    const double flux_det = f_npp_det * npp_pi + f_plant_det * net_plant;
    const double tau_det = flux_det > 1e-10 ? det_pi / flux_det : 3.0;
    const double flux_soil = f_npp_soil * npp_pi + (1.0 - f_plant_det) * net_plant + f_det_soil * (det_pi / tau_det);
    const double tau_soil = flux_soil > 1e-10 ? soil_pi / flux_soil : 50.0;
    const double flux_hum = f_soil_hum * (soil_pi / tau_soil);
    const double tau_hum = flux_hum > 1e-10 ? hum_pi / flux_hum : 1000.0;
    d[0] = f_npp_soil;
    d[1] = guarded_rcp(tau_plant);
    d[2] = guarded_rcp(tau_det);
    d[3] = guarded_rcp(tau_soil);
    d[4] = guarded_rcp(tau_hum);
    d[5] = d[6] = d[7] = 0.0;
}

template <int SRC>
__device__ __forceinline__ void terrestrial_body(const CarbonArgs& a, int64_t i, int32_t step_begin, int32_t step_end)
{
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    // state and the first step's rows first: in flight together with the parameters (rscm_device.hpp, StepRows)
    const MemberInputs<SRC, 3> in(a.inputs, a.scen, a.links, T, N, i);
    const size_t vs = (size_t)a.rows * N;
    const size_t r0 = (size_t)step_begin * N + i;
    double plant = a.series[r0], det = a.series[vs + r0], soil = a.series[2 * vs + r0], hum = a.series[3 * vs + r0];
    StepRows<3> ahead = rows_at(in, step_begin);
    double mc[kDerivedRows];   // the member's constants (launch_terrestrial_derive), in flight with the parameters
    params_block<kDerivedRows>(a.derived, a.derived_uniform ? ~0ull : 0ull, N, i, mc);
    const double npp_pi = P(0), co2_pi = P(1), beta = P(2), npp_ts = P(3), resp_ts = P(4), det_ts = P(5), soil_ts = P(6),
                 hum_ts = P(7), resp_pi = P(12), f_npp_plant = P(13), f_npp_det = P(14), f_plant_det = P(15), f_det_soil = P(16),
                 f_soil_hum = P(17);
    const bool fert_on = P(18) != 0.0, temp_on = P(19) != 0.0;
    const double f_npp_soil = mc[0], r_tau_plant = mc[1], r_tau_det = mc[2], r_tau_soil = mc[3], r_tau_hum = mc[4];
    for (int32_t n = step_begin; n < step_end; ++n) {
        const double dt = a.bounds[n + 1] - a.bounds[n];
        const StepRows<3> now = ahead;
        if (n + 1 < step_end) ahead = rows_at(in, n + 1);
        const double co2 = now.v[0], temperature = now.v[1], landuse = now.v[2];
        const double fert = (!fert_on || co2 <= 0.0) ? 1.0 : fmax(1.0 + beta * log_f64(co2 / co2_pi), 0.1);
        auto tf = [&](double sens) -> double { return temp_on ? exp(sens * temperature) : 1.0; };
        const double npp = npp_pi * fert * tf(npp_ts);
        const double respiration = resp_pi * fert * tf(resp_ts);
        const double tf_det = tf(det_ts), tf_soil = tf(soil_ts), tf_hum = tf(hum_ts);
        double n_plant, to_plant, n_det, to_det, n_soil, to_soil, n_hum, to_hum;
        implicit_pool_step(plant, r_tau_plant, npp * f_npp_plant - respiration - landuse, 1.0, dt, n_plant, to_plant);
        implicit_pool_step(det, r_tau_det, npp * f_npp_det + f_plant_det * to_plant, tf_det, dt, n_det, to_det);
        const double npp_to_soil = npp * f_npp_soil;
        const double plant_to_soil = (1.0 - f_plant_det) * to_plant;
        const double det_to_soil = f_det_soil * to_det;
        implicit_pool_step(soil, r_tau_soil, npp_to_soil + plant_to_soil + det_to_soil, tf_soil, dt, n_soil, to_soil);
        implicit_pool_step(hum, r_tau_hum, f_soil_hum * to_soil, tf_hum, dt, n_hum, to_hum);
        const double det_to_atm = (1.0 - f_det_soil) * to_det;
        const double soil_to_atm = (1.0 - f_soil_hum) * to_soil;
        const double total_resp = respiration + det_to_atm + soil_to_atm + to_hum;
        const size_t r = (size_t)(n + 1) * N + i;
        a.series[r] = plant = n_plant;
        a.series[vs + r] = det = n_det;
        a.series[2 * vs + r] = soil = n_soil;
        a.series[3 * vs + r] = hum = n_hum;
        a.series[4 * vs + r] = npp - total_resp - landuse;
    }
    a.status[i] = 0;
}


}  // namespace carbon
}  // namespace rscm
