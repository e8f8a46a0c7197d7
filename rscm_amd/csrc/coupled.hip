// Coupled chain CarbonCycle -> CO2ERF -> Sum aggregate -> TwoLayer, fused per member (gfx950).
//
// Registration order CarbonCycle, CO2ERF, TwoLayer with the schema aggregate
// "Effective Radiative Forcing" = Sum(["Effective Radiative Forcing|CO2"])
// (docs/notebooks/coupled_model.py:435-483) fixes, through the registration-order rule of
// crates/rscm-core/src/model/builder.rs:470-482, which time index every read uses; the kernel
// hard-codes that resolved schedule.  Per step n:
//   CarbonCycle  reads E[n] (exogenous) and Ts[n] (classified Exogenous because TwoLayer is
//                registered later => lagged feedback), own states C,U,S at n
//                crates/rscm-components/src/components/carbon_cycle.rs:102-159
//                lifetime = tau*exp(alpha*T) is constant over the step (get() ignores t), so
//                exp() and the reciprocal of lifetime are evaluated once per step, not 4x10.
//   CO2ERF       reads C[n+1] (upstream)       co2_erf.rs:57-80
//   Sum          ERF[n+1] = sum of non-NaN contributors at n+1   schema.rs:760-773,886-901
//   TwoLayer     reads ERF[n+1] (upstream), Ts[n], Td[n]         rscm-two-layer/src/component.rs:159-251
// Seven series are stored per step: 56 B per member-year.
//
// Arithmetic follows the reference's expression order with no FMA contraction; divisions by the
// per-year lifetime and the per-member heat capacities use the speculative-year scheme of
// two_layer.hip (three-instruction quotient + exponent-window tags, full IEEE division replay
// when a tag falls outside).  exp/log come from the device math library and may differ from
// glibc's by an ulp, so this kind is tolerance-parity (tests/test_gpu_parity.py), not bit-parity.
#include "carbon_body.hpp"
#include "two_layer_body.hpp"

namespace rscm {

namespace {

constexpr double kGtcPerPpm = 2.13;               // crates/rscm-components/src/constants.rs:37
constexpr double kLn2 = 0.693147180559945309417;  // 2.0_f64.ln()

struct CPConst {
    double lambda0, a, eff_eta, eta, cs, cd, rcs, rcd;  // TwoLayer
    double tau, conc_pi, alpha, erf_scale;              // CarbonCycle, CO2ERF (erf_2xco2 / ln 2)
};

struct CPState {
    double ts, td, conc, cum_u, cum_e;
};

template <bool SPEC>
__device__ __forceinline__ double quot(double n, double d, double r, int32_t& tag)
{
    if constexpr (SPEC) {
        tag = window_tag(n);
        return spec_div(n, d, r);
    } else {
        return n / d;
    }
}

template <bool SPEC>
__device__ __forceinline__ void tl_rhs(const CPConst& p, double erf, double ts, double td,
                                       double& dts, double& dtd, int32_t& acc)
{
    const double diff = ts - td;
    const double lambda_eff = p.lambda0 - p.a * ts;
    const double hx_s = p.eff_eta * diff;
    int32_t t1 = 0, t2 = 0;
    dts = quot<SPEC>(erf - lambda_eff * ts - hx_s, p.cs, p.rcs, t1);
    dtd = quot<SPEC>(p.eta * diff, p.cd, p.rcd, t2);
    if constexpr (SPEC) acc = max3_i32(acc, t1, t2);
}

// One model year for one member.  Returns the exponent-window accumulator (SPEC) or 0.
template <bool SPEC>
__device__ __forceinline__ int32_t year(const CPConst& p, double emis, int32_t mc, int32_t mt,
                                        double hc, double ht, CPState& y, double& erf_co2,
                                        double& erf, int32_t acc)
{
    // ---- CarbonCycle over [b[n], b[n+1]] with E[n] and T = Ts[n]
    const double lifetime = p.tau * exp(p.alpha * y.ts);
    double rlife = 0.0;
    if constexpr (SPEC) {
        rlife = refined_rcp(lifetime);
        if (!divisor_in_window(lifetime)) acc = 0;
    }
    const double e_ppm = emis / kGtcPerPpm;
    const double half_c = hc / 2.0, sixth_c = hc / 6.0;
    for (int32_t s = 0; s < mc; ++s) {
        // y = (conc, cum_uptake, cum_emissions); dy = (E/2.13 - up, up*2.13, E)
        int32_t t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const double up1 = quot<SPEC>(y.conc - p.conc_pi, lifetime, rlife, t1);
        const double k1c = e_ppm - up1, k1u = up1 * kGtcPerPpm;
        const double up2 = quot<SPEC>((y.conc + k1c * half_c) - p.conc_pi, lifetime, rlife, t2);
        const double k2c = e_ppm - up2, k2u = up2 * kGtcPerPpm;
        const double up3 = quot<SPEC>((y.conc + k2c * half_c) - p.conc_pi, lifetime, rlife, t3);
        const double k3c = e_ppm - up3, k3u = up3 * kGtcPerPpm;
        const double up4 = quot<SPEC>((y.conc + k3c * hc) - p.conc_pi, lifetime, rlife, t4);
        const double k4c = e_ppm - up4, k4u = up4 * kGtcPerPpm;
        if constexpr (SPEC) {
            acc = max3_i32(max3_i32(acc, t1, t2), t3, t4);
            y.conc = rk4_combine_fused2(y.conc, k1c, k2c, k3c, k4c, sixth_c);
            y.cum_u = rk4_combine_fused2(y.cum_u, k1u, k2u, k3u, k4u, sixth_c);
        } else {
            y.conc = rk4_combine(y.conc, k1c, k2c, k3c, k4c, sixth_c);
            y.cum_u = rk4_combine(y.cum_u, k1u, k2u, k3u, k4u, sixth_c);
        }
        y.cum_e = rk4_combine(y.cum_e, emis, emis, emis, emis, sixth_c);
    }
    // ---- CO2ERF on C[n+1], then the Sum aggregate over its single contributor
    erf_co2 = p.erf_scale * log_f64(1.0 + (y.conc - p.conc_pi) / p.conc_pi);
    erf = (erf_co2 != erf_co2) ? erf_co2 : 0.0 + erf_co2;
    // ---- TwoLayer with ERF[n+1]
    const double half_t = ht / 2.0, sixth_t = ht / 6.0;
    for (int32_t s = 0; s < mt; ++s) {
        double k1s, k1d, k2s, k2d, k3s, k3d, k4s, k4d;
        tl_rhs<SPEC>(p, erf, y.ts, y.td, k1s, k1d, acc);
        tl_rhs<SPEC>(p, erf, y.ts + k1s * half_t, y.td + k1d * half_t, k2s, k2d, acc);
        tl_rhs<SPEC>(p, erf, y.ts + k2s * half_t, y.td + k2d * half_t, k3s, k3d, acc);
        tl_rhs<SPEC>(p, erf, y.ts + k3s * ht, y.td + k3d * ht, k4s, k4d, acc);
        if constexpr (SPEC) {
            y.ts = rk4_combine_fused2(y.ts, k1s, k2s, k3s, k4s, sixth_t);
            y.td = rk4_combine_fused2(y.td, k1d, k2d, k3d, k4d, sixth_t);
        } else {
            y.ts = rk4_combine(y.ts, k1s, k2s, k3s, k4s, sixth_t);
            y.td = rk4_combine(y.td, k1d, k2d, k3d, k4d, sixth_t);
        }
    }
    return acc;
}

// RSCM_MODE_FAST: one member, the whole axis.  CarbonCycle as the collapsed linear RK4 step of carbon_body.hpp
// (no division, one exp per year), CO2ERF and the Sum aggregate in their one arithmetic (a division by the
// pre-industrial concentration and a log per year), TwoLayer with the heat capacities folded into its
// coefficients and FMA stages (two_layer_body.hpp, rk4_year_fast) -- the very functions the linked components
// call in this mode, so "four linked ensembles == the fused kind, bit for bit" holds per mode.
template <bool LDS>
__global__ __launch_bounds__(kBlock) void coupled_fast_kernel(CoupledArgs a)
{
    extern __shared__ double lds_emis[];
    const int32_t len = a.step_end - a.step_begin;
    if constexpr (LDS) {
        const int32_t total = a.n_scen * len;
        for (int32_t idx = threadIdx.x; idx < total; idx += kBlock) {
            const int32_t s = idx / len, k = idx - s * len;
            lds_emis[idx] = a.emissions[(size_t)s * a.n_times + a.step_begin + k];
        }
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    const int64_t N = a.row_stride;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };

    double inv_cs;
    const tl::TLFast tp = tl::make_fast(P(0), P(1), P(2), P(3), P(4), P(5), inv_cs);
    const double rtau = 1.0 / P(6), conc_pi = P(7), alpha = P(8), erf_scale = P(9) / kLn2;
    const int32_t scen = a.scen ? a.scen[i] : 0;
    const double* eglob = a.emissions + (size_t)scen * a.n_times;
    const int32_t el0 = scen * len - a.step_begin;
    auto emissions_at = [&](int32_t n) -> double {
        if constexpr (LDS) return lds_emis[el0 + n];
        else return eglob[n];
    };

    const size_t r0 = (size_t)a.step_begin * N + i;
    double ts = a.ts[r0], td = a.td[r0], conc = a.conc[r0], cum_u = a.cum_uptake[r0], cum_e = a.cum_emis[r0];
    size_t r = r0 + (size_t)N;
    const int32_t last = a.step_end - 1;
    double emis_next = emissions_at(a.step_begin);
    int32_t mc_next = a.nsub_cc[a.step_begin], mt_next = a.nsub_tl[a.step_begin];
    const double hc = a.h_cc, sixth_c = a.h_cc / 6.0;
    const double ht = a.h_tl, half_t = a.h_tl / 2.0, third_t = a.h_tl / 3.0, sixth_t = a.h_tl / 6.0;

    for (int32_t n = a.step_begin; n < a.step_end; ++n) {
        const double emis = emis_next;
        const int32_t mc = mc_next, mt = mt_next;
        const int32_t np = n < last ? n + 1 : n;
        emis_next = emissions_at(np);
        mc_next = a.nsub_cc[np];
        mt_next = a.nsub_tl[np];
        carbon::carbon_cycle_year_fast(rtau, alpha, conc_pi, emis, ts, mc, hc, sixth_c, conc, cum_u, cum_e);
        const double erf_co2 = erf_scale * log_f64(1.0 + (conc - conc_pi) / conc_pi);
        const double erf = (erf_co2 != erf_co2) ? erf_co2 : 0.0 + erf_co2;
        const double erf_cs = erf * inv_cs;
        tl::rk4_year_fast(tp, erf_cs, mt, ht, half_t, third_t, sixth_t, ts, td);
        a.conc[r] = conc;
        a.cum_uptake[r] = cum_u;
        a.cum_emis[r] = cum_e;
        a.erf_co2[r] = erf_co2;
        a.erf_total[r] = erf;
        a.ts[r] = ts;
        a.td[r] = td;
        r += (size_t)N;
    }
    a.status[i] = (is_finite(ts) && is_finite(td) && is_finite(conc) && is_finite(cum_u) && is_finite(cum_e)) ? 0 : 1;
}

template <bool LDS>
__global__ __launch_bounds__(kBlock) void coupled_kernel(CoupledArgs a)
{
    extern __shared__ double lds_emis[];
    const int32_t len = a.step_end - a.step_begin;
    if constexpr (LDS) {
        const int32_t total = a.n_scen * len;
        for (int32_t idx = threadIdx.x; idx < total; idx += kBlock) {
            const int32_t s = idx / len, k = idx - s * len;
            lds_emis[idx] = a.emissions[(size_t)s * a.n_times + a.step_begin + k];
        }
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    const int64_t N = a.row_stride;

    CPConst p;
    p.lambda0 = param_at(a.params, a.uniform_rows, 0, N, i);
    p.a = param_at(a.params, a.uniform_rows, 1, N, i);
    p.eff_eta = param_at(a.params, a.uniform_rows, 2, N, i) * param_at(a.params, a.uniform_rows, 3, N, i);
    p.eta = param_at(a.params, a.uniform_rows, 3, N, i);
    p.cs = param_at(a.params, a.uniform_rows, 4, N, i);
    p.cd = param_at(a.params, a.uniform_rows, 5, N, i);
    p.tau = param_at(a.params, a.uniform_rows, 6, N, i);
    p.conc_pi = param_at(a.params, a.uniform_rows, 7, N, i);
    p.alpha = param_at(a.params, a.uniform_rows, 8, N, i);
    p.erf_scale = param_at(a.params, a.uniform_rows, 9, N, i) / kLn2;
    const ConstDiv dcs = make_const_div(p.cs), dcd = make_const_div(p.cd);
    p.rcs = dcs.r;
    p.rcd = dcd.r;
    const int32_t acc0 = (dcs.ok && dcd.ok) ? (int32_t)0x80000000 : 0;
    const int32_t scen = a.scen ? a.scen[i] : 0;
    const double* eglob = a.emissions + (size_t)scen * a.n_times;
    const int32_t el0 = scen * len - a.step_begin;
    auto emissions_at = [&](int32_t n) -> double {
        if constexpr (LDS) return lds_emis[el0 + n];
        else return eglob[n];
    };

    const size_t r0 = (size_t)a.step_begin * N + i;
    CPState y = {a.ts[r0], a.td[r0], a.conc[r0], a.cum_uptake[r0], a.cum_emis[r0]};
    size_t r = r0 + (size_t)N;
    const int32_t last = a.step_end - 1;
    double emis_next = emissions_at(a.step_begin);
    int32_t mc_next = a.nsub_cc[a.step_begin], mt_next = a.nsub_tl[a.step_begin];

    for (int32_t n = a.step_begin; n < a.step_end; ++n) {
        const double emis = emis_next;
        const int32_t mc = mc_next, mt = mt_next;
        const int32_t np = n < last ? n + 1 : n;
        emis_next = emissions_at(np);
        mc_next = a.nsub_cc[np];
        mt_next = a.nsub_tl[np];
        const CPState y0 = y;
        double erf_co2, erf;
        const int32_t acc = year<true>(p, emis, mc, mt, a.h_cc, a.h_tl, y, erf_co2, erf, acc0);
        // Ts = NaN at the start of the year makes lifetime NaN and every TwoLayer numerator NaN:
        // all outputs of the year are NaN on either path (cumulative emissions involve no
        // division at all), so no replay is needed however the tags read.
        const bool settled = y0.ts != y0.ts;
        if (__builtin_expect(acc >= 0 && !settled, 0)) {
            y = y0;
            year<false>(p, emis, mc, mt, a.h_cc, a.h_tl, y, erf_co2, erf, 0);
        }
        a.conc[r] = y.conc;
        a.cum_uptake[r] = y.cum_u;
        a.cum_emis[r] = y.cum_e;
        a.erf_co2[r] = erf_co2;
        a.erf_total[r] = erf;
        a.ts[r] = y.ts;
        a.td[r] = y.td;
        r += (size_t)N;
    }
    a.status[i] = (is_finite(y.ts) && is_finite(y.td) && is_finite(y.conc) && is_finite(y.cum_u) &&
                   is_finite(y.cum_e)) ? 0 : 1;
}

}  // namespace

hipError_t launch_coupled(const CoupledArgs& a, int mode, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    const size_t lds = a.lds_forcing ? (size_t)a.n_scen * (a.step_end - a.step_begin) * sizeof(double) : 0;
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    void (*kern)(CoupledArgs) = mode != 0 ? (a.lds_forcing ? coupled_fast_kernel<true> : coupled_fast_kernel<false>)
                                          : (a.lds_forcing ? coupled_kernel<true> : coupled_kernel<false>);
    if (lds > (size_t)kMaxStaticLds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

}  // namespace rscm
