// Device code of GhgForcing, shared by ghg.hip and group.hip (see ghg.hip).
#pragma once

#include "rk4_device.hpp"
#include "rscm_device.hpp"

namespace rscm {
namespace ghg {

// forcing/ghg.rs:119-129 with the products of powers split into scenario and member parts:
// m75*n75 = (M N)^0.75, m_m152*n152 = M (M N)^1.52
__device__ __forceinline__ double overlap_split(double m75, double n75, double m_m152, double n152)
{
    return 0.47 * log_f64(1.0 + 2.01e-5 * (m75 * n75) + 5.31e-15 * (m_m152 * n152));
}

// x^0.75 and x^1.52 of one x > 0 from ONE logarithm: exp(y ln x) with |y ln x| <= ~12 keeps ~1.5e-15 relative accuracy at a third
// of the instructions of two general pow() calls (the pointwise kinds and the chemistry do the same, pointwise_body.hpp).
__device__ __forceinline__ void powers_75_152(double x, double& p75, double& p152)
{
    const double lx = log_f64(x);
    p75 = exp(0.75 * lx);
    p152 = exp(1.52 * lx);
}

// What depends on the parameters alone, formed once per parameter set (launch_ghg_derive) and read back by the body:
//   [0] ln C0   [1] sqrt M0   [2] sqrt N0
//   IPCCTAR: [3] N0^0.75  [4] N0^1.52  [5] M0^0.75  [6] M0 M0^1.52  [7] overlap(M0, N0)
//   OLBL:    [3] C_alpha_max = C0 - b1/(2 a1)   [4] alpha at and beyond it = d1 - b1^2/(4 a1)
template <int METHOD>
__device__ __forceinline__ void member_constants(double co2_pi, double ch4_pi, double n2o_pi, double a1, double b1, double d1, double (&d)[kDerivedRows])
{
    d[0] = log_f64(co2_pi);
    d[1] = sqrt(ch4_pi);
    d[2] = sqrt(n2o_pi);
    d[3] = d[4] = d[5] = d[6] = d[7] = 0.0;
    if (METHOD == 0) {
        double m152;
        powers_75_152(n2o_pi, d[3], d[4]);
        powers_75_152(ch4_pi, d[5], m152);
        d[6] = ch4_pi * m152;
        d[7] = overlap_split(d[5], d[3], d[6], d[4]);
    } else {
        d[3] = co2_pi - b1 / (2.0 * a1);
        d[4] = -b1 * b1 / (4.0 * a1) + d1;
    }
}

// LINKED: concentrations come per member from other ensembles' series (rscm_ens_link_input) mixed
// with rows of the raw scenario block, and the table rows are evaluated on the fly with the device
// math library -- the same factorisation, so both paths agree to the last-place error of sqrt /
// log / pow.
template <int METHOD, bool HAS_SCEN, bool LINKED>
__device__ __forceinline__ void ghg_body(const GhgArgs& a, const double* __restrict__ tables, int64_t i, int32_t step_begin,
                                         int32_t step_end)
{
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    // LINKED: the first step's concentrations first, in flight together with the parameters (rscm_device.hpp, StepRows)
    const MemberInputs<LINKED ? 2 : 0, 3> conc(a.conc, a.scen, a.links, T, N, i);
    StepRows<3> ahead = {};
    if constexpr (LINKED) ahead = rows_at(conc, step_begin);
    double mc[kDerivedRows];   // the member's constants (launch_ghg_derive), one batch of loads beside the parameters
    params_block<kDerivedRows>(a.derived, a.derived_uniform ? ~0ull : 0ull, N, i, mc);
    const double co2_pi = P(1);
    const double adj_co2 = P(18), adj_ch4 = P(19), adj_n2o = P(20);
    const double ln_c0 = mc[0], sq_m0 = mc[1], sq_n0 = mc[2];
    // scenario table of this member: rows [kGhgRows][T]
    // read-only __restrict__ kernel argument: without a scenario map the row addresses are
    // wave-uniform and become scalar loads
    const double* __restrict__ tab = tables + (HAS_SCEN ? (size_t)a.scen[i] : (size_t)0) * kGhgRows * T;
    double live[kGhgRows];  // LINKED: this year's rows
    auto row = [&](int r, int32_t n) -> double {
        if constexpr (LINKED) return live[r];
        else return tab[(size_t)r * T + n];
    };

    double alpha_tar = 0.0, radeff_m = 0.0, radeff_n = 0.0, n0_75 = 0.0, n0_152 = 0.0, m0_75 = 0.0, m0_m152 = 0.0, ovl00 = 0.0;
    double a1 = 0.0, b1 = 0.0, c1 = 0.0, d1 = 0.0, c_max = 0.0, alpha_sat = 0.0;
    double a3 = 0.0, b3 = 0.0, d3 = 0.0, a2 = 0.0, b2 = 0.0, c2 = 0.0, d2 = 0.0;
    if (METHOD == 0) {
        alpha_tar = P(4) / 0.693147180559945309417;   // 2.0_f64.ln()
        radeff_m = P(5);
        radeff_n = P(6);
        n0_75 = mc[3];
        n0_152 = mc[4];
        m0_75 = mc[5];
        m0_m152 = mc[6];
        ovl00 = mc[7];
    } else {
        a1 = P(7); b1 = P(8); c1 = P(9); d1 = P(10);
        a3 = P(11); b3 = P(12); d3 = P(13);
        a2 = P(14); b2 = P(15); c2 = P(16); d2 = P(17);
        c_max = mc[3];
        alpha_sat = mc[4];
    }
    for (int32_t n = step_begin; n < step_end; ++n) {
        double f_co2, f_ch4, f_n2o;
        if constexpr (LINKED) {
            const StepRows<3> now = ahead;
            if (n + 1 < step_end) ahead = rows_at(conc, n + 1);
            const double c = now.v[0], m = now.v[1], nn = now.v[2];
            live[kGhgCo2] = c;
            live[kGhgLnCo2] = log_f64(c);
            live[kGhgSqrtCo2] = sqrt(c);
            live[kGhgSqrtCh4] = sqrt(m);
            live[kGhgSqrtN2o] = sqrt(nn);
            if (METHOD == 0) {
                double m152;
                powers_75_152(m, live[kGhgCh4P75], m152);
                live[kGhgCh4TimesP152] = m * m152;
                powers_75_152(nn, live[kGhgN2oP75], live[kGhgN2oP152]);
            }
        }
        const double ln_ratio = row(kGhgLnCo2, n) - ln_c0;
        const double sq_m = row(kGhgSqrtCh4, n), sq_n = row(kGhgSqrtN2o, n);
        if (METHOD == 0) {
            f_co2 = alpha_tar * ln_ratio;
            const double ovl_m = overlap_split(row(kGhgCh4P75, n), n0_75, row(kGhgCh4TimesP152, n), n0_152);
            f_ch4 = radeff_m * (sq_m - sq_m0) - (ovl_m - ovl00);
            const double ovl_n = overlap_split(m0_75, row(kGhgN2oP75, n), m0_m152, row(kGhgN2oP152, n));
            f_n2o = radeff_n * (sq_n - sq_n0) - (ovl_n - ovl00);
        } else {
            const double co2 = row(kGhgCo2, n);
            const double dc = co2 - co2_pi;
            const double n2o_overlap = c1 * sq_n;
            double alpha;
            if (co2 >= c_max) alpha = alpha_sat + n2o_overlap;
            else if (co2 <= co2_pi) alpha = d1 + n2o_overlap;
            else alpha = a1 * dc * dc + b1 * dc + d1 + n2o_overlap;
            f_co2 = alpha * ln_ratio;
            f_ch4 = (a3 * sq_m + b3 * sq_n + d3) * (sq_m - sq_m0);
            f_n2o = (a2 * row(kGhgSqrtCo2, n) + b2 * sq_n + c2 * sq_m + d2) * (sq_n - sq_n0);
        }
        const size_t r = (a.rows > 1 ? (size_t)(n + 1) : (size_t)0) * N + i;
        a.erf_co2[r] = f_co2 * adj_co2;
        a.erf_ch4[r] = f_ch4 * adj_ch4;
        a.erf_n2o[r] = f_n2o * adj_n2o;
    }
    a.status[i] = 0;
}

}  // namespace ghg
}  // namespace rscm
