// ClimateUDEB ensemble kernel for gfx950 (MI355X): rscm-magicc's 4-box upwelling-diffusion
// energy-balance model, one thread per ensemble member.
//
// What it replaces, per model step n (reference file:line):
//   ClimateUDEB::solve_impl            crates/rscm-magicc/src/climate/udeb/mod.rs:399-656
//   adjusted_ecs / LAMCALC re-solve    mod.rs:302-350, climate/lamcalc.rs
//   step_hemisphere (implicit 50-layer column, Thomas solve), update_upwelling, diagnostics
//                                      crates/rscm-magicc/src/climate/udeb/ocean_column.rs
//   thomas_solve, invert_4x4           crates/rscm-core/src/utils/linear_algebra.rs
// around the stepper conventions of crates/rscm-core/src/model/runtime.rs (ERF exogenous:
// at_start = F[n], at_end = F[n+1]; outputs written at n+1).
//
// A different kernel class from two_layer.hip: ~110 doubles of private state per member and a
// serial tridiagonal recurrence per hemisphere and sub-step.  Layout and placement:
//   * the two 50-layer columns stay on chip for the whole launch: the hemisphere being solved is
//     in registers (shared with the Thomas d' array), the other one is parked in the lane's LDS
//     slots (25.6 KB per wavefront) and the two are exchanged after every solve.  HBM sees the
//     columns once at the start (resume) and once at the end: ocean[hemi][layer][N], member
//     fastest.  Keeping them in HBM cost 19.2 KB of traffic per member-year and left the single
//     resident wavefront per SIMD waiting on it (profiles/r1_udeb_5e4.txt, the earlier layout);
//   * the Thomas work array c' (50 doubles) stays in registers (layer loops fully unrolled, NL is
//     a template constant); one copy of the solver serves both hemispheres;
//   * everything that depends only on the ocean geometry (area factors, 1 - relative depth, the
//     profile-advection weights built from the CMIP5 initial profiles) is a uniform table in the
//     kernel-argument segment, read through the scalar cache;
//   * the scalar state values live in registers across the whole launch and go to
//     `scal[kUdebScalars][N]` only at the end (resume); the temperature history for the
//     time-varying ECS is hist[T][N], consumed through a running window sum.
// Tolerance parity: quotients are products with refined reciprocals, sums of products are fused,
// the row coefficients are regrouped (step_hemisphere) and LAMCALC solves its 4x4 system by
// structured elimination.  Measured deviation from the CPU oracle over a 1024-member, 400-year
// ensemble: 5e-14 relative (scripts/udeb_deviation.py); tests/test_gpu_udeb.py states 1e-9.
// LAMCALC's convergence test and the |adjusted ECS - ECS| > 1e-10 switch are discontinuous in
// their inputs, so a member sitting within rounding of one of those thresholds may take the
// other branch than the reference for that year (an O(1e-3) relative change of lambda for one
// year); no such member occurs in the test ensembles.
#include <cstdlib>

#include "udeb_body.hpp"

namespace rscm {

namespace {

using namespace udeb;

// One implicit sub-step of one hemisphere's column (ocean_column.rs step_hemisphere).
// dp[] holds this member's column on entry (registers) and the new column on return; slot i
// holds d'[i] in between, so the column and the d' array share registers.  Returns the new
// mixed-layer temperature.
//
// Row algebra relative to the reference (same tridiagonal system, regrouped):
//   kappa_l    = max(omr[l]*(dkdt*C*(T0-Tbottom)) + kappa*C, kappa_min*C)
//   b_i        = 1 + (tdu + tul)*af_top[i] + tdd*af_bot[i]
//   d_i        = T_i + (pi*tul*T0)*af_diff[i] + (dt/dz*dw)*G[i]
//   G[i]       = init[i+1]*af_bot[i] - init[i]*af_top[i] + T_polar*af_diff[i]   (host table)
// and the Thomas recurrences with one refined reciprocal per row; c' is kept negated.
template <int NL>
__device__ __forceinline__ double step_hemisphere(const UdebP& p, const YearGeom& y,
                                                  const double* tables, int32_t land_hc,
                                                  double (&dp)[NL], int hemi,
                                                  double forcing, double hemi_hx, double ground_temp,
                                                  double land_temp, double alpha_eff, double w)
{
    const double* af_top = tables;            // [NL]
    const double* af_bot = tables + NL;       // [NL]
    const double* af_diff = tables + 2 * NL;  // [NL]
    const double* omr = tables + 3 * NL;      // 1 - relative depth, [NL-1]
    const double* G = tables + 4 * NL + (size_t)hemi * NL;  // profile-advection weights
    const bool sh = hemi != 0;
    const double t_top = dp[0];
    const double kslope = y.kdC * (t_top - dp[NL - 1]);
    auto kappa_at = [&](int l) -> double { return fmax(__builtin_fma(omr[l], kslope, y.kC), y.kminC); };
    // kappa_l * dt/dz^2 with the (positive) factor folded into the three constants: one multiply
    // less per interior row, the same value to rounding
    const double kslope2 = y.kdC2 * (t_top - dp[NL - 1]);
    auto tdd_at = [&](int l) -> double { return fmax(__builtin_fma(omr[l], kslope2, y.kC2), y.kminC2); };
    const double delta_w = w - p.w0;
    // |delta_w| <= 1e-15: the reference skips the profile-advection terms; adding exact zeros is
    // the same thing without a branch per row
    const double dwv = fabs(delta_w) > 1e-15 ? delta_w : 0.0;

    double ncp[NL];  // -c'
    const double kap0 = kappa_at(0);
    {   // ---- row 0 (mixed layer)
        const double term_diff = kap0 * y.dt_dzmixdz1;
        const double term_upwell = w * y.dt_dzmix;
        const double tf = alpha_eff * (sh ? y.fb[1] : y.fb[0]);
        const double b0 = __builtin_fma(tf, af_top[0],
                                        __builtin_fma(__builtin_fma(term_upwell, p.pi_ratio, term_diff), af_bot[0], 1.0));
        const double nc0 = (term_diff + term_upwell) * af_bot[0];
        const double q = __builtin_fma(forcing, sh ? y.famp[1] : y.famp[0], hemi_hx) * y.dt_cmix;
        double d0 = __builtin_fma(q, af_top[0], t_top);
        if (land_hc) d0 = __builtin_fma(-(land_temp - ground_temp) * (sh ? y.lhc[1] : y.lhc[0]), af_top[0], d0);
        d0 = __builtin_fma(y.dt_dzmix * dwv, G[0], d0);
        const double r = refined_rcp(b0);
        ncp[0] = nc0 * r;
        dp[0] = d0 * r;
    }
    // ---- interior rows and the bottom row: forward sweep
    const double tul = w * y.dt_dz;
    const double s_afd = p.pi_ratio * tul * t_top;
    const double dwq = y.dt_dz * dwv;
    double tdu = kap0 * y.dt_dzdz1;  // row 1: dz_up = dz/2
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        const double t_i = dp[i];
        const double tdu_aft = tdu * af_top[i];
        double bi, di;
        if (i < NL - 1) {
            const double tdd = tdd_at(i);
            bi = __builtin_fma(tdu + tul, af_top[i], __builtin_fma(tdd, af_bot[i], 1.0));
            di = __builtin_fma(dwq, G[i], __builtin_fma(s_afd, af_diff[i], t_i));
            const double denom = __builtin_fma(-tdu_aft, ncp[i - 1], bi);
            // 1/denom = r0 (1 + e + e^2 + ...), e = 1 - denom*r0: the hardware estimate is good
            // to ~2^-23, so the series cut after e^2 is exact to rounding, and the c' chain that
            // feeds the next row's denominator is five dependent operations instead of seven
            const double r0 = __builtin_amdgcn_rcp(denom);
            const double e = __builtin_fma(-denom, r0, 1.0);
            const double u = __builtin_fma(e, e, e);
            const double t = (tdd + tul) * af_bot[i] * r0;
            ncp[i] = __builtin_fma(t, u, t);
            const double sdp = __builtin_fma(tdu_aft, dp[i - 1], di) * r0;
            dp[i] = __builtin_fma(sdp, u, sdp);
            tdu = tdd;
        } else {
            bi = __builtin_fma(tdu + tul, af_top[i], 1.0);
            di = __builtin_fma(dwq, G[i], __builtin_fma(s_afd, af_top[i], t_i));
            const double denom = __builtin_fma(-tdu_aft, ncp[i - 1], bi);
            dp[i] = __builtin_fma(tdu_aft, dp[i - 1], di) * refined_rcp(denom);
        }
    }
    // ---- back substitution, clamp.  thomas_solve returns the unclamped vector; the state keeps
    // min(x, max_temp)
    double x = dp[NL - 1];
    dp[NL - 1] = fmin(x, p.max_temp);
#pragma unroll
    for (int i = NL - 2; i >= 0; --i) {
        x = __builtin_fma(ncp[i], x, dp[i]);
        dp[i] = fmin(x, p.max_temp);
    }
    return dp[0];
}

template <int NL>
__global__ __launch_bounds__(kUdebBlock) void udeb_kernel(UdebArgs a)
{
    // The two 50-layer columns never leave the chip during a launch: the active hemisphere is in
    // registers (col[], shared with the solver's d' array), the other one is parked in this
    // lane's LDS slots and the two are exchanged after every column solve.  25.6 KB of LDS per
    // wavefront, each lane touches only its own slots (no barriers, no bank conflicts:
    // consecutive lanes, consecutive 8-byte words).
    __shared__ double park[NL][kUdebBlock];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kUdebBlock + threadIdx.x;
    if (i >= a.n_members) return;
    const int64_t N = a.n_members;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    UdebP p;
    p.dz_mix = P(1); p.dz = P(2); p.kappa = P(3); p.kappa_min = P(4); p.kappa_dkdt = P(5);
    p.w0 = P(6); p.f_var = P(7); p.t_thresh_nh = P(8); p.t_thresh_sh = P(9);
    p.ecs = P(10); p.rf_2x = P(11); p.rlo = P(12); p.fb_q = P(13); p.fb_cumt = P(14); p.fb_period = P(15);
    p.k_lo = P(16); p.k_ns = P(17); p.amplify = P(18); p.nh_land = P(19); p.sh_land = P(20);
    p.alpha = P(22); p.gamma = P(23); p.pi_ratio = P(24); p.k_lg = P(26); p.land_hc_thick = P(27);
    p.rf0 = P(28); p.rf1 = P(29); p.rf2 = P(30); p.rf3 = P(31); p.prescribed_eff = P(33); p.max_temp = P(36);
    p.fgnl = p.nh_land / 2.0; p.fgno = 0.5 - p.fgnl; p.fgsl = p.sh_land / 2.0; p.fgso = 0.5 - p.fgsl;
    {   // compute_qfrac
        const double rf_sum = p.rf0 * p.fgno + p.rf1 * p.fgnl + p.rf2 * p.fgso + p.rf3 * p.fgsl;
        if (fabs(rf_sum) <= 1e-15) { p.q0 = p.q1 = p.q2 = p.q3 = 1.0; }
        else { p.q0 = p.rf0 / rf_sum; p.q1 = p.rf1 / rf_sum; p.q2 = p.rf2 / rf_sum; p.q3 = p.rf3 / rf_sum; }
    }
    double* st[4] = {a.st0, a.st1, a.st2, a.st3};
    double* T_nh = a.ocean + i;
    double* T_sh = a.ocean + (size_t)NL * N + i;

    // ---- construction: from_parameters (mod.rs:161-227)
    int32_t status = 0;
    if (!is_finite(p.prescribed_eff) || p.prescribed_eff <= 0.0) status = 2;
    LamResult base = {0.0, 0.0, 1.0, false};
    if (status == 0) {
        base = lamcalc(p, p.ecs);
        if (!base.ok) status = 4;
    }
    a.status[i] = (uint8_t)status;
    if (status != 0) {  // the reference refuses to build this component: every output NaN
        for (int32_t n = a.step_begin; n < a.step_end; ++n) {
            const size_t r = (size_t)(n + 1) * N + i;
            const double nan = __builtin_nan("");
            a.st0[r] = nan; a.st1[r] = nan; a.st2[r] = nan; a.st3[r] = nan;
            a.heat_uptake[r] = nan; a.ohc[r] = nan; a.sst[r] = nan;
        }
        return;
    }

    // ---- internal state (ClimateUDEBState::new) or resume
    double up_nh, up_sh, land_nh, land_sh, gr_nh, gr_sh, ae_nh, ae_sh, hx_nh, hx_sh;
    double col[NL];
    // running window sum of the temperature history: entries [win_lo, n-1) after year n-1
    double win_sum = 0.0, hist_last = 0.0;
    int32_t win_lo = 0;
    if (a.step_begin == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            col[l] = 0.0;
            park[l][lane] = 0.0;
        }
        up_nh = up_sh = p.w0;
        land_nh = land_sh = gr_nh = gr_sh = hx_nh = hx_sh = 0.0;
        ae_nh = ae_sh = p.alpha;
    } else {
        const double* s = a.scal + i;
        up_nh = s[0 * N]; up_sh = s[1 * N]; land_nh = s[2 * N]; land_sh = s[3 * N];
        gr_nh = s[4 * N]; gr_sh = s[5 * N]; ae_nh = s[6 * N]; ae_sh = s[7 * N];
        hx_nh = s[8 * N]; hx_sh = s[9 * N];
        win_sum = s[10 * N];
        win_lo = a.step_begin > 1 ? a.win_kfull[a.step_begin - 1] : 0;
        hist_last = a.hist[(size_t)(a.step_begin - 1) * N + i];
        // One wavefront per SIMD: nothing hides a load's latency but the loads that are in flight with it.
        // The southern column first, all 50 loads at once into the registers of col[], from there to its LDS
        // slots; then the northern one (two round trips to HBM; interleaved with the LDS writes, a layer at a
        // time, the compiler waited for every pair: 25 round trips per launch, and the graph launches every step).
#pragma unroll
        for (int l = 0; l < NL; ++l) col[l] = T_sh[(size_t)l * N];
#pragma unroll
        for (int l = 0; l < NL; ++l) park[l][lane] = col[l];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int l = 0; l < NL; ++l) col[l] = T_nh[(size_t)l * N];
    }
    const int32_t scen = a.scen ? a.scen[i] : 0;
    // a linked forcing (rscm_ens_link_input) is another ensemble's [T][N] series
    const double* F = a.link ? a.link + i : a.erf + (size_t)scen * a.n_times;
    const size_t f_stride = a.link ? (size_t)N : (size_t)1;
    const double steps = (double)a.steps_per_year;
    const double c_ground = a.land_hc ? heat_capacity_per_unit_area(p.land_hc_thick) : 0.0;
    const double c_mix = heat_capacity_per_unit_area(p.dz_mix);
    const double* tables = a.tables;  // kernarg segment
    const double* __restrict__ bounds = a.bounds;

    for (int32_t n = a.step_begin; n < a.step_end; ++n) {
        const double erf_start = F[(size_t)n * f_stride], erf_end = F[(size_t)(n + 1) * f_stride];
        const size_t r0 = (size_t)n * N + i, r1 = r0 + (size_t)N;
        // warm start (mod.rs:436-446)
        {
            const double prev0 = st[0][r0];
            if (col[0] == 0.0 && prev0 != 0.0) {
                col[0] = prev0;
                park[0][lane] = st[2][r0];
                land_nh = st[1][r0];
                land_sh = st[3][r0];
                gr_nh = land_nh;
                gr_sh = land_sh;
            }
        }
        const double dt_year = bounds[n + 1] - bounds[n];
        const double dt_sub = dt_year / steps;
        // ---- time-varying ECS (adjusted_ecs) and the LAMCALC re-solve
        const double erf_mid = (erf_start + erf_end) / 2.0;
        double cum_t = 0.0;
        if (n > 0) {
            // The window of adjusted_ecs() depends only on the time axis and the (uniform)
            // feedback_cumt_period: the host walked it once per year (rscm_gpu.cpp): entries
            // [k_full, n) enter whole, entry k_full-1 with weight part_w if part_w > 0.  The whole
            // part is a running sum (last year's entry comes in from a register, the entries the
            // window has moved past are read back and subtracted: 0-2 loads a year instead of a
            // 300-year walk); the reference re-sums newest to oldest, which this matches to
            // rounding.
            const int32_t k_full = a.win_kfull[n];
            const double part_w = a.win_partw[n];
            const double* hcol = a.hist + i;
            win_sum += hist_last;
            for (; win_lo < k_full; ++win_lo) win_sum -= hcol[(size_t)win_lo * N];
            if (p.fb_cumt != 0.0) {
                cum_t = win_sum;
                if (part_w > 0.0) cum_t += hcol[(size_t)(k_full - 1) * N] * part_w;
            }
        }
        const double cumt_2x = p.ecs * p.fb_period;
        const double cumt_factor = fabs(cumt_2x) > 1e-15 ? 1.0 + p.fb_cumt * (cum_t - cumt_2x) / cumt_2x : 1.0;
        const double q_factor = 1.0 + p.fb_q * (fmax(erf_mid, 0.0) - p.rf_2x);
        const double adj_ecs = p.ecs * cumt_factor * q_factor;
        double lam_o = base.lam_o, lam_l = base.lam_l, co2_eff = base.eff;
        if (fabs(adj_ecs - p.ecs) > 1e-10) {
            const LamResult rr = lamcalc(p, adj_ecs);
            if (rr.ok) {
                lam_o = rr.lam_o;
                lam_l = rr.lam_l;
                co2_eff = rr.eff;
            }
        }
        double eff_scale = 1.0;  // apply_efficacy_and_qfrac
        int eff_mode = 0;
        if (a.efficacy_apply == 1) { eff_mode = 1; }
        else if (a.efficacy_apply == 2 && is_finite(co2_eff) && co2_eff > 0.0) { eff_mode = 2; }
        (void)eff_scale;
        const double ae_nh_y = ae_nh, ae_sh_y = ae_sh;  // alpha_eff is fixed for the year
        YearGeom y;
        {
            const double dz1 = p.dz / 2.0;
            y.dt_dz = dt_sub / p.dz;
            y.dt_dzmix = dt_sub / p.dz_mix;
            y.dt_cmix = dt_sub / c_mix;
            y.dt_dz2 = dt_sub / (p.dz * p.dz);
            y.dt_dzdz1 = dt_sub / (p.dz * dz1);
            y.dt_dzmixdz1 = dt_sub / (p.dz_mix * dz1);
            y.kC = p.kappa * kDiffCm2sToM2yr;
            y.kdC = p.kappa_dkdt * kDiffCm2sToM2yr;
            y.kminC = p.kappa_min * kDiffCm2sToM2yr;
            y.kC2 = y.kC * y.dt_dz2;
            y.kdC2 = y.kdC * y.dt_dz2;
            y.kminC2 = y.kminC * y.dt_dz2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double f_l = (h == 0 ? p.nh_land : p.sh_land) / 2.0;
                const double f_o = 0.5 - f_l;
                const double den = f_o * (p.k_lo + f_l * lam_l);
                y.fb[h] = (lam_o + lam_l * p.k_lo * p.amplify * f_l / den) * y.dt_cmix;
                y.famp[h] = 1.0 + p.k_lo * f_l / den;
                y.lhc[h] = a.land_hc ? p.k_lg * dt_sub / (c_mix * f_o) : 0.0;
            }
        }
        for (int32_t step_idx = 1; step_idx <= a.steps_per_year; ++step_idx) {
            const double frac = (double)step_idx / steps;
            const double erf = erf_start + frac * (erf_end - erf_start);
            double adj = erf;
            if (eff_mode == 1) adj = erf * p.prescribed_eff;
            else if (eff_mode == 2) adj = erf * p.prescribed_eff / co2_eff;
            const double f0 = adj * p.q0, f1 = adj * p.q1, f2 = adj * p.q2, f3 = adj * p.q3;
            if (a.land_hc) {
                if (!(p.fgnl < 1e-15)) gr_nh += p.k_lg * (land_nh - gr_nh) / (p.fgnl * c_ground) * dt_sub;
                if (!(p.fgsl < 1e-15)) gr_sh += p.k_lg * (land_sh - gr_sh) / (p.fgsl * c_ground) * dt_sub;
            }
            // one copy of the column solver, run for NH then SH (uniform selects)
            double sst_pair[2];
#pragma unroll 1
            for (int hemi = 0; hemi < 2; ++hemi) {
                const bool sh = hemi != 0;
                sst_pair[hemi] = step_hemisphere<NL>(p, y, tables, a.land_hc, col, hemi, sh ? f2 : f0,
                                                     sh ? hx_sh : hx_nh, sh ? gr_sh : gr_nh,
                                                     sh ? land_sh : land_nh, sh ? ae_sh_y : ae_nh_y,
                                                     sh ? up_sh : up_nh);
                // exchange the solved column with the parked hemisphere
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double other = park[l][lane];
                    park[l][lane] = col[l];
                    col[l] = other;
                }
            }
            const double sst_nh = sst_pair[0], sst_sh = sst_pair[1];
            const double t_air_nho = sst_to_air(p, sst_nh), t_air_sho = sst_to_air(p, sst_sh);
            land_nh = land_temperature(p, t_air_nho, f1, p.fgnl, lam_l);
            land_sh = land_temperature(p, t_air_sho, f3, p.fgsl, lam_l);
            if (p.fgno > 1e-15) hx_nh = p.k_ns / p.fgno * (t_air_sho - t_air_nho);
            if (p.fgso > 1e-15) hx_sh = p.k_ns / p.fgso * (t_air_nho - t_air_sho);
            const double global_temp = t_air_nho * p.fgno + land_nh * p.fgnl + t_air_sho * p.fgso + land_sh * p.fgsl;
            {   // update_upwelling
                const double w_min = p.w0 * (1.0 - p.f_var);
                up_nh = fmax(p.w0 * (1.0 - p.f_var * fmin(global_temp / p.t_thresh_nh, 1.0)), w_min);
                up_sh = fmax(p.w0 * (1.0 - p.f_var * fmin(global_temp / p.t_thresh_sh, 1.0)), w_min);
            }
        }
        // ---- end of year
        const double sst_nh = col[0], sst_sh = park[0][lane];
        const double air_nh = sst_to_air(p, sst_nh), air_sh = sst_to_air(p, sst_sh);
        ae_nh = fabs(sst_nh) < 1e-15 ? p.alpha : air_nh / sst_nh;
        ae_sh = fabs(sst_sh) < 1e-15 ? p.alpha : air_sh / sst_sh;
        const double global_temp = air_nh * p.fgno + land_nh * p.fgnl + air_sh * p.fgso + land_sh * p.fgsl;
        hist_last = global_temp * dt_year;
        a.hist[r0] = hist_last;
        double adj_end = erf_end;
        if (eff_mode == 1) adj_end = erf_end * p.prescribed_eff;
        else if (eff_mode == 2) adj_end = erf_end * p.prescribed_eff / co2_eff;
        {
            const double w[4] = {p.fgno, p.fgnl, p.fgso, p.fgsl};
            const double lambdas[4] = {lam_o, lam_l, lam_o, lam_l};
            const double fe[4] = {adj_end * p.q0, adj_end * p.q1, adj_end * p.q2, adj_end * p.q3};
            const double tt[4] = {air_nh, land_nh, air_sh, land_sh};
            double q_global = 0.0, feedback_global = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q_global += w[k] * fe[k];
                feedback_global += w[k] * lambdas[k] * tt[k];
            }
            a.heat_uptake[r1] = q_global - feedback_global;
        }
        {   // calculate_ocean_heat_content: hemisphere by hemisphere, layer by layer
            const double rho_c = kRhoSeawater * kCpSeawater;
            double total = 0.0;
            total += rho_c * p.dz_mix * sst_nh;
#pragma unroll
            for (int l = 1; l < NL; ++l) total += rho_c * p.dz * col[l];
            total += rho_c * p.dz_mix * sst_sh;
#pragma unroll 1
            for (int l = 1; l < NL; ++l) total += rho_c * p.dz * park[l][lane];
            a.ohc[r1] = total / 2.0;
        }
        a.st0[r1] = air_nh;
        a.st1[r1] = land_nh;
        a.st2[r1] = air_sh;
        a.st3[r1] = land_sh;
        a.sst[r1] = (sst_nh + sst_sh) / 2.0;
    }
    double* s = a.scal + i;
    s[0 * N] = up_nh; s[1 * N] = up_sh; s[2 * N] = land_nh; s[3 * N] = land_sh;
    s[4 * N] = gr_nh; s[5 * N] = gr_sh; s[6 * N] = ae_nh; s[7 * N] = ae_sh;
    s[8 * N] = hx_nh; s[9 * N] = hx_sh;
    s[10 * N] = win_sum;
    // the columns go back to HBM once per launch (rscm_ens_run resumes from them)
#pragma unroll
    for (int l = 0; l < NL; ++l) T_nh[(size_t)l * N] = col[l];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int l = 0; l < NL; ++l) col[l] = park[l][lane];
#pragma unroll
    for (int l = 0; l < NL; ++l) T_sh[(size_t)l * N] = col[l];
}


// Two wavefronts per 64 members, one hemisphere each (udeb_body.hpp).  WAVES: wavefronts per SIMD the register
// budget is cut for (1: all of c' in registers; 2: KC of its entries in LDS).
template <int NL, int KC, int WAVES>
__global__ __launch_bounds__(kUdeb2Block, WAVES) void udeb2_kernel(UdebArgs a)
{
    __shared__ Udeb2Lds<KC> lds;
    Udeb2<NL, KC> m(lds);
    m.begin(a);
    for (int32_t n = a.step_begin; n < a.step_end; ++n) m.step(a, n);
    m.end(a);
}

}  // namespace

template <int NL>
static void launch_udeb2(const UdebArgs& a, int variant, hipStream_t s)
{
    const dim3 grid((unsigned)((a.n_members + 63) / 64));
    if (variant == 2) hipLaunchKernelGGL((udeb2_kernel<NL, 0, 1>), grid, dim3(kUdeb2Block), 0, s, a);
    else if (variant == 3) hipLaunchKernelGGL((udeb2_kernel<NL, (NL >= 40 ? 24 : NL / 2), 2>), grid, dim3(kUdeb2Block), 0, s, a);
    else hipLaunchKernelGGL((udeb2_kernel<NL, (NL >= 32 ? 16 : NL / 2), 2>), grid, dim3(kUdeb2Block), 0, s, a);
}

hipError_t launch_udeb(const UdebArgs& a, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    // development switch: 0 the one-thread-per-member kernel, 1..3 the two-wavefront variants
    static const int variant = [] { const char* e = getenv("RSCM_UDEB_VARIANT"); return e ? atoi(e) : 1; }();
    if (variant == 0) {
        if (a.n_layers != 50) return hipErrorInvalidValue;
        const dim3 grid((unsigned)((a.n_members + kUdebBlock - 1) / kUdebBlock));
        hipLaunchKernelGGL(udeb_kernel<50>, grid, dim3(kUdebBlock), 0, s, a);
        return hipGetLastError();
    }
    switch (a.n_layers) {   // the column loops are unrolled: one instance per supported layer count
        case 50: launch_udeb2<50>(a, variant, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rscm
