// ClimateUDEB for ANY number of ocean layers: the completeness path beside the unrolled kernels of udeb_body.hpp.
//
// The reference takes every n_layers >= 2 (crates/rscm-magicc/src/parameters/climate_udeb.rs:41, validated at
// climate/udeb/mod.rs:162-165).  The fast kernels keep a member's column in registers and unroll the Thomas sweep, which
// needs the layer count at compile time (20 / 30 / 40 / 50 are instantiated).  Every other count runs here: one thread per
// member, the two columns where they live anyway -- the ensemble's internal state in HBM, [hemisphere][layer][N], a
// coalesced row per layer -- updated in place, the sweep's c' array in a work buffer of the same layout, the geometry
// table [NL][6] in device memory (wave-uniform scalar loads), plain loops over the layers.  The arithmetic of a row, of the
// scalar model code between the solves and of the end-of-year diagnostics is that of udeb_body.hpp, statement for
// statement (ocean_column.rs step_hemisphere, mod.rs:399-656): same 1e-9 bar against the CPU restatement, which takes any
// count (tests/test_gpu_udeb.py).  It is several times slower per layer than the unrolled kernels (every row is a trip
// to L2) and not meant to be fast.
#pragma once

#include "udeb_body.hpp"

namespace rscm {
namespace udeb {

// One implicit sub-step of one hemisphere's column, in place.  T: this member's column (layer stride N), holds d' between the
// sweeps; ncp: -c' (layer stride N).  Returns the new mixed-layer temperature.
template <bool FAST>
__device__ __forceinline__ double step_hemisphere_any(const UdebP& p, const YearGeom& y, const double* __restrict__ tab, int32_t NL,
                                                      int32_t land_hc, double* T, double* ncp, size_t N, int hemi, double forcing,
                                                      double hemi_hx, double ground_temp, double land_temp, double alpha_eff, double w)
{
    const bool sh = hemi != 0;
    const double t_top = T[0];
    const double t_bottom = T[(size_t)(NL - 1) * N];
    const double kslope = y.kdC * (t_top - t_bottom);
    const double kslope2 = y.kdC2 * (t_top - t_bottom);
    const double delta_w = w - p.w0;
    const double dwv = fabs(delta_w) > 1e-15 ? delta_w : 0.0;
    const double tul = w * y.dt_dz;
    const double s_afd = p.pi_ratio * tul * t_top;
    const double dwq = y.dt_dz * dwv;
    double tdu, ncp_prev, dp_prev;
    {   // ---- row 0 (mixed layer)
        const double af_top = tab[0], af_bot = tab[1], omr = tab[3], G = sh ? tab[5] : tab[4];
        const double kap0 = fmax(__builtin_fma(omr, kslope, y.kC), y.kminC);
        const double term_diff = kap0 * y.dt_dzmixdz1;
        const double term_upwell = w * y.dt_dzmix;
        const double tf = alpha_eff * (sh ? y.fb[1] : y.fb[0]);
        const double b0 = __builtin_fma(tf, af_top, __builtin_fma(__builtin_fma(term_upwell, p.pi_ratio, term_diff), af_bot, 1.0));
        const double nc0 = (term_diff + term_upwell) * af_bot;
        const double q = __builtin_fma(forcing, sh ? y.famp[1] : y.famp[0], hemi_hx) * y.dt_cmix;
        double d0 = __builtin_fma(q, af_top, t_top);
        if (land_hc) d0 = __builtin_fma(-(land_temp - ground_temp) * (sh ? y.lhc[1] : y.lhc[0]), af_top, d0);
        d0 = __builtin_fma(y.dt_dzmix * dwv, G, d0);
        const double rr = refined_rcp(b0);
        ncp_prev = nc0 * rr;
        dp_prev = d0 * rr;
        ncp[0] = ncp_prev;
        T[0] = dp_prev;
        // row 1 as an interior row: dz_up = dz/2; as the BOTTOM row (two layers) the reference takes dz for it (ocean_column.rs:191)
        tdu = kap0 * (NL == 2 ? y.dt_dz2 : y.dt_dzdz1);
    }
    // Row i + 1 -- its temperature and its six table values -- is requested while row i is computed: the sweep is a chain of trips
    // to L2 and through the scalar cache otherwise (one wavefront per SIMD at 65 536 members: nothing else hides them).
    double t_ahead = T[N];
    double r_ahead[kTabCols];
#pragma unroll
    for (int k = 0; k < kTabCols; ++k) r_ahead[k] = tab[kTabCols + k];
    for (int32_t i = 1; i < NL; ++i) {   // ---- interior rows and the bottom row: forward sweep
        const double af_top = r_ahead[0], af_bot = r_ahead[1], af_diff = r_ahead[2], omr = r_ahead[3], G = sh ? r_ahead[5] : r_ahead[4];
        const double t_i = t_ahead;
        if (i + 1 < NL) {
            t_ahead = T[(size_t)(i + 1) * N];
            const double* nxt = tab + (size_t)(i + 1) * kTabCols;
#pragma unroll
            for (int k = 0; k < kTabCols; ++k) r_ahead[k] = nxt[k];
        }
        // one formula for every row below the mixed layer: the bottom row's table entries say af_bot = 0, af_diff = af_top
        // (udeb_tables.hpp), as in the unrolled kernels -- the same bits
        const double tdu_aft = tdu * af_top;
        const double tdd = fmax(__builtin_fma(omr, kslope2, y.kC2), y.kminC2);
        const double bi = __builtin_fma(tdu + tul, af_top, __builtin_fma(tdd, af_bot, 1.0));
        const double di = __builtin_fma(dwq, G, __builtin_fma(s_afd, af_diff, t_i));
        const double denom = __builtin_fma(-tdu_aft, ncp_prev, bi);
        const double r0 = __builtin_amdgcn_rcp(denom);
        const double e = __builtin_fma(-denom, r0, 1.0);
        const double u = FAST ? e : __builtin_fma(e, e, e);
        const double t = (tdd + tul) * af_bot * r0;
        ncp_prev = __builtin_fma(t, u, t);
        const double sdp = __builtin_fma(tdu_aft, dp_prev, di) * r0;
        dp_prev = __builtin_fma(sdp, u, sdp);
        ncp[(size_t)i * N] = ncp_prev;
        tdu = tdd;
        T[(size_t)i * N] = dp_prev;
    }
    // ---- back substitution, clamp (thomas_solve returns the unclamped vector; the state keeps min(x, max_temp))
    double x = dp_prev;
    T[(size_t)(NL - 1) * N] = fmin(x, p.max_temp);
    double c_ahead = ncp[(size_t)(NL - 2) * N], d_ahead = T[(size_t)(NL - 2) * N];   // (NL >= 2)
    for (int32_t i = NL - 2; i >= 0; --i) {
        const double c_i = c_ahead, d_i = d_ahead;
        if (i > 0) {
            c_ahead = ncp[(size_t)(i - 1) * N];
            d_ahead = T[(size_t)(i - 1) * N];
        }
        x = __builtin_fma(c_i, x, d_i);
        T[(size_t)i * N] = fmin(x, p.max_temp);
    }
    return fmin(x, p.max_temp);
}

// Member i over the model steps [a.step_begin, a.step_end): what Udeb1 (udeb_body.hpp) does with its columns in registers and
// LDS, here with the columns in a.ocean throughout.
template <bool FAST>
__device__ __forceinline__ void udeb_any_member(const UdebArgs& a, int64_t i)
{
    const int64_t N = a.row_stride;
    const int32_t NL = a.n_layers;
    UdebP p;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
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
    // ---- construction: from_parameters (mod.rs:161-227)
    int32_t status = 0;
    if (!is_finite(p.prescribed_eff) || p.prescribed_eff <= 0.0) status = 2;
    LamResult base = LamResult{0.0, 0.0, 1.0, false};
    if (status == 0) {
        base = base_lamcalc_from_block(a.derived, a.derived_uniform, N, i);
        if (!base.ok) status = 4;
    }
    a.status[i] = (uint8_t)status;
    double* st[4] = {a.st0, a.st1, a.st2, a.st3};
    if (status != 0) {  // the reference refuses to build this component: every output NaN
        const double nan = __builtin_nan("");
        for (int32_t n = a.step_begin; n < a.step_end; ++n) {
            const size_t r1 = (size_t)(n + 1) * N + i;
            a.st0[r1] = nan; a.st1[r1] = nan; a.st2[r1] = nan; a.st3[r1] = nan;
            a.heat_uptake[r1] = nan; a.ohc[r1] = nan; a.sst[r1] = nan;
        }
        return;
    }
    double* T_nh = a.ocean + i;
    double* T_sh = a.ocean + (size_t)NL * N + i;
    double* ncp = a.work + i;
    double up_nh = p.w0, up_sh = p.w0, land_nh = 0.0, land_sh = 0.0, gr_nh = 0.0, gr_sh = 0.0, hx_nh = 0.0, hx_sh = 0.0;
    double ae_nh = p.alpha, ae_sh = p.alpha, win_sum = 0.0, hist_last = 0.0;
    int32_t win_lo = 0;
    if (a.step_begin == 0) {   // ClimateUDEBState::new
        for (int32_t l = 0; l < NL; ++l) {
            T_nh[(size_t)l * N] = 0.0;
            T_sh[(size_t)l * N] = 0.0;
        }
    } else {   // resume
        const double* s = a.scal + i;
        up_nh = s[0 * N]; up_sh = s[1 * N]; land_nh = s[2 * N]; land_sh = s[3 * N];
        gr_nh = s[4 * N]; gr_sh = s[5 * N]; ae_nh = s[6 * N]; ae_sh = s[7 * N];
        hx_nh = s[8 * N]; hx_sh = s[9 * N];
        win_sum = s[10 * N];
        win_lo = a.step_begin > 1 ? a.win_kfull[a.step_begin - 1] : 0;
        hist_last = a.hist[(size_t)(a.step_begin - 1) * N + i];
    }
    const int32_t scen = a.scen ? a.scen[i] : 0;
    const double* F = a.link ? a.link + i : a.erf + (size_t)scen * a.n_times;
    const size_t f_stride = a.link ? (size_t)N : (size_t)1;
    const double steps = (double)a.steps_per_year, inv_steps = 1.0 / steps;
    const double c_ground = a.land_hc ? heat_capacity_per_unit_area(p.land_hc_thick) : 0.0;
    const double c_mix = heat_capacity_per_unit_area(p.dz_mix);
    const AirMap airmap = make_air_map(p);
    const double hxf_nh = p.fgno > 1e-15 ? p.k_ns / p.fgno : 0.0, hxf_sh = p.fgso > 1e-15 ? p.k_ns / p.fgso : 0.0;
    const double inv_thresh_nh = 1.0 / p.t_thresh_nh, inv_thresh_sh = 1.0 / p.t_thresh_sh;
    const double ka = p.k_lo * p.amplify, w_min = p.w0 * (1.0 - p.f_var);
    const double* tables = a.tables_dev;

    for (int32_t n = a.step_begin; n < a.step_end; ++n) {
        const size_t r0 = (size_t)n * N + i, r1 = r0 + (size_t)N;
        const double erf_start = F[(size_t)n * f_stride], erf_end = F[(size_t)(n + 1) * f_stride];
        const double bound_lo = a.bounds[n], bound_hi = a.bounds[n + 1];
        const int32_t k_full = a.win_kfull[n];
        const double part_w = a.win_partw[n];
        {   // warm start (mod.rs:436-446)
            const double prev0 = st[0][r0];
            if (T_nh[0] == 0.0 && prev0 != 0.0) {
                T_nh[0] = prev0;
                T_sh[0] = st[2][r0];
                land_nh = st[1][r0];
                land_sh = st[3][r0];
                gr_nh = land_nh;
                gr_sh = land_sh;
            }
        }
        const double dt_year = bound_hi - bound_lo;
        const double dt_sub = dt_year / steps;
        // ---- time-varying ECS (adjusted_ecs) and the LAMCALC re-solve
        const double erf_mid = (erf_start + erf_end) / 2.0;
        double cum_t = 0.0;
        if (n > 0) {
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
        int eff_mode = 0;  // apply_efficacy_and_qfrac
        if (a.efficacy_apply == 1) { eff_mode = 1; }
        else if (a.efficacy_apply == 2 && is_finite(co2_eff) && co2_eff > 0.0) { eff_mode = 2; }
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
            for (int h = 0; h < 2; ++h) {
                const double f_l = (h == 0 ? p.nh_land : p.sh_land) / 2.0;
                const double f_o = 0.5 - f_l;
                const double den = f_o * (p.k_lo + f_l * lam_l);
                y.fb[h] = (lam_o + lam_l * p.k_lo * p.amplify * f_l / den) * y.dt_cmix;
                y.famp[h] = 1.0 + p.k_lo * f_l / den;
                y.lhc[h] = a.land_hc ? p.k_lg * dt_sub / (c_mix * f_o) : 0.0;
            }
        }
        const double eff_scale = eff_mode == 1 ? p.prescribed_eff : eff_mode == 2 ? p.prescribed_eff / co2_eff : 1.0;
        const double r_land_nh = 1.0 / (lam_l * p.fgnl + p.k_lo), r_land_sh = 1.0 / (lam_l * p.fgsl + p.k_lo);
        const double gfac_nh = (a.land_hc && !(p.fgnl < 1e-15)) ? p.k_lg / (p.fgnl * c_ground) * dt_sub : 0.0;
        const double gfac_sh = (a.land_hc && !(p.fgsl < 1e-15)) ? p.k_lg / (p.fgsl * c_ground) * dt_sub : 0.0;
        for (int32_t step_idx = 1; step_idx <= a.steps_per_year; ++step_idx) {
            const double adj = substep_forcing(erf_start, erf_end, step_idx, inv_steps, eff_scale);
            const double f0 = adj * p.q0, f1 = adj * p.q1, f2 = adj * p.q2, f3 = adj * p.q3;
            if (a.land_hc) {
                gr_nh = __builtin_fma(land_nh - gr_nh, gfac_nh, gr_nh);
                gr_sh = __builtin_fma(land_sh - gr_sh, gfac_sh, gr_sh);
            }
            const double sst_nh = step_hemisphere_any<FAST>(p, y, tables, NL, a.land_hc, T_nh, ncp, (size_t)N, 0, f0, hx_nh, gr_nh, land_nh, ae_nh_y, up_nh);
            const double sst_sh = step_hemisphere_any<FAST>(p, y, tables, NL, a.land_hc, T_sh, ncp, (size_t)N, 1, f2, hx_sh, gr_sh, land_sh, ae_sh_y, up_sh);
            const double t_air_nho = sst_to_air(airmap, sst_nh), t_air_sho = sst_to_air(airmap, sst_sh);
            land_nh = land_temperature(ka, p.max_temp, t_air_nho, f1, p.fgnl, r_land_nh);
            land_sh = land_temperature(ka, p.max_temp, t_air_sho, f3, p.fgsl, r_land_sh);
            if (p.fgno > 1e-15) hx_nh = hxf_nh * (t_air_sho - t_air_nho);
            if (p.fgso > 1e-15) hx_sh = hxf_sh * (t_air_nho - t_air_sho);
            const double global_temp = t_air_nho * p.fgno + land_nh * p.fgnl + t_air_sho * p.fgso + land_sh * p.fgsl;
            // update_upwelling
            up_nh = fmax(p.w0 * (1.0 - p.f_var * fmin(global_temp * inv_thresh_nh, 1.0)), w_min);
            up_sh = fmax(p.w0 * (1.0 - p.f_var * fmin(global_temp * inv_thresh_sh, 1.0)), w_min);
        }
        // ---- end of year
        const double sst_nh = T_nh[0], sst_sh = T_sh[0];
        const double air_nh = sst_to_air(airmap, sst_nh), air_sh = sst_to_air(airmap, sst_sh);
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
            for (int32_t l = 1; l < NL; ++l) total += rho_c * p.dz * T_nh[(size_t)l * N];
            total += rho_c * p.dz_mix * sst_sh;
            for (int32_t l = 1; l < NL; ++l) total += rho_c * p.dz * T_sh[(size_t)l * N];
            a.ohc[r1] = total / 2.0;
        }
        a.st0[r1] = air_nh;
        a.st1[r1] = land_nh;
        a.st2[r1] = air_sh;
        a.st3[r1] = land_sh;
        a.sst[r1] = (sst_nh + sst_sh) / 2.0;
    }
    // the scalars go back to HBM once per launch (the columns never left it)
    double* s = a.scal + i;
    s[0 * N] = up_nh; s[1 * N] = up_sh; s[2 * N] = land_nh; s[3 * N] = land_sh;
    s[4 * N] = gr_nh; s[5 * N] = gr_sh; s[6 * N] = ae_nh; s[7 * N] = ae_sh;
    s[8 * N] = hx_nh; s[9 * N] = hx_sh;
    s[10 * N] = win_sum;
}

}  // namespace udeb
}  // namespace rscm
