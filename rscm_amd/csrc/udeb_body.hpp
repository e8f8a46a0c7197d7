// ClimateUDEB per-member arithmetic of the kernels in csrc/udeb.hip: rscm-magicc's 4-box upwelling-diffusion
// energy-balance model.
//
// What it replaces, per model step n (reference file:line):
//   ClimateUDEB::solve_impl            crates/rscm-magicc/src/climate/udeb/mod.rs:399-656
//   adjusted_ecs / LAMCALC re-solve    mod.rs:302-350, climate/lamcalc.rs
//   step_hemisphere (implicit column, Thomas solve), update_upwelling, diagnostics
//                                      crates/rscm-magicc/src/climate/udeb/ocean_column.rs
//   thomas_solve, invert_4x4           crates/rscm-core/src/utils/linear_algebra.rs
// around the stepper conventions of crates/rscm-core/src/model/runtime.rs (ERF exogenous:
// at_start = F[n], at_end = F[n+1]; outputs written at n+1).
#pragma once

#include "rk4_device.hpp"
#include "rscm_device.hpp"

namespace rscm {
namespace udeb {

constexpr double kDiffCm2sToM2yr = 3155.76;  // parameters/climate_udeb.rs
constexpr double kRhoSeawater = 1026.0;
constexpr double kCpSeawater = 3985.0;
constexpr double kSecondsPerYear = 31557600.0;

struct UdebP {
    double dz_mix, dz, kappa, kappa_min, kappa_dkdt, w0, f_var, t_thresh_nh, t_thresh_sh;
    double ecs, rf_2x, rlo, fb_q, fb_cumt, fb_period, k_lo, k_ns, amplify, nh_land, sh_land;
    double alpha, gamma, pi_ratio, k_lg, land_hc_thick, rf0, rf1, rf2, rf3, prescribed_eff, max_temp;
    double fgno, fgnl, fgso, fgsl, q0, q1, q2, q3;  // box fractions, co2_qfrac
};

struct LamResult {
    double lam_o, lam_l, eff;
    bool ok;
};

__device__ __forceinline__ double heat_capacity_per_unit_area(double depth_m)
{
    return kRhoSeawater * kCpSeawater * depth_m / kSecondsPerYear;
}

// rscm-core/src/utils/linear_algebra.rs invert_4x4 (Gauss-Jordan, partial pivoting), rows kept in
// registers: every index is static, row swaps are per-lane selects.
__device__ __forceinline__ bool invert_4x4(const double m[4][4], double inv[4][4])
{
    double aug[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) aug[i][j] = j < 4 ? m[i][j] : (j - 4 == i ? 1.0 : 0.0);
    }
    bool ok = true;
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int max_row = col;
        double max_val = fabs(aug[col][col]);
#pragma unroll
        for (int row = col + 1; row < 4; ++row) {
            const double val = fabs(aug[row][col]);
            if (val > max_val) {
                max_val = val;
                max_row = row;
            }
        }
        if (max_val < 1e-15) ok = false;
#pragma unroll
        for (int row = col + 1; row < 4; ++row) {
            const bool sw = max_row == row;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double x = aug[col][j], y = aug[row][j];
                aug[col][j] = sw ? y : x;
                aug[row][j] = sw ? x : y;
            }
        }
        // aug[col][j] /= pivot for j = 0..7: one refined reciprocal, then the 3-instruction
        // quotient (identical to IEEE division for these O(1) operands; rk4_device.hpp)
        const double pivot = aug[col][col];
        const double rp = refined_rcp(pivot);
#pragma unroll
        for (int j = 0; j < 8; ++j) aug[col][j] = spec_div(aug[col][j], pivot, rp);
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            if (row == col) continue;
            const double factor = aug[row][col];
#pragma unroll
            for (int j = 0; j < 8; ++j) aug[row][j] -= factor * aug[col][j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) inv[i][j] = aug[i][j + 4];
    return ok;
}

// Equilibrium box temperatures per unit forcing for one (lambda_ocean, lambda_land) pair the way
// the reference forms them (lamcalc.rs: build the 4x4 exchange matrix, invert_4x4, multiply by
// area*qfrac).  Used only for the lanes whose matrix is too close to singular for the direct
// elimination below -- it keeps the reference's partial pivoting and its singularity verdict.
// (Out of line and by value: a reference to the parameter block here would pin all of it in scratch memory for
// the whole kernel -- the column solves would reload their parameters from there.)
struct LamBoxes { double x0, x1, x2, x3; bool ok; };
__device__ __noinline__ LamBoxes lam_box_temps_general(double k_lo, double k_ns, double alpha, double fgno, double fgnl, double fgso, double fgsl,
                                                       double q0, double q1, double q2, double q3, double lam_o, double lam_l)
{
    const double area[4] = {fgno, fgnl, fgso, fgsl};
    const double qfrac[4] = {q0, q1, q2, q3};
    const double m[4][4] = {{fgno * lam_o + k_lo * alpha + k_ns, -k_lo, -k_ns, 0.0},
                            {-k_lo * alpha, fgnl * lam_l + k_lo, 0.0, 0.0},
                            {-k_ns, 0.0, fgso * lam_o + k_lo * alpha + k_ns, -k_lo},
                            {0.0, 0.0, -k_lo * alpha, fgsl * lam_l + k_lo}};
    double inv[4][4];
    LamBoxes out = {0.0, 0.0, 0.0, 0.0, false};
    if (!invert_4x4(m, inv)) return out;
    double x[4];
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        double sum = 0.0;
#pragma unroll
        for (int col = 0; col < 4; ++col) sum += inv[row][col] * area[col] * qfrac[col];
        x[row] = sum;
    }
    out.x0 = x[0]; out.x1 = x[1]; out.x2 = x[2]; out.x3 = x[3];
    out.ok = true;
    return out;
}

// climate/lamcalc.rs lamcalc(): secant-style iteration on lambda_ocean until the land/ocean
// warming ratio matches RLO within 1e-3; only the last three iterates are ever read.
//
// The exchange matrix couples each land box to its own ocean box only,
//     [ A0   -k   -n    0 ]        A0 = fgno*lam_o + k*a + n     k = k_lo, n = k_ns, a = amplify
//     [-k*a   B1   0    0 ]        B1 = fgnl*lam_l + k
//     [ -n    0    A2  -k ]        A2 = fgso*lam_o + k*a + n
//     [  0    0  -k*a   B3]        B3 = fgsl*lam_l + k
// so M x = v is solved by eliminating the two land rows and applying Cramer's rule to the 2x2
// ocean system: ~40 flops and three reciprocals per iterate where the general inverse costs ~600
// instructions.  Tolerance parity like the rest of this kernel; an ill-conditioned elimination
// (small B1, B3 or determinant) takes the reference's pivoted inverse instead.
__device__ __forceinline__ LamResult lamcalc(const UdebP& p, double ecs)
{
    const double q = p.rf_2x, k_lo = p.k_lo, k_ns = p.k_ns;
    const double ka = k_lo * p.amplify, kka = k_lo * ka, nn = k_ns * k_ns, kan = ka + k_ns;
    const double lam = q / ecs;
    const double fgosum = p.fgno + p.fgso, fglsum = p.fgnl + p.fgsl, fratio = fgosum / fglsum;
    const double fr_rlo = fratio / p.rlo;
    const double r_fo = 1.0 / fgosum, r_fl = 1.0 / fglsum;
    const double v0 = p.fgno * p.q0, v1 = p.fgnl * p.q1, v2 = p.fgso * p.q2, v3 = p.fgsl * p.q3;
    const double kv1 = k_lo * v1, kv3 = k_lo * v3;
    // lamo[i-2], lamo[i-1], lamo[i]; diff likewise (arrays start zero-filled in the reference)
    double lamo_m2 = 0.0, lamo_m1 = lam, lamo_i = lam + 0.7;
    double diff_m2 = 0.0, diff_m1 = 0.0;
    double dlamo = 0.7;
    int iflag = 0;
    LamResult out = {0.0, 0.0, 1.0, false};
    for (int i = 2; i <= 40; ++i) {
        const double lam_l = __builtin_fma(fr_rlo, lam - lamo_i, lam);
        const double lam_o = lamo_i;
        const double A0 = __builtin_fma(p.fgno, lam_o, kan), A2 = __builtin_fma(p.fgso, lam_o, kan);
        const double B1 = __builtin_fma(p.fgnl, lam_l, k_lo), B3 = __builtin_fma(p.fgsl, lam_l, k_lo);
        const double r1 = refined_rcp(B1), r3 = refined_rcp(B3);
        const double P = __builtin_fma(-kka, r1, A0), Q = __builtin_fma(-kka, r3, A2);
        const double e0 = __builtin_fma(kv1, r1, v0), e2 = __builtin_fma(kv3, r3, v2);
        const double pq = P * Q;
        const double det = pq - nn;
        double x[4];
        const bool direct = fabs(B1) > 0.05 * k_lo && fabs(B3) > 0.05 * k_lo && fabs(det) > 0.01 * (fabs(pq) + nn);
        if (__builtin_expect(direct, 1)) {
            const double rd = refined_rcp(det);
            x[0] = __builtin_fma(e0, Q, k_ns * e2) * rd;
            x[2] = __builtin_fma(P, e2, k_ns * e0) * rd;
            x[1] = __builtin_fma(ka, x[0], v1) * r1;
            x[3] = __builtin_fma(ka, x[2], v3) * r3;
        } else {
            const LamBoxes g = lam_box_temps_general(k_lo, k_ns, p.amplify, p.fgno, p.fgnl, p.fgso, p.fgsl, p.q0, p.q1, p.q2, p.q3, lam_o, lam_l);
            if (!g.ok) return out;
            x[0] = g.x0; x[1] = g.x1; x[2] = g.x2; x[3] = g.x3;
        }
        const double t0 = q * x[0], t1 = q * x[1], t2 = q * x[2], t3 = q * x[3];
        const double ocean_mean = __builtin_fma(p.fgno, t0, p.fgso * t2) * r_fo;
        const double land_mean = __builtin_fma(p.fgnl, t1, p.fgsl * t3) * r_fl;
        const double diff_i = p.rlo - land_mean / ocean_mean;
        if (fabs(diff_i) < 0.001) {
            out.lam_o = lam_o;
            out.lam_l = lam_l;
            out.ok = true;
            const double rf_sum = p.rf0 * p.fgno + p.rf1 * p.fgnl + p.rf2 * p.fgso + p.rf3 * p.fgsl;
            if (fabs(rf_sum) <= 1e-15) {
                out.eff = 1.0;
            } else {
                const double t_global = p.fgno * t0 + p.fgnl * t1 + p.fgso * t2 + p.fgsl * t3;
                out.eff = t_global / ecs;
            }
            return out;
        }
        if (diff_i * diff_m1 < 0.0) iflag = 1;
        double next;
        if (iflag == 0) {
            if (fabs(diff_i) > fabs(diff_m1)) dlamo = -dlamo;
            next = lamo_i + dlamo;
        } else if (diff_i * diff_m1 < 0.0) {
            const double denom = diff_i - diff_m1;
            next = fabs(denom) < 1e-30 ? lamo_i + dlamo : lamo_i - diff_i * (lamo_i - lamo_m1) / denom;
        } else {
            const double denom = diff_i - diff_m2;
            next = fabs(denom) < 1e-30 ? lamo_i + dlamo : lamo_i - diff_i * (lamo_i - lamo_m2) / denom;
        }
        lamo_m2 = lamo_m1;
        lamo_m1 = lamo_i;
        lamo_i = next;
        diff_m2 = diff_m1;
        diff_m1 = diff_i;
    }
    return out;
}

// The base LAMCALC solve (from_parameters, mod.rs:161-227) depends on the parameters alone: formed once per parameter set by
// launch_udeb_derive (udeb.hip) -- this very function on the same inputs -- and read back at the top of every launch instead of iterating
// again (a one-step launch of a graph paid the secant iteration every model step).  Rows of the block: [0] lambda_ocean
// [1] lambda_land [2] CO2 efficacy [3] 1.0 if the iteration converged.
__device__ __forceinline__ LamResult base_lamcalc_from_block(const double* __restrict__ derived, int32_t uniform, int64_t N, int64_t i)
{
    const uint64_t u = uniform ? 0xFull : 0ull;
    LamResult r;
    r.lam_o = param_at(derived, u, 0, N, i);
    r.lam_l = param_at(derived, u, 1, N, i);
    r.eff = param_at(derived, u, 2, N, i);
    r.ok = param_at(derived, u, 3, N, i) != 0.0;
    return r;
}

// what lamcalc() reads of a member's parameters (the derive kernel fills just this)
__device__ __forceinline__ void fill_lamcalc_inputs(UdebP& p, double ecs, double rf_2x, double rlo, double k_lo, double k_ns, double amplify,
                                                    double nh_land, double sh_land, double rf0, double rf1, double rf2, double rf3)
{
    p.ecs = ecs; p.rf_2x = rf_2x; p.rlo = rlo; p.k_lo = k_lo; p.k_ns = k_ns; p.amplify = amplify; p.nh_land = nh_land; p.sh_land = sh_land;
    p.rf0 = rf0; p.rf1 = rf1; p.rf2 = rf2; p.rf3 = rf3;
    p.fgnl = p.nh_land / 2.0; p.fgno = 0.5 - p.fgnl; p.fgsl = p.sh_land / 2.0; p.fgso = 0.5 - p.fgsl;
    const double rf_sum = p.rf0 * p.fgno + p.rf1 * p.fgnl + p.rf2 * p.fgso + p.rf3 * p.fgsl;   // compute_qfrac
    if (fabs(rf_sum) <= 1e-15) { p.q0 = p.q1 = p.q2 = p.q3 = 1.0; }
    else { p.q0 = p.rf0 / rf_sum; p.q1 = p.rf1 / rf_sum; p.q2 = p.rf2 / rf_sum; p.q3 = p.rf3 / rf_sum; }
}

// The scalar model code between the column solves (mod.rs:487-560), shared by the two kernels (so that they carry the same
// bits).  ClimateUDEB is a tolerance-parity kind: a quotient by a value that is fixed for the launch or for the year is a
// product with its reciprocal, formed once -- six IEEE divisions (58 cycles each) less per sub-step.
struct AirMap {       // sst_to_air (ocean_column.rs): alpha sst + gamma sst^2 up to the vertex of the parabola, parallel to sst beyond it
    double alpha, gamma, t_star, delta_max;
};
__device__ __forceinline__ AirMap make_air_map(const UdebP& p)
{
    AirMap m;
    m.alpha = p.alpha;
    m.gamma = p.gamma;
    m.t_star = fabs(p.gamma) > 1e-15 ? -(p.alpha - 1.0) / (2.0 * p.gamma) : __builtin_inf();
    m.delta_max = p.alpha * m.t_star + p.gamma * m.t_star * m.t_star - m.t_star;   // (not used when t_star is infinite)
    return m;
}
__device__ __forceinline__ double sst_to_air(const AirMap& m, double sst)
{
    if (sst < m.t_star) return m.alpha * sst + m.gamma * sst * sst;
    return sst + m.delta_max;
}

// land_temperature: (f_land fg_l + k_lo amplify t_air) / (lambda_land fg_l + k_lo), capped; the denominator is the year's
__device__ __forceinline__ double land_temperature(double ka, double max_temp, double ocean_temp, double land_forcing, double land_fraction,
                                                   double r_den)
{
    const double numerator = land_forcing * land_fraction + ka * ocean_temp;
    return fmin(numerator * r_den, max_temp);
}

// the effective forcing at sub-step step_idx of the year: linear between the year's ends, times the efficacy factor
__device__ __forceinline__ double substep_forcing(double erf_start, double erf_end, int32_t step_idx, double inv_steps, double eff_scale)
{
    const double frac = (double)step_idx * inv_steps;
    return (erf_start + frac * (erf_end - erf_start)) * eff_scale;
}

// Per-member geometry folded with this year's sub-step length, and everything else of a column
// solve that only changes once a year (the lambdas come out of LAMCALC per year): the sub-step
// loop is left with multiplies.  ClimateUDEB is a tolerance-parity kind (tests/test_gpu_udeb.py
// states 1e-9 against the CPU oracle), so quotients by these denominators are products with a
// refined reciprocal and sums of products are fused; the bit-exact two-layer kernel does neither.
struct YearGeom {
    double dt_dz, dt_dzmix, dt_cmix;        // dt/dz, dt/dz_mix, dt/c_mix
    double dt_dz2, dt_dzdz1, dt_dzmixdz1;   // dt/(dz*dz), dt/(dz*dz/2), dt/(dz_mix*dz/2)
    double kC, kdC, kminC;                  // kappa, dkappa/dT, kappa_min in m^2/yr
    double kC2, kdC2, kminC2;               // the same times dt/(dz*dz): the interior rows want kappa_l * dt/dz^2
    double fb[2];                           // (lambda_o + lambda_l*k_lo*amp*f_l/den) * dt/c_mix
    double famp[2];                         // 1 + k_lo*f_l/den
    double lhc[2];                          // k_lg * dt / (c_mix * f_o), land heat capacity only
};


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
// and the Thomas recurrences with one reciprocal per row, refined by its series; c' is kept negated.  The bottom row is an
// interior row whose table entries say af_bot = 0, af_diff = af_top (udeb_tables.hpp): one formula for every row below the
// mixed layer.
// FAST (RSCM_MODE_FAST): one refinement term of the row reciprocals instead of two (relative error 2^-46 instead of 2^-69 per row).
//
// The geometry tables are rows of six values per layer, [NL][6] = {af_top, af_bot, af_diff, 1 - relative depth, G_nh, G_sh}
// (udeb_tables.hpp), wave-uniform and read with scalar loads.  Scalar loads return out of order, so a wavefront can only wait for
// ALL of its outstanding ones (s_waitcnt lgkmcnt(0)); with one wavefront per SIMD every batch of table values that is loaded where
// it is needed exposes its full latency -- 49 waits per sub-step, a fifth of the solve's time.  The sweep therefore walks the
// column in chunks of kRowsAhead rows: at the top of a chunk the values of the chunk (requested a whole chunk earlier) are
// awaited, then the next chunk's rows are requested, then the chunk's arithmetic runs under those loads.  An opaque zero offset
// per solve keeps the loads from being hoisted out of the sub-step and year loops (hundreds of scalar registers, spilled into
// vector lanes), and a scheduling barrier per chunk keeps each request where it is written.
//
// DYN (a runtime layer count in a column of CAPACITY NL): the reference takes every n_layers >= 2 at the same cost per layer
// (parameters/climate_udeb.rs:41, mod.rs:162-165).  The sweep stays one straight line of NL rows -- a first attempt that skipped
// the rows past the end with a scalar branch per row ran 3.4x slower than the fixed-count kernel (182 against 53 ms at 49 / 50
// layers, 65 536 members x 750 years): each row became its own basic block and nothing of the next row's arithmetic ran under
// the latency of this row's reciprocal.  Instead the rows [nl, NL) are DATA: their table rows are zero, so b = 1, c' = 0 and
// d = the row's own (zero) temperature; the forward sweep leaves them zero, the back substitution passes through them with
// x = fma(0, x, 0), and the live bottom row, whose c' is 0 as well, gets fma(0, 0, d') = d' exactly.  The statements of a live
// row are those of the fixed-count kernel: a count carries the same bits whichever kernel holds it.  What needs the count at
// run time: the bottom temperature dp[nl - 1] (no static register index: the caller carries it in *bottom; it is picked up
// from the clamped solution with a select on the rows [LOW - 1, NL) this instance can end at), and the two-layer case of the
// reference's dz_up rule.
constexpr int kRowsAhead = 3;
constexpr int kTabCols = 6;

//
// NCP_LDS (capacities above 64 rows): the column alone fills the vector registers, so the sweep's c' array lives in this lane's LDS
// slots (ncp_lds[row * 64]; the caller's pointer is already at the lane) -- written in the forward sweep, read back in the back
// substitution; the forward recurrence itself takes the previous row's c' from a register, as before, so its dependent chain does
// not grow by an LDS round trip.  The same statements on the same values: the same bits as with c' in registers.
template <int NL, bool FAST, bool DYN = false, int LOW = NL, bool NCP_LDS = false>
__device__ __forceinline__ double step_hemisphere(const UdebP& p, const YearGeom& y,
                                                  const double* tables, int32_t land_hc,
                                                  double (&dp)[NL], int hemi,
                                                  double forcing, double hemi_hx, double ground_temp,
                                                  double land_temp, double alpha_eff, double w,
                                                  int32_t nl_rt = NL, double* bottom = nullptr, double* ncp_lds = nullptr)
{
    constexpr int R = kRowsAhead;
    constexpr int NCH = (NL + R - 1) / R;
    const int32_t nl = DYN ? nl_rt : NL;
    int32_t opaque = 0;
    asm volatile("" : "+s"(opaque));
    const double* __restrict__ tab = tables + opaque;
    const bool sh = hemi != 0;
    double cur[R][kTabCols], nxt[R][kTabCols];
    auto request = [&](double (&dst)[R][kTabCols], const double* __restrict__ from, int first) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < kTabCols; ++k) dst[r][k] = (first + r < NL) ? from[(size_t)(first + r) * kTabCols + k] : 0.0;
    };
    // The values of a chunk must have arrived before the next chunk is requested (one counter for all scalar loads): naming them
    // as inputs of an empty asm puts the wait here, a whole chunk after the request.
    auto await = [&](const double (&v)[R][kTabCols]) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" ::"s"(v[r][0]), "s"(v[r][1]), "s"(v[r][2]), "s"(v[r][3]), "s"(v[r][4]), "s"(v[r][5]));
    };
    request(cur, tab, 0);

    const double t_top = dp[0];
    double t_bot;
    if constexpr (DYN) t_bot = *bottom;
    else t_bot = dp[NL - 1];
    const double kslope = y.kdC * (t_top - t_bot);
    // kappa_l * dt/dz^2 with the (positive) factor folded into the three constants: one multiply
    // less per interior row, the same value to rounding
    const double kslope2 = y.kdC2 * (t_top - t_bot);
    const double delta_w = w - p.w0;
    // |delta_w| <= 1e-15: the reference skips the profile-advection terms; adding exact zeros is
    // the same thing without a branch per row
    const double dwv = fabs(delta_w) > 1e-15 ? delta_w : 0.0;
    const double tul = w * y.dt_dz;
    const double s_afd = p.pi_ratio * tul * t_top;
    const double dwq = y.dt_dz * dwv;

    double ncp[NCP_LDS ? 1 : NL];  // -c'
    double ncp_prev = 0.0;         // (NCP_LDS: the previous row's, for the recurrence)
    double tdu = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        await(cur);
        if (c + 1 < NCH) request(nxt, tab, (c + 1) * R);
        // Nothing crosses: the request stays above the chunk it runs under (left alone the scheduler sinks it to where its values are
        // used), and no arithmetic of the NEXT chunk is pulled up to wait on it.  Tried instead: ordering through data dependences
        // only (the next request's address and the chunk's vector operands made to depend on the asm above) -- the requests sink
        // again, 59.7 ms against 54.5 ms at 65 536 members x 750 years.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = c * R + r;
            if (i >= NL) break;
            const double af_top = cur[r][0], af_bot = cur[r][1], af_diff = cur[r][2], omr = cur[r][3], G = sh ? cur[r][5] : cur[r][4];
            if (i == 0) {   // ---- row 0 (mixed layer)
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
                if constexpr (NCP_LDS) {
                    ncp_prev = nc0 * rr;
                    ncp_lds[0] = ncp_prev;
                } else {
                    ncp[0] = nc0 * rr;
                }
                dp[0] = d0 * rr;
                // row 1 as an interior row: dz_up = dz/2; as the BOTTOM row (two layers) the reference takes dz for it (ocean_column.rs:191)
                tdu = kap0 * ((DYN && nl == 2) ? y.dt_dz2 : y.dt_dzdz1);
                continue;
            }
            // ---- every row below the mixed layer (the bottom row by its table entries; rows past the end of a DYN column: no-ops)
            const double t_i = dp[i];
            const double tdu_aft = tdu * af_top;
            const double tdd = fmax(__builtin_fma(omr, kslope2, y.kC2), y.kminC2);
            const double bi = __builtin_fma(tdu + tul, af_top, __builtin_fma(tdd, af_bot, 1.0));
            const double di = __builtin_fma(dwq, G, __builtin_fma(s_afd, af_diff, t_i));
            double c_up;
            if constexpr (NCP_LDS) c_up = ncp_prev;
            else c_up = ncp[i - 1];
            const double denom = __builtin_fma(-tdu_aft, c_up, bi);
            // 1/denom = r0 (1 + e + e^2 + ...), e = 1 - denom*r0: the hardware estimate is good
            // to ~2^-23, so the series cut after e^2 is exact to rounding, and the c' chain that
            // feeds the next row's denominator is five dependent operations instead of seven
            const double r0 = __builtin_amdgcn_rcp(denom);
            const double e = __builtin_fma(-denom, r0, 1.0);
            const double u = FAST ? e : __builtin_fma(e, e, e);
            const double t = (tdd + tul) * af_bot * r0;
            if constexpr (NCP_LDS) {
                ncp_prev = __builtin_fma(t, u, t);
                ncp_lds[(size_t)i * 64] = ncp_prev;
            } else {
                ncp[i] = __builtin_fma(t, u, t);
            }
            const double sdp = __builtin_fma(tdu_aft, dp[i - 1], di) * r0;
            dp[i] = __builtin_fma(sdp, u, sdp);
            tdu = tdd;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < kTabCols; ++k) cur[r][k] = nxt[r][k];
    }
    // ---- back substitution, clamp.  thomas_solve returns the unclamped vector; the state keeps
    // min(x, max_temp).  (The last row's c' is 0 -- the bottom row's by its table entries, a dead row's because all of its are.)
    double bot = 0.0;
    auto pick_bottom = [&](int i) {   // DYN: the clamped solution of the row that is this column's last
        if constexpr (DYN) bot = (i == nl - 1) ? dp[i] : bot;
    };
    double x = dp[NL - 1];
    dp[NL - 1] = fmin(x, p.max_temp);
    pick_bottom(NL - 1);
#pragma unroll
    for (int i = NL - 2; i >= 0; --i) {
        double c_i;
        if constexpr (NCP_LDS) c_i = ncp_lds[(size_t)i * 64];
        else c_i = ncp[i];
        x = __builtin_fma(c_i, x, dp[i]);
        dp[i] = fmin(x, p.max_temp);
        if (i >= LOW - 1) pick_bottom(i);
    }
    if constexpr (DYN) *bottom = bot;
    return dp[0];
}


// One member of a ClimateUDEB ensemble, both hemispheres in one thread, across the model steps of a launch:
// begin() (construction or resume), step(n) for consecutive n, end() (internal state back to HBM).
// The two columns never leave the chip in between: the hemisphere being solved is in registers (col[], shared with
// the solver's d' array), the other one is parked in this lane's LDS slots (park[layer][lane]: 8 NL bytes per lane,
// each lane touches only its own slots -- no barriers, no bank conflicts) and the two are exchanged after every
// column solve.  HBM sees the columns once in begin() (resume) and once in end(): ocean[hemi][layer][N].
// DYN: NL is the capacity, the column has nl = a.n_layers rows, LOW <= nl <= NL (step_hemisphere); HBM keeps [2][nl][N]; the
// rows past the end are zero in registers and in LDS and are carried along.
template <int NL, bool DYN = false, int LOW = NL>
struct Udeb1 {
    double (*park)[kUdebBlock];
    int lane;
    int32_t nl;                 // rows of a column (wave-uniform; NL when not DYN)
    double bot_cur, bot_park;   // DYN: the bottom-row temperature of the column in col[] / of the parked one
    int64_t N, i;
    UdebP p;
    int32_t status;
    LamResult base;
    double col[NL];
    double up_nh, up_sh, land_nh, land_sh, gr_nh, gr_sh, ae_nh, ae_sh, hx_nh, hx_sh;
    double win_sum, hist_last;   // running window sum of the temperature history: entries [win_lo, n-1) after year n-1
    int32_t win_lo;
    double c_ground, c_mix, steps, inv_steps;
    AirMap airmap;
    double hxf_nh, hxf_sh, inv_thresh_nh, inv_thresh_sh, ka, w_min;   // k_ns / f_ocean, 1 / t_thresh, k_lo amplify, w0 (1 - f_var)
    const double* F;
    size_t f_stride;

    __device__ __forceinline__ explicit Udeb1(double (*park_)[kUdebBlock]) : park(park_) {}

    __device__ __forceinline__ void begin(const UdebArgs& a, int64_t member)
    {
        lane = threadIdx.x;
        i = member;
        nl = DYN ? a.n_layers : NL;
        bot_cur = bot_park = 0.0;
        N = a.row_stride;   // the stride of every [..][N] array (the caller has checked `member` against a.n_members)
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
        status = 0;
        if (!is_finite(p.prescribed_eff) || p.prescribed_eff <= 0.0) status = 2;
        base = LamResult{0.0, 0.0, 1.0, false};
        if (status == 0) {
            base = base_lamcalc_from_block(a.derived, a.derived_uniform, N, i);
            if (!base.ok) status = 4;
        }
        a.status[i] = (uint8_t)status;
        win_sum = 0.0;
        hist_last = 0.0;
        win_lo = 0;
        up_nh = up_sh = p.w0;
        land_nh = land_sh = gr_nh = gr_sh = hx_nh = hx_sh = 0.0;
        ae_nh = ae_sh = p.alpha;
#pragma unroll
        for (int l = 0; l < NL; ++l) col[l] = 0.0;
        if (status != 0) return;  // the reference refuses to build this component: step() writes NaN rows
        // ---- internal state (ClimateUDEBState::new) or resume
        double* T_nh = a.ocean + i;
        double* T_sh = a.ocean + (size_t)nl * N + i;
        if (a.step_begin == 0) {
#pragma unroll
            for (int l = 0; l < NL; ++l) park[l][lane] = 0.0;
        } else {
            const double* s = a.scal + i;
            up_nh = s[0 * N]; up_sh = s[1 * N]; land_nh = s[2 * N]; land_sh = s[3 * N];
            gr_nh = s[4 * N]; gr_sh = s[5 * N]; ae_nh = s[6 * N]; ae_sh = s[7 * N];
            hx_nh = s[8 * N]; hx_sh = s[9 * N];
            win_sum = s[10 * N];
            win_lo = a.step_begin > 1 ? a.win_kfull[a.step_begin - 1] : 0;
            hist_last = a.hist[(size_t)(a.step_begin - 1) * N + i];
            // One wavefront per SIMD: nothing hides a load's latency but the loads that are in flight with it.
            // The southern column first, all NL loads at once into the registers of col[], from there to its LDS
            // slots; then the northern one (two round trips to HBM; interleaved with the LDS writes, a layer at a
            // time, the compiler waited for every pair: 25 round trips per launch).
            if constexpr (DYN) {   // (rows past the end stay zero)
                bot_park = T_sh[(size_t)(nl - 1) * N];
                bot_cur = T_nh[(size_t)(nl - 1) * N];
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (DYN && l >= nl) continue;
                col[l] = T_sh[(size_t)l * N];
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) park[l][lane] = col[l];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (DYN && l >= nl) continue;
                col[l] = T_nh[(size_t)l * N];
            }
        }
        const int32_t scen = a.scen ? a.scen[i] : 0;
        F = a.link ? a.link + i : a.erf + (size_t)scen * a.n_times;   // a linked forcing is another ensemble's [T][N] series
        f_stride = a.link ? (size_t)N : (size_t)1;
        steps = (double)a.steps_per_year;
        inv_steps = 1.0 / steps;
        c_ground = a.land_hc ? heat_capacity_per_unit_area(p.land_hc_thick) : 0.0;
        c_mix = heat_capacity_per_unit_area(p.dz_mix);
        airmap = make_air_map(p);
        hxf_nh = p.fgno > 1e-15 ? p.k_ns / p.fgno : 0.0;
        hxf_sh = p.fgso > 1e-15 ? p.k_ns / p.fgso : 0.0;
        inv_thresh_nh = 1.0 / p.t_thresh_nh;
        inv_thresh_sh = 1.0 / p.t_thresh_sh;
        ka = p.k_lo * p.amplify;
        w_min = p.w0 * (1.0 - p.f_var);
    }

    // model step n -> n + 1.  CHECK_DEAD = false: the caller has dealt with members the reference refuses to build
    // (nan_rows) and calls step() only for the others.
    template <bool CHECK_DEAD = true, bool FAST = false>
    __device__ __forceinline__ void step(const UdebArgs& a, int32_t n)
    {
        const size_t r0 = (size_t)n * N + i, r1 = r0 + (size_t)N;
        if (CHECK_DEAD && status != 0) {  // every output NaN
            const double nan = __builtin_nan("");
            a.st0[r1] = nan; a.st1[r1] = nan; a.st2[r1] = nan; a.st3[r1] = nan;
            a.heat_uptake[r1] = nan; a.ohc[r1] = nan; a.sst[r1] = nan;
            return;
        }
        double* st[4] = {a.st0, a.st1, a.st2, a.st3};
        const double* tables = a.tables;  // kernarg segment
        // everything the step reads at its start that does not depend on anything else, requested together (one wavefront per
        // SIMD: each dependent trip to memory is exposed in full); the history rows below need k_full first
        const double erf_start = F[(size_t)n * f_stride], erf_end = F[(size_t)(n + 1) * f_stride];
        const double bound_lo = a.bounds[n], bound_hi = a.bounds[n + 1];
        const int32_t k_full = a.win_kfull[n];
        const double part_w = a.win_partw[n];
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
        const double dt_year = bound_hi - bound_lo;
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
        // the year's constants of the scalar model code
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
            // NH then SH, the solver instantiated for each (a loop around one copy ties the register allocation of the
            // solve to its back edge: 360 instead of 92 accumulator moves per sub-step)
            double sst_pair[2];
#pragma unroll
            for (int hemi = 0; hemi < 2; ++hemi) {
                const bool sh = hemi != 0;
                sst_pair[hemi] = step_hemisphere<NL, FAST, DYN, LOW>(p, y, tables, a.land_hc, col, hemi, sh ? f2 : f0,
                                                                sh ? hx_sh : hx_nh, sh ? gr_sh : gr_nh,
                                                                sh ? land_sh : land_nh, sh ? ae_sh_y : ae_nh_y,
                                                                sh ? up_sh : up_nh, nl, &bot_cur);
                // exchange the solved column with the parked hemisphere.  (Tried: both columns in registers and the c' array of the solve
                // in progress in the LDS slots instead -- as many LDS instructions, none of them between two solves: the register
                // allocator answers with 505 spilled VGPRs and 652 spilled SGPRs.  Tried: the exchange layer by layer inside the back
                // substitution, so that the LDS traffic runs under it -- the compiler interleaves it as written, the yearly code
                // starts to spill, and 65 536 members x 750 years take 58.8 ms instead of 54.8.)
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double other = park[l][lane];
                    park[l][lane] = col[l];
                    col[l] = other;
                }
                if constexpr (DYN) {
                    const double other = bot_park;
                    bot_park = bot_cur;
                    bot_cur = other;
                }
            }
            const double sst_nh = sst_pair[0], sst_sh = sst_pair[1];
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
        const double sst_nh = col[0], sst_sh = park[0][lane];
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
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                if (DYN && l >= nl) continue;
                total += rho_c * p.dz * col[l];
            }
            total += rho_c * p.dz_mix * sst_sh;
#pragma unroll 7
            for (int l = 1; l < NL; ++l) {
                if (DYN && l >= nl) continue;
                total += rho_c * p.dz * park[l][lane];
            }
            a.ohc[r1] = total / 2.0;
        }
        a.st0[r1] = air_nh;
        a.st1[r1] = land_nh;
        a.st2[r1] = air_sh;
        a.st3[r1] = land_sh;
        a.sst[r1] = (sst_nh + sst_sh) / 2.0;
    }

    // the scalars and the columns go back to HBM once per launch (rscm_ens_run resumes from them)
    __device__ __forceinline__ void end(const UdebArgs& a)
    {
        if (status != 0) return;
        double* s = a.scal + i;
        s[0 * N] = up_nh; s[1 * N] = up_sh; s[2 * N] = land_nh; s[3 * N] = land_sh;
        s[4 * N] = gr_nh; s[5 * N] = gr_sh; s[6 * N] = ae_nh; s[7 * N] = ae_sh;
        s[8 * N] = hx_nh; s[9 * N] = hx_sh;
        s[10 * N] = win_sum;
        double* T_nh = a.ocean + i;
        double* T_sh = a.ocean + (size_t)nl * N + i;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (DYN && l >= nl) continue;
            T_nh[(size_t)l * N] = col[l];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int l = 0; l < NL; ++l) col[l] = park[l][lane];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (DYN && l >= nl) continue;
            T_sh[(size_t)l * N] = col[l];
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Two wavefronts per 64 members: wavefront 0 of a 128-thread workgroup carries the northern column of
// members [64 b, 64 b + 64), wavefront 1 the southern one.  Why: one thread per member holds ~245 doubles
// during a column solve (two columns, the Thomas c' array, ~45 parameters, the year's folded geometry) --
// 256 VGPRs + 256 AGPRs + a column parked in LDS and swapped after every solve; a third of the kernel's
// vector instructions were VGPR <-> AGPR moves, spills and that swap (profiles/r2_udeb_resource_usage.txt).
// A hemisphere per lane holds one column + c' + the parameters of its own boxes: ~135 doubles, no swap.
// The two solves of a sub-step are independent (mod.rs:487-560: both read the previous sub-step's land
// temperatures, inter-hemispheric exchange and upwelling); what couples them afterwards -- air and land
// temperatures for the exchange term, the global mean for the upwelling -- goes through 2 x 8 bytes per
// member of LDS and one s_barrier per sub-step.  The scalar work between the solves (LAMCALC, efficacies,
// land temperatures) is done by both wavefronts: a few per cent of the column arithmetic.
// ---------------------------------------------------------------------------------------------
constexpr int kUdeb2Block = 128;

struct Udeb2Lds {
    double xs[2][2][2][64];   // sub-step exchange [parity][hemisphere][air, land][lane]
    double xy[2][2][3][64];   // end-of-year exchange [parity][hemisphere][sst, air, heat-content partial][lane]
};

// One member-hemisphere of a ClimateUDEB ensemble across the model steps of a launch: begin() (construction
// or resume), step(n) for consecutive n, end() (internal state back to HBM).  Every thread of the workgroup
// must make the same calls: step() and begin() hold workgroup barriers.  Lanes past the end of the ensemble
// compute on a copy of the last member and store nothing.
template <int NL, bool DYN = false, int LOW = NL, bool NCP_LDS = false>
struct Udeb2 {
    Udeb2Lds& lds;
    double* ncp_lds = nullptr;   // NCP_LDS: this lane's column of c' slots ([NL][64] doubles per wavefront; the kernel sets it)
    int tid, lane;
    int32_t nl;      // rows of the column (wave-uniform; NL when not DYN)
    double bot;      // DYN: the column's bottom-row temperature
    int hemi;        // 0: northern column, 1: southern (wave-uniform)
    int64_t N, i;
    bool live;       // this lane stands for a member of the ensemble
    UdebP p;
    int32_t status;
    LamResult base;
    double col[NL];
    double up, land, gr, ae, hx;   // this hemisphere's upwelling, land and ground temperature, alpha_eff, exchange term
    double land_o;                 // the other hemisphere's land temperature as of the last sub-step
    double top_o;                  // the other hemisphere's mixed-layer temperature as of the last year end
    double win_sum, hist_last;
    int32_t win_lo;
    double c_ground, c_mix, steps, inv_steps;
    AirMap airmap;
    double hxf, inv_thresh, ka, w_min;   // this hemisphere's k_ns / f_ocean and 1 / t_thresh; k_lo amplify, w0 (1 - f_var)
    const double* F;
    size_t f_stride;
    uint32_t n_sub;                // sub-steps taken in this launch (parity of the exchange slots)

    __device__ __forceinline__ explicit Udeb2(Udeb2Lds& l) : lds(l) {}

    // (the launch arguments are handed to every call instead of being kept: a reference held in this object makes
    // the compiler copy the by-value kernel argument -- 2.6 KB with the geometry tables -- into scratch)
    static __device__ __forceinline__ double* box(const UdebArgs& a, int k) { return k == 0 ? a.st0 : k == 1 ? a.st1 : k == 2 ? a.st2 : a.st3; }

    __device__ __forceinline__ void begin(const UdebArgs& a)
    {
        tid = threadIdx.x;
        lane = tid & 63;
        hemi = __builtin_amdgcn_readfirstlane(tid >> 6);
        nl = DYN ? a.n_layers : NL;
        bot = 0.0;
        N = a.row_stride;   // the stride of every [..][N] array; a.n_members: the members this launch covers
        const int64_t i_raw = (int64_t)blockIdx.x * 64 + lane;
        live = i_raw < a.n_members;
        i = live ? i_raw : a.n_members - 1;
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
        status = 0;
        if (!is_finite(p.prescribed_eff) || p.prescribed_eff <= 0.0) status = 2;
        base = LamResult{0.0, 0.0, 1.0, false};
        if (status == 0) {
            base = base_lamcalc_from_block(a.derived, a.derived_uniform, N, i);
            if (!base.ok) status = 4;
        }
        if (live && hemi == 0) a.status[i] = (uint8_t)status;
        // ---- internal state (ClimateUDEBState::new) or resume
        win_sum = 0.0;
        hist_last = 0.0;
        win_lo = 0;
        n_sub = 0;
        double* T_own = a.ocean + (size_t)hemi * nl * N + i;
        if (a.step_begin == 0) {
#pragma unroll
            for (int l = 0; l < NL; ++l) col[l] = 0.0;
            up = p.w0;
            land = gr = hx = land_o = top_o = 0.0;
            ae = p.alpha;
        } else {
            const double* s = a.scal + (size_t)hemi * N + i;   // rows 2k + hemisphere
            up = s[0 * N]; land = s[2 * N]; gr = s[4 * N]; ae = s[6 * N]; hx = s[8 * N];
            land_o = a.scal[(size_t)(2 + (1 - hemi)) * N + i];
            win_sum = a.scal[(size_t)10 * N + i];
            win_lo = a.step_begin > 1 ? a.win_kfull[a.step_begin - 1] : 0;
            hist_last = a.hist[(size_t)(a.step_begin - 1) * N + i];
            top_o = a.ocean[(size_t)(1 - hemi) * nl * N + i];
            if constexpr (DYN) bot = T_own[(size_t)(nl - 1) * N];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (DYN && l >= nl) { col[l] = 0.0; continue; }
                col[l] = T_own[(size_t)l * N];
            }
        }
        const int32_t scen = a.scen ? a.scen[i] : 0;
        F = a.link ? a.link + i : a.erf + (size_t)scen * a.n_times;   // a linked forcing is another ensemble's [T][N] series
        f_stride = a.link ? (size_t)N : (size_t)1;
        steps = (double)a.steps_per_year;
        inv_steps = 1.0 / steps;
        c_ground = a.land_hc ? heat_capacity_per_unit_area(p.land_hc_thick) : 0.0;
        c_mix = heat_capacity_per_unit_area(p.dz_mix);
        airmap = make_air_map(p);
        {
            const double fg_ocean = hemi ? p.fgso : p.fgno;
            hxf = fg_ocean > 1e-15 ? p.k_ns / fg_ocean : 0.0;
            inv_thresh = 1.0 / (hemi ? p.t_thresh_sh : p.t_thresh_nh);
        }
        ka = p.k_lo * p.amplify;
        w_min = p.w0 * (1.0 - p.f_var);
    }

    // model step n -> n + 1 (one launch may take many)
    template <bool FAST = false>
    __device__ __forceinline__ void step(const UdebArgs& a, int32_t n)
    {
        const bool sh = hemi != 0;
        // kernarg segment; above 64 rows the table does not fit there (6 KB at 128 rows): the same rows in device memory, still wave-uniform
        // (read by CAPACITY: both homes are zero-padded to the largest capacity that reads them, rscm_device.hpp)
        static_assert(NCP_LDS ? NL <= kUdebDevTableRows : NL <= kUdebArgTableRows,
                      "this instance reads NL rows of a geometry table that holds fewer: raise kUdebDevTableRows / kUdebArgTableRows with the capacity");
        const double* tables = NCP_LDS ? a.tables_dev : a.tables;
        const double nan = __builtin_nan("");
        const bool dead = status != 0;    // the reference refuses to build this component: every output NaN
        const double erf_start = F[(size_t)n * f_stride], erf_end = F[(size_t)(n + 1) * f_stride];
        const size_t r0 = (size_t)n * N + i, r1 = r0 + (size_t)N;
        // warm start (mod.rs:436-446)
        {
            const double prev0 = a.st0[r0];
            const double top_nh = sh ? top_o : col[0];
            if (top_nh == 0.0 && prev0 != 0.0) {
                col[0] = box(a, 2 * hemi)[r0];
                top_o = box(a, 2 * (1 - hemi))[r0];
                land = box(a, 2 * hemi + 1)[r0];
                land_o = box(a, 2 * (1 - hemi) + 1)[r0];
                gr = land;
            }
        }
        const double dt_year = a.bounds[n + 1] - a.bounds[n];
        const double dt_sub = dt_year / steps;
        // ---- time-varying ECS (adjusted_ecs) and the LAMCALC re-solve: both wavefronts, same values
        const double erf_mid = (erf_start + erf_end) / 2.0;
        double cum_t = 0.0;
        if (n > 0) {
            const int32_t k_full = a.win_kfull[n];
            const double part_w = a.win_partw[n];
            const double* hcol = a.hist + i;
            win_sum += hist_last;
            // Entry n-1 is the one the hemisphere-0 wavefront stored AFTER the previous year's last barrier: the other wavefront may
            // get here before that store has landed.  Both wavefronts hold its value (hist_last), so neither reads it back -- a
            // window shorter than the previous model step (k_full == n) would otherwise race on it.
            auto hist_at = [&](int32_t k) -> double { return k == n - 1 ? hist_last : hcol[(size_t)k * N]; };
            for (; win_lo < k_full; ++win_lo) win_sum -= hist_at(win_lo);
            if (p.fb_cumt != 0.0) {
                cum_t = win_sum;
                if (part_w > 0.0) cum_t += hist_at(k_full - 1) * part_w;
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
        const double ae_y = ae;  // alpha_eff is fixed for the year
        // this hemisphere's boxes
        const double fg_o = sh ? p.fgso : p.fgno, fg_l = sh ? p.fgsl : p.fgnl;
        const double q_o = sh ? p.q2 : p.q0, q_l = sh ? p.q3 : p.q1;
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
            const double f_l = (sh ? p.sh_land : p.nh_land) / 2.0;
            const double f_o = 0.5 - f_l;
            const double den = f_o * (p.k_lo + f_l * lam_l);
            y.fb[0] = y.fb[1] = (lam_o + lam_l * p.k_lo * p.amplify * f_l / den) * y.dt_cmix;
            y.famp[0] = y.famp[1] = 1.0 + p.k_lo * f_l / den;
            y.lhc[0] = y.lhc[1] = a.land_hc ? p.k_lg * dt_sub / (c_mix * f_o) : 0.0;
        }
        // the year's constants of the scalar model code (the same expressions as in Udeb1::step)
        const double eff_scale = eff_mode == 1 ? p.prescribed_eff : eff_mode == 2 ? p.prescribed_eff / co2_eff : 1.0;
        const double r_land = 1.0 / (lam_l * fg_l + p.k_lo);
        const double gfac = (a.land_hc && !(fg_l < 1e-15)) ? p.k_lg / (fg_l * c_ground) * dt_sub : 0.0;
        double t_air = 0.0, t_air_o = 0.0;
        for (int32_t step_idx = 1; step_idx <= a.steps_per_year; ++step_idx) {
            const double adj = substep_forcing(erf_start, erf_end, step_idx, inv_steps, eff_scale);
            const double f_ocean = adj * q_o, f_land = adj * q_l;
            if (a.land_hc) gr = __builtin_fma(land - gr, gfac, gr);
            const double sst = step_hemisphere<NL, FAST, DYN, LOW, NCP_LDS>(p, y, tables, a.land_hc, col, hemi, f_ocean, hx, gr, land, ae_y, up, nl, &bot, ncp_lds);   // the same function as the one-thread kernel: the same bits
            t_air = sst_to_air(airmap, sst);
            land = land_temperature(ka, p.max_temp, t_air, f_land, fg_l, r_land);
            // what the other hemisphere needs of this one: air and land temperature
            const uint32_t par = n_sub & 1u;
            lds.xs[par][hemi][0][lane] = t_air;
            lds.xs[par][hemi][1][lane] = land;
            __syncthreads();
            t_air_o = lds.xs[par][1 - hemi][0][lane];
            land_o = lds.xs[par][1 - hemi][1][lane];
            ++n_sub;
            if (fg_o > 1e-15) hx = hxf * (t_air_o - t_air);
            const double a_nh = sh ? t_air_o : t_air, l_nh = sh ? land_o : land;
            const double a_sh = sh ? t_air : t_air_o, l_sh = sh ? land : land_o;
            const double global_temp = a_nh * p.fgno + l_nh * p.fgnl + a_sh * p.fgso + l_sh * p.fgsl;
            up = fmax(p.w0 * (1.0 - p.f_var * fmin(global_temp * inv_thresh, 1.0)), w_min);   // update_upwelling
        }
        // ---- end of year
        const double sst = col[0];
        const double air = sst_to_air(airmap, sst);
        ae = fabs(sst) < 1e-15 ? p.alpha : air / sst;
        const double rho_c = kRhoSeawater * kCpSeawater;
        const uint32_t ypar = (uint32_t)n & 1u;
        if (!sh) {   // calculate_ocean_heat_content adds hemisphere by hemisphere, layer by layer: the northern part first
            double total = 0.0;
            total += rho_c * p.dz_mix * sst;
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                if (DYN && l >= nl) continue;
                total += rho_c * p.dz * col[l];
            }
            lds.xy[ypar][0][2][lane] = total;
        }
        lds.xy[ypar][hemi][0][lane] = sst;
        lds.xy[ypar][hemi][1][lane] = air;
        __syncthreads();
        const double sst_o = lds.xy[ypar][1 - hemi][0][lane];
        const double air_o = lds.xy[ypar][1 - hemi][1][lane];
        top_o = sst_o;
        const double air_nh = sh ? air_o : air, land_nh = sh ? land_o : land;
        const double air_sh = sh ? air : air_o, land_sh = sh ? land : land_o;
        const double global_temp = air_nh * p.fgno + land_nh * p.fgnl + air_sh * p.fgso + land_sh * p.fgsl;
        hist_last = global_temp * dt_year;
        if (!sh) {
            if (live) a.hist[r0] = hist_last;
            double adj_end = erf_end;
            if (eff_mode == 1) adj_end = erf_end * p.prescribed_eff;
            else if (eff_mode == 2) adj_end = erf_end * p.prescribed_eff / co2_eff;
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
            if (live) {
                a.heat_uptake[r1] = dead ? nan : q_global - feedback_global;
                a.st0[r1] = dead ? nan : air;
                a.st1[r1] = dead ? nan : land;
                a.sst[r1] = dead ? nan : (sst + sst_o) / 2.0;
            }
        } else {
            double total = lds.xy[ypar][0][2][lane];
            total += rho_c * p.dz_mix * sst;
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                if (DYN && l >= nl) continue;
                total += rho_c * p.dz * col[l];
            }
            if (live) {
                a.ohc[r1] = dead ? nan : total / 2.0;
                a.st2[r1] = dead ? nan : air;
                a.st3[r1] = dead ? nan : land;
            }
        }
    }

    // the internal state goes back to HBM once per launch (rscm_ens_run resumes from it)
    __device__ __forceinline__ void end(const UdebArgs& a)
    {
        if (!live || status != 0) return;
        double* s = a.scal + (size_t)hemi * N + i;
        s[0 * N] = up; s[2 * N] = land; s[4 * N] = gr; s[6 * N] = ae; s[8 * N] = hx;
        if (hemi == 0) a.scal[(size_t)10 * N + i] = win_sum;
        double* T_own = a.ocean + (size_t)hemi * nl * N + i;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (DYN && l >= nl) continue;
            T_own[(size_t)l * N] = col[l];
        }
    }
};

}  // namespace udeb
}  // namespace rscm
