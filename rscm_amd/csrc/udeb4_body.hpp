// ClimateUDEB with FOUR wavefronts per 64 members: a hemisphere's column is cut in the middle, the upper half is eliminated from
// the top down by one wavefront, the lower half from the bottom up by another, the two meet in a 2 x 2 system at the cut and
// substitute outwards again (a "twisted" factorisation of the tridiagonal system of ocean_column.rs step_hemisphere).
//
// Why: the one-thread kernel (udeb_body.hpp: Udeb1) and the hemisphere-per-wavefront kernel (Udeb2) keep a whole column and its
// c' array in registers -- ~210 doubles per lane with the parameters and the year's constants, 512 registers, ONE wavefront per
// SIMD.  A lone wavefront issues every instruction in sequence: its vector pipe is busy 77 % of the time, the rest is scalar, LDS
// and wait instructions nothing overlaps with.  Half a column and half a c' array leave a lane at ~120
// doubles: 256 registers, TWO wavefronts per SIMD, and what one wavefront cannot overlap the other one fills.
//
// The same tridiagonal system row for row (coefficients as in udeb_body.hpp: step_hemisphere), solved in a different order of
// operations: the results agree with the one-thread kernel to rounding (tests/test_gpu_udeb.py states the same 1e-9 against the
// CPU oracle for every kernel), not bit for bit.
//
// Roles, per workgroup of 256 threads = 64 members: wavefront r in 0..3, hemisphere r >> 1; the "top" wavefront of a hemisphere
// carries rows 0 .. H-1 (H = NL / 2), the mixed layer's forcing terms and ALL of the scalar model code of its hemisphere
// (efficacy, land and ground temperatures, hemispheric exchange, upwelling, LAMCALC, outputs: what Udeb2's wavefront does); the
// "bottom" wavefront carries rows H .. NL-1 and a handful of parameters.  Which of a hemisphere's two wavefronts is the top one
// alternates with the workgroup's index, so that a SIMD hosting wavefront r of two workgroups gets one of each role.
// Per sub-step three workgroup barriers: (1) the top wavefront's upwelling velocity and mixed-layer temperature against the bottom
// wavefront's bottom-layer temperature (both sweeps need T_top - T_bottom), (2) the two rows at the cut, (3) the hemispheres' air
// and land temperatures (as in Udeb2).  Every wavefront of a workgroup makes the same calls.  (Tried: two barriers -- the bottom
// wavefront forms the upwelling velocity for itself from the published air and land temperatures, the column's ends travel with
// them: 27.7 / 74.1 ms at 4096 / 65 536 members x 750 years against 27.2 / 73.1 ms; the barriers are not what the wavefronts wait for.)
#pragma once

#include "udeb_body.hpp"

namespace rscm {
namespace udeb {

struct Udeb4Lds {
    double top[2][2][64];      // [hemisphere][upwelling velocity, mixed-layer temperature]   top -> bottom wavefront, per sub-step
    double bottom[2][64];      // [hemisphere] bottom-layer temperature                       bottom -> top wavefront, per sub-step
    double cut[2][2][2][64];   // [hemisphere][half][c' or a', d' or e'] of the row at the cut, per sub-step
    double xs[2][2][2][64];    // sub-step exchange between the hemispheres [parity][hemisphere][air, land]
    double xy[2][2][2][64];    // end-of-year exchange [parity][hemisphere][sst, air]
    double heat[2][2][2][64];  // end-of-year heat-content partial sums [parity][hemisphere][half]
    int32_t status[2][64];     // [hemisphere] construction status, top -> bottom wavefront, once per launch
    double base[2][3][64];     // [hemisphere] the LAMCALC base solve (lambda_ocean, lambda_land, efficacy), kept from begin()
    double ncp[2][25][64];     // [hemisphere][row] c' of the top half's rows (written by its sweep, read by its substitution)
};
static_assert(sizeof(Udeb4Lds) <= 64 * 1024, "two workgroups per CU");

constexpr int kUdeb4Block = 256;

// The rows of one half of a column, swept towards the cut.  TOP: rows 0 .. H-1 downwards (the forward sweep of step_hemisphere,
// unchanged); bottom: rows NL-1 .. H upwards, the mirror image (the super-diagonal is eliminated instead of the sub-diagonal).
// On return slot j of dp holds d' (top: row j; bottom: e' of row H + j) and slot j of cp the matching c' / a' (negated, as in
// step_hemisphere), and the caller couples the two halves.
struct SolveScalars {
    double t_top, t_bottom, w;   // the column's ends and the upwelling velocity of this sub-step
};

template <int NL, bool FAST>
__device__ __forceinline__ void sweep_top(double w0, double pi_ratio, const YearGeom& y, const double* tables, int32_t land_hc, double (&dp)[NL / 2],
                                          double* __restrict__ ncp_lds, int hemi, const SolveScalars& s, double forcing, double hemi_hx,
                                          double ground_temp, double land_temp, double alpha_eff)
{
    // c' goes to this lane's LDS slots (ncp_lds[row * 64]): the top wavefront also carries the hemisphere's scalar model code, and
    // its registers are what decides whether two wavefronts fit a SIMD; the row before is all the sweep itself reads back
    double ncp_prev = 0.0;
    constexpr int H = NL / 2;
    constexpr int R = kRowsAhead;
    constexpr int NCH = (H + R - 1) / R;
    int32_t opaque = 0;
    asm volatile("" : "+s"(opaque));
    const double* __restrict__ tab = tables + opaque;
    const bool sh = hemi != 0;
    double cur[R][kTabCols], nxt[R][kTabCols];
    auto request = [&](double (&dst)[R][kTabCols], int first) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < kTabCols; ++k) dst[r][k] = (first + r < H) ? tab[(size_t)(first + r) * kTabCols + k] : 0.0;
    };
    auto await = [&](const double (&v)[R][kTabCols]) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" ::"s"(v[r][0]), "s"(v[r][1]), "s"(v[r][2]), "s"(v[r][3]), "s"(v[r][4]), "s"(v[r][5]));
    };
    request(cur, 0);
    const double t_top = s.t_top, w = s.w;
    const double kslope = y.kdC * (t_top - s.t_bottom);
    const double kslope2 = y.kdC2 * (t_top - s.t_bottom);
    const double delta_w = w - w0;
    const double dwv = fabs(delta_w) > 1e-15 ? delta_w : 0.0;
    const double tul = w * y.dt_dz;
    const double s_afd = pi_ratio * tul * t_top;
    const double dwq = y.dt_dz * dwv;
    double tdu = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        await(cur);
        if (c + 1 < NCH) request(nxt, (c + 1) * R);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = c * R + r;
            if (i >= H) break;
            const double af_top = cur[r][0], af_bot = cur[r][1], af_diff = cur[r][2], omr = cur[r][3], G = sh ? cur[r][5] : cur[r][4];
            if (i == 0) {   // ---- row 0 (mixed layer), as in step_hemisphere
                const double kap0 = fmax(__builtin_fma(omr, kslope, y.kC), y.kminC);
                const double term_diff = kap0 * y.dt_dzmixdz1;
                const double term_upwell = w * y.dt_dzmix;
                const double tf = alpha_eff * (sh ? y.fb[1] : y.fb[0]);
                const double b0 = __builtin_fma(tf, af_top, __builtin_fma(__builtin_fma(term_upwell, pi_ratio, term_diff), af_bot, 1.0));
                const double nc0 = (term_diff + term_upwell) * af_bot;
                const double q = __builtin_fma(forcing, sh ? y.famp[1] : y.famp[0], hemi_hx) * y.dt_cmix;
                double d0 = __builtin_fma(q, af_top, t_top);
                if (land_hc) d0 = __builtin_fma(-(land_temp - ground_temp) * (sh ? y.lhc[1] : y.lhc[0]), af_top, d0);
                d0 = __builtin_fma(y.dt_dzmix * dwv, G, d0);
                const double rr = refined_rcp(b0);
                ncp_prev = nc0 * rr;
                ncp_lds[0] = ncp_prev;
                dp[0] = d0 * rr;
                tdu = kap0 * y.dt_dzdz1;  // row 1: dz_up = dz/2
                continue;
            }
            const double t_i = dp[i];
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
            ncp_lds[(size_t)i * 64] = ncp_prev;
            const double sdp = __builtin_fma(tdu_aft, dp[i - 1], di) * r0;
            dp[i] = __builtin_fma(sdp, u, sdp);
            tdu = tdd;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < kTabCols; ++k) cur[r][k] = nxt[r][k];
    }
}

// Rows NL-1 .. H upwards.  Slot j of the arrays is row H + j.  With x_{i+1} = e'_{i+1} + a'_{i+1} x_i substituted into row i
// (a' kept negated like c'):  denom = b_i - cb_i a'_{i+1},  a'_i = tdu_aft_i / denom,  e'_i = (d_i + cb_i e'_{i+1}) / denom,
// cb_i = (tdd_i + tul) af_bot[i] the (negated) super-diagonal entry, tdu_aft_i = tdu_i af_top[i] the (negated) sub-diagonal one
// and tdu_i = tdd_{i-1} the diffusivity at the row's upper interface -- which the row ABOVE owns, i.e. the one the sweep comes to
// next: its table value (1 - relative depth of row i-1) is asked for with row i's chunk.
template <int NL, bool FAST>
__device__ __forceinline__ void sweep_bottom(double w0, double pi_ratio, const YearGeom& y, const double* tables, double (&dp)[NL / 2],
                                             double (&nap)[NL / 2], int hemi, const SolveScalars& s)
{
    constexpr int H = NL / 2;
    constexpr int NB = NL - H;                 // rows of this half (= H: the supported layer counts are even)
    constexpr int R = kRowsAhead;
    constexpr int NCH = (NB + R - 1) / R;
    int32_t opaque = 0;
    asm volatile("" : "+s"(opaque));
    const double* __restrict__ tab = tables + opaque;
    const bool sh = hemi != 0;
    // chunk c holds rows NL-1 - cR - r (r = 0 .. R-1) and, in `above`, 1 - relative depth of the row above its last one
    double cur[R][kTabCols], nxt[R][kTabCols], cur_above, nxt_above;
    auto request = [&](double (&dst)[R][kTabCols], double& above, int c) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = NL - 1 - c * R - r;
#pragma unroll
            // (row H - 1 belongs to the other half; its 1 - relative depth is what row H's upper interface needs)
            for (int k = 0; k < kTabCols; ++k) dst[r][k] = (row >= H - 1) ? tab[(size_t)row * kTabCols + k] : 0.0;
        }
        const int last = NL - 1 - c * R - (R - 1);
        above = (last - 1 >= H - 1 && last - 1 >= 0) ? tab[(size_t)(last - 1) * kTabCols + 3] : 0.0;
    };
    auto await = [&](const double (&v)[R][kTabCols], const double& above) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" ::"s"(v[r][0]), "s"(v[r][1]), "s"(v[r][2]), "s"(v[r][3]), "s"(v[r][4]), "s"(v[r][5]));
        asm volatile("" ::"s"(above));
    };
    request(cur, cur_above, 0);
    const double t_top = s.t_top, w = s.w;
    const double kslope2 = y.kdC2 * (t_top - s.t_bottom);
    const double delta_w = w - w0;
    const double dwv = fabs(delta_w) > 1e-15 ? delta_w : 0.0;
    const double tul = w * y.dt_dz;
    const double s_afd = pi_ratio * tul * t_top;
    const double dwq = y.dt_dz * dwv;
    auto kappa_at = [&](double omr) -> double { return fmax(__builtin_fma(omr, kslope2, y.kC2), y.kminC2); };
    double tdd = 0.0;   // the row's own (lower-interface) diffusivity term: carried down from the row below, where it was tdu
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        await(cur, cur_above);
        if (c + 1 < NCH) request(nxt, nxt_above, c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = NL - 1 - c * R - r;
            if (i < H) break;
            const int j = i - H;
            const double af_top = cur[r][0], af_bot = cur[r][1], af_diff = cur[r][2], G = sh ? cur[r][5] : cur[r][4];
            const double omr_above = (r + 1 < R) ? cur[r + 1][3] : cur_above;   // 1 - relative depth of row i - 1
            const double tdu = kappa_at(omr_above);
            const double tdu_aft = tdu * af_top;
            const double t_i = dp[j];
            if (i == NL - 1) {   // ---- the bottom row: no lower neighbour
                const double bi = __builtin_fma(tdu + tul, af_top, 1.0);
                const double di = __builtin_fma(dwq, G, __builtin_fma(s_afd, af_top, t_i));
                const double rr = refined_rcp(bi);
                nap[j] = tdu_aft * rr;
                dp[j] = di * rr;
            } else {
                const double bi = __builtin_fma(tdu + tul, af_top, __builtin_fma(tdd, af_bot, 1.0));
                const double cb = (tdd + tul) * af_bot;
                const double di = __builtin_fma(dwq, G, __builtin_fma(s_afd, af_diff, t_i));
                const double denom = __builtin_fma(-cb, nap[j + 1], bi);
                const double r0 = __builtin_amdgcn_rcp(denom);
                const double e = __builtin_fma(-denom, r0, 1.0);
                const double u = FAST ? e : __builtin_fma(e, e, e);
                const double t = tdu_aft * r0;
                nap[j] = __builtin_fma(t, u, t);
                const double sdp = __builtin_fma(cb, dp[j + 1], di) * r0;
                dp[j] = __builtin_fma(sdp, u, sdp);
            }
            tdd = tdu;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < kTabCols; ++k) cur[r][k] = nxt[r][k];
        cur_above = nxt_above;
    }
}

// A member's parameter block with the box fractions and compute_qfrac (what begin() of the other kernels forms once per launch).
__device__ __forceinline__ UdebP load_udeb_params(const UdebArgs& a, int64_t N, int64_t i)
{
    UdebP p;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    p.dz_mix = P(1); p.dz = P(2); p.kappa = P(3); p.kappa_min = P(4); p.kappa_dkdt = P(5);
    p.w0 = P(6); p.f_var = P(7); p.t_thresh_nh = P(8); p.t_thresh_sh = P(9);
    p.ecs = P(10); p.rf_2x = P(11); p.rlo = P(12); p.fb_q = P(13); p.fb_cumt = P(14); p.fb_period = P(15);
    p.k_lo = P(16); p.k_ns = P(17); p.amplify = P(18); p.nh_land = P(19); p.sh_land = P(20);
    p.alpha = P(22); p.gamma = P(23); p.pi_ratio = P(24); p.k_lg = P(26); p.land_hc_thick = P(27);
    p.rf0 = P(28); p.rf1 = P(29); p.rf2 = P(30); p.rf3 = P(31); p.prescribed_eff = P(33); p.max_temp = P(36);
    p.fgnl = p.nh_land / 2.0; p.fgno = 0.5 - p.fgnl; p.fgsl = p.sh_land / 2.0; p.fgso = 0.5 - p.fgsl;
    const double rf_sum = p.rf0 * p.fgno + p.rf1 * p.fgnl + p.rf2 * p.fgso + p.rf3 * p.fgsl;   // compute_qfrac
    if (fabs(rf_sum) <= 1e-15) { p.q0 = p.q1 = p.q2 = p.q3 = 1.0; }
    else { p.q0 = p.rf0 / rf_sum; p.q1 = p.rf1 / rf_sum; p.q2 = p.rf2 / rf_sum; p.q3 = p.rf3 / rf_sum; }
    return p;
}

// One member-hemisphere-half of a ClimateUDEB ensemble across the model steps of a launch (see the head of this file).
// What a lane keeps ACROSS the model steps is its half column and the hemisphere's handful of state values; the parameter block
// is read again every model step (uniform rows mostly: a few cache lines per wavefront and year) and everything derived from it
// lives for that step only -- the registers decide whether two wavefronts fit a SIMD.
template <int NL>
struct Udeb4 {
    static constexpr int H = NL / 2;
    static_assert(NL % 2 == 0 && H >= 2 && H <= 25, "the column is cut in the middle; Udeb4Lds::ncp holds 25 rows");
    Udeb4Lds& lds;
    int lane;
    int hemi;        // 0: northern column, 1: southern (wave-uniform)
    bool top;        // this wavefront carries rows 0 .. H-1 and the hemisphere's scalar model code (wave-uniform)
    int64_t N, i;
    bool live;       // this lane stands for a member of the ensemble
    int32_t status;
    double col[H];
    double steps, inv_steps;
    // ---- the top wavefront's state (Udeb2's)
    double up, land, gr, ae, hx, land_o, top_o;
    double win_sum, hist_last;
    int32_t win_lo;
    const double* F;
    size_t f_stride;
    uint32_t n_sub;
    // ---- the bottom wavefront's parameters
    double b_dz, b_kappa, b_kappa_min, b_kappa_dkdt, b_w0, b_pi_ratio, b_max_temp;

    __device__ __forceinline__ explicit Udeb4(Udeb4Lds& l) : lds(l) {}

    static __device__ __forceinline__ double* box(const UdebArgs& a, int k) { return k == 0 ? a.st0 : k == 1 ? a.st1 : k == 2 ? a.st2 : a.st3; }

    __device__ __forceinline__ void begin(const UdebArgs& a)
    {
        const int tid = threadIdx.x;
        lane = tid & 63;
        const int r = __builtin_amdgcn_readfirstlane(tid >> 6);
        hemi = r >> 1;
        top = (((uint32_t)r ^ blockIdx.x) & 1u) == 0u;
        N = a.n_members;
        const int64_t i_raw = (int64_t)blockIdx.x * 64 + lane;
        live = i_raw < N;
        i = live ? i_raw : N - 1;
        steps = (double)a.steps_per_year;
        inv_steps = 1.0 / steps;
        n_sub = 0;
        status = 0;
        up = land = gr = ae = hx = land_o = top_o = win_sum = hist_last = 0.0;
        win_lo = 0;
        F = nullptr;
        f_stride = 1;
        b_dz = b_kappa = b_kappa_min = b_kappa_dkdt = b_w0 = b_pi_ratio = b_max_temp = 0.0;
        double* T_own = a.ocean + ((size_t)hemi * NL + (top ? 0 : H)) * N + i;
        if (top) {
            const UdebP p = load_udeb_params(a, N, i);
            // ---- construction: from_parameters (mod.rs:161-227)
            if (!is_finite(p.prescribed_eff) || p.prescribed_eff <= 0.0) status = 2;
            LamResult base = LamResult{0.0, 0.0, 1.0, false};
            if (status == 0) {
                base = lamcalc(p, p.ecs);
                if (!base.ok) status = 4;
            }
            if (live && hemi == 0) a.status[i] = (uint8_t)status;
            lds.status[hemi][lane] = status;
            lds.base[hemi][0][lane] = base.lam_o;
            lds.base[hemi][1][lane] = base.lam_l;
            lds.base[hemi][2][lane] = base.eff;
            if (a.step_begin == 0) {
                up = p.w0;
                ae = p.alpha;
            } else {
                const double* s = a.scal + (size_t)hemi * N + i;   // rows 2k + hemisphere
                up = s[0 * N]; land = s[2 * N]; gr = s[4 * N]; ae = s[6 * N]; hx = s[8 * N];
                land_o = a.scal[(size_t)(2 + (1 - hemi)) * N + i];
                win_sum = a.scal[(size_t)10 * N + i];
                win_lo = a.step_begin > 1 ? a.win_kfull[a.step_begin - 1] : 0;
                hist_last = a.hist[(size_t)(a.step_begin - 1) * N + i];
                top_o = a.ocean[(size_t)(1 - hemi) * NL * N + i];
            }
            const int32_t scen = a.scen ? a.scen[i] : 0;
            F = a.link ? a.link + i : a.erf + (size_t)scen * a.n_times;   // a linked forcing is another ensemble's [T][N] series
            f_stride = a.link ? (size_t)N : (size_t)1;
        } else {
            auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
            b_dz = P(2); b_kappa = P(3); b_kappa_min = P(4); b_kappa_dkdt = P(5); b_w0 = P(6); b_pi_ratio = P(24); b_max_temp = P(36);
        }
        if (a.step_begin == 0) {
#pragma unroll
            for (int l = 0; l < H; ++l) col[l] = 0.0;
        } else {
#pragma unroll
            for (int l = 0; l < H; ++l) col[l] = T_own[(size_t)l * N];
        }
        __syncthreads();
        if (!top) status = lds.status[hemi][lane];
    }

    // the part of YearGeom both halves' sweeps read
    static __device__ __forceinline__ void column_geometry(YearGeom& y, double dt_sub, double dz, double kappa, double kappa_min, double kappa_dkdt)
    {
        y.dt_dz = dt_sub / dz;
        y.dt_dz2 = dt_sub / (dz * dz);
        y.kC = kappa * kDiffCm2sToM2yr;
        y.kdC = kappa_dkdt * kDiffCm2sToM2yr;
        y.kminC = kappa_min * kDiffCm2sToM2yr;
        y.kC2 = y.kC * y.dt_dz2;
        y.kdC2 = y.kdC * y.dt_dz2;
        y.kminC2 = y.kminC * y.dt_dz2;
    }

    // The two halves meet: x_{H-1} = d' + c' x_H (top), x_H = e' + a' x_{H-1} (bottom); x_H from the 2 x 2 system of the two rows
    // at the cut, which both wavefronts of the hemisphere have published.
    __device__ __forceinline__ void cut_solution(double& x_hm1, double& x_h) const
    {
        const double cp = lds.cut[hemi][0][0][lane], dpt = lds.cut[hemi][0][1][lane];   // c', d' of row H - 1
        const double ap = lds.cut[hemi][1][0][lane], ep = lds.cut[hemi][1][1][lane];    // a', e' of row H
        x_h = __builtin_fma(ap, dpt, ep) * refined_rcp(__builtin_fma(-ap, cp, 1.0));
        x_hm1 = __builtin_fma(cp, x_h, dpt);
    }

    // model step n -> n + 1 (one launch may take many)
    template <bool FAST = false>
    __device__ __forceinline__ void step(const UdebArgs& a, int32_t n)
    {
        const double* tables = a.tables;  // kernarg segment
        const double dt_year = a.bounds[n + 1] - a.bounds[n];
        const double dt_sub = dt_year / steps;
        const double rho_c = kRhoSeawater * kCpSeawater;
        const uint32_t ypar = (uint32_t)n & 1u;
        if (!top) {
            // ---- the lower half of the column: sweeps, nothing else
            YearGeom y;
            column_geometry(y, dt_sub, b_dz, b_kappa, b_kappa_min, b_kappa_dkdt);
            double nap[H];
            for (int32_t step_idx = 1; step_idx <= a.steps_per_year; ++step_idx) {
                lds.bottom[hemi][lane] = col[H - 1];
                __syncthreads();   // (1)
                SolveScalars s;
                s.w = lds.top[hemi][0][lane];
                s.t_top = lds.top[hemi][1][lane];
                s.t_bottom = col[H - 1];
                sweep_bottom<NL, FAST>(b_w0, b_pi_ratio, y, tables, col, nap, hemi, s);
                lds.cut[hemi][1][0][lane] = nap[0];
                lds.cut[hemi][1][1][lane] = col[0];
                __syncthreads();   // (2)
                double x, x_h;
                cut_solution(x, x_h);   // x = x_{H-1}: the substitution's first row reproduces x_H
                // thomas_solve returns the unclamped vector; the state keeps min(x, max_temp)
#pragma unroll
                for (int l = 0; l < H; ++l) {
                    x = __builtin_fma(nap[l], x, col[l]);
                    col[l] = fmin(x, b_max_temp);
                }
                __syncthreads();   // (3)
            }
            // ---- end of year: this half's share of the heat content
            double total = 0.0;
#pragma unroll
            for (int l = 0; l < H; ++l) total += rho_c * b_dz * col[l];
            lds.heat[ypar][hemi][1][lane] = total;
            __syncthreads();
            return;
        }
        // ---- the upper half and the hemisphere's scalar model code (Udeb2::step with the solve cut in two)
        // The parameter block, read again (the member index is opaque per step: nothing of it is hoisted out of the step loop and
        // kept in registers across the sub-steps that do not need it)
        int32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const UdebP p = load_udeb_params(a, N, i + zero);
        const bool sh = hemi != 0;
        const double nan = __builtin_nan("");
        const bool dead = status != 0;    // the reference refuses to build this component: every output NaN
        const double erf_start = F[(size_t)n * f_stride], erf_end = F[(size_t)(n + 1) * f_stride];
        const size_t r0 = (size_t)n * N + i, r1 = r0 + (size_t)N;
        const int32_t k_full = a.win_kfull[n];
        const double part_w = a.win_partw[n];
        const double c_ground = a.land_hc ? heat_capacity_per_unit_area(p.land_hc_thick) : 0.0;
        const double c_mix = heat_capacity_per_unit_area(p.dz_mix);
        const AirMap airmap = make_air_map(p);
        const double fg_o = sh ? p.fgso : p.fgno, fg_l = sh ? p.fgsl : p.fgnl;
        const double hxf = fg_o > 1e-15 ? p.k_ns / fg_o : 0.0;
        const double inv_thresh = 1.0 / (sh ? p.t_thresh_sh : p.t_thresh_nh);
        const double ka = p.k_lo * p.amplify;
        const double w_min = p.w0 * (1.0 - p.f_var);
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
        // The half column waits in this lane's c' slots (unused between the solves) while the year's constants are formed:
        // LAMCALC and the parameter block need the registers, and what does not fit goes to scratch memory -- 60 trips to
        // memory per model step that one or two resident wavefronts cannot hide.
        double* __restrict__ ncp_lds = &lds.ncp[hemi][0][lane];
#pragma unroll
        for (int l = 0; l < H; ++l) ncp_lds[(size_t)l * 64] = col[l];
        // ---- time-varying ECS (adjusted_ecs) and the LAMCALC re-solve: both hemispheres' top wavefronts, same values
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
        double lam_o = lds.base[hemi][0][lane], lam_l = lds.base[hemi][1][lane], co2_eff = lds.base[hemi][2][lane];
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
        const double q_o = sh ? p.q2 : p.q0, q_l = sh ? p.q3 : p.q1;
        YearGeom y;
        column_geometry(y, dt_sub, p.dz, p.kappa, p.kappa_min, p.kappa_dkdt);
        {
            const double dz1 = p.dz / 2.0;
            y.dt_dzmix = dt_sub / p.dz_mix;
            y.dt_cmix = dt_sub / c_mix;
            y.dt_dzdz1 = dt_sub / (p.dz * dz1);
            y.dt_dzmixdz1 = dt_sub / (p.dz_mix * dz1);
            const double f_l = (sh ? p.sh_land : p.nh_land) / 2.0;
            const double f_o = 0.5 - f_l;
            const double den = f_o * (p.k_lo + f_l * lam_l);
            y.fb[0] = y.fb[1] = (lam_o + lam_l * p.k_lo * p.amplify * f_l / den) * y.dt_cmix;
            y.famp[0] = y.famp[1] = 1.0 + p.k_lo * f_l / den;
            y.lhc[0] = y.lhc[1] = a.land_hc ? p.k_lg * dt_sub / (c_mix * f_o) : 0.0;
        }
        const double eff_scale = eff_mode == 1 ? p.prescribed_eff : eff_mode == 2 ? p.prescribed_eff / co2_eff : 1.0;
        const double r_land = 1.0 / (lam_l * fg_l + p.k_lo);
        const double gfac = (a.land_hc && !(fg_l < 1e-15)) ? p.k_lg / (fg_l * c_ground) * dt_sub : 0.0;
        // what the sub-step loop reads of the parameters
        const double w0 = p.w0, f_var = p.f_var, pi_ratio = p.pi_ratio, max_temp = p.max_temp;
        const double fgno = p.fgno, fgnl = p.fgnl, fgso = p.fgso, fgsl = p.fgsl;
        {   // (an opaque offset: the compiler would otherwise forward the stored values to these loads and keep them in registers)
            int32_t back = 0;
            asm volatile("" : "+v"(back));
            const double* parked = ncp_lds + back;
#pragma unroll
            for (int l = 0; l < H; ++l) col[l] = parked[(size_t)l * 64];
        }
        double t_air = 0.0, t_air_o = 0.0;
        for (int32_t step_idx = 1; step_idx <= a.steps_per_year; ++step_idx) {
            const double adj = substep_forcing(erf_start, erf_end, step_idx, inv_steps, eff_scale);
            const double f_ocean = adj * q_o, f_land = adj * q_l;
            if (a.land_hc) gr = __builtin_fma(land - gr, gfac, gr);
            lds.top[hemi][0][lane] = up;
            lds.top[hemi][1][lane] = col[0];
            __syncthreads();   // (1)
            SolveScalars s;
            s.w = up;
            s.t_top = col[0];
            s.t_bottom = lds.bottom[hemi][lane];
            sweep_top<NL, FAST>(w0, pi_ratio, y, tables, a.land_hc, col, ncp_lds, hemi, s, f_ocean, hx, gr, land, ae_y);
            lds.cut[hemi][0][0][lane] = ncp_lds[(size_t)(H - 1) * 64];
            lds.cut[hemi][0][1][lane] = col[H - 1];
            __syncthreads();   // (2)
            double x, x_hm1;
            cut_solution(x_hm1, x);   // x = x_H: the substitution's first row reproduces x_{H-1}
            // thomas_solve returns the unclamped vector; the state keeps min(x, max_temp)
#pragma unroll
            for (int l = H - 1; l >= 0; --l) {
                x = __builtin_fma(ncp_lds[(size_t)l * 64], x, col[l]);
                col[l] = fmin(x, max_temp);
            }
            const double sst = col[0];
            t_air = sst_to_air(airmap, sst);
            land = land_temperature(ka, max_temp, t_air, f_land, fg_l, r_land);
            // what the other hemisphere needs of this one: air and land temperature
            const uint32_t par = n_sub & 1u;
            lds.xs[par][hemi][0][lane] = t_air;
            lds.xs[par][hemi][1][lane] = land;
            __syncthreads();   // (3)
            t_air_o = lds.xs[par][1 - hemi][0][lane];
            land_o = lds.xs[par][1 - hemi][1][lane];
            ++n_sub;
            if (fg_o > 1e-15) hx = hxf * (t_air_o - t_air);
            const double a_nh = sh ? t_air_o : t_air, l_nh = sh ? land_o : land;
            const double a_sh = sh ? t_air : t_air_o, l_sh = sh ? land : land_o;
            const double global_temp = a_nh * fgno + l_nh * fgnl + a_sh * fgso + l_sh * fgsl;
            up = fmax(w0 * (1.0 - f_var * fmin(global_temp * inv_thresh, 1.0)), w_min);   // update_upwelling
        }
        // ---- end of year
        const double sst = col[0];
        const double air = sst_to_air(airmap, sst);
        ae = fabs(sst) < 1e-15 ? p.alpha : air / sst;
        {   // calculate_ocean_heat_content: the four parts are summed on their own and added north before south, top before bottom
            double total = rho_c * p.dz_mix * sst;
#pragma unroll
            for (int l = 1; l < H; ++l) total += rho_c * p.dz * col[l];
            lds.heat[ypar][hemi][0][lane] = total;
        }
        lds.xy[ypar][hemi][0][lane] = sst;
        lds.xy[ypar][hemi][1][lane] = air;
        __syncthreads();
        const double sst_o = lds.xy[ypar][1 - hemi][0][lane];
        const double air_o = lds.xy[ypar][1 - hemi][1][lane];
        top_o = sst_o;
        const double air_nh = sh ? air_o : air, land_nh = sh ? land_o : land;
        const double air_sh = sh ? air : air_o, land_sh = sh ? land : land_o;
        const double global_temp = air_nh * fgno + land_nh * fgnl + air_sh * fgso + land_sh * fgsl;
        hist_last = global_temp * dt_year;
        if (!sh) {
            if (live) a.hist[r0] = hist_last;
            double adj_end = erf_end;
            if (eff_mode == 1) adj_end = erf_end * p.prescribed_eff;
            else if (eff_mode == 2) adj_end = erf_end * p.prescribed_eff / co2_eff;
            const double w[4] = {fgno, fgnl, fgso, fgsl};
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
            const double total = ((lds.heat[ypar][0][0][lane] + lds.heat[ypar][0][1][lane]) + lds.heat[ypar][1][0][lane]) + lds.heat[ypar][1][1][lane];
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
        if (top) {
            double* s = a.scal + (size_t)hemi * N + i;
            s[0 * N] = up; s[2 * N] = land; s[4 * N] = gr; s[6 * N] = ae; s[8 * N] = hx;
            if (hemi == 0) a.scal[(size_t)10 * N + i] = win_sum;
        }
        double* T_own = a.ocean + ((size_t)hemi * NL + (top ? 0 : H)) * N + i;
#pragma unroll
        for (int l = 0; l < H; ++l) T_own[(size_t)l * N] = col[l];
    }
};

}  // namespace udeb
}  // namespace rscm
