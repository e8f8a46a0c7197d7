/*
 * udeb_oracle.c -- CPU restatement of rscm-magicc's ClimateUDEB.  TEST INFRASTRUCTURE ONLY
 * (same rules as rscm_oracle.c: only tests/, smoke() and bench.py's cpu_baseline may load it).
 *
 * Restates, all f64, evaluation order as written (build with -ffp-contract=off):
 *   ClimateUDEB::solve_impl             crates/rscm-magicc/src/climate/udeb/mod.rs:399-656
 *   adjusted_ecs, land temperature, SST->air   same file :302-397
 *   step_hemisphere, layer_diffusivities, update_upwelling, heat uptake, heat content
 *                                       crates/rscm-magicc/src/climate/udeb/ocean_column.rs
 *   LAMCALC                             crates/rscm-magicc/src/climate/lamcalc.rs
 *   ClimateUDEBParameters helpers       crates/rscm-magicc/src/parameters/climate_udeb.rs
 *   ClimateUDEBState::new               crates/rscm-magicc/src/climate/state.rs
 *   thomas_solve, invert_4x4            crates/rscm-core/src/utils/linear_algebra.rs
 * and the stepper around it (Model::run with an exogenous scalar ERF series, outputs at n+1).
 *
 * PARITY PIN STATUS: pinned against the MAGICC7 outputs the reference's regression tests hold
 * (tests/golden/udeb_magicc7.json, phased tolerances of tests/regression/test_ocean_udeb.py) and
 * the reference's in-file unit-test properties; bit-level agreement with the Rust binary is
 * "parity unpinned" (no Rust toolchain here).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#define NL_MAX 256

/* parameters/climate_udeb.rs constants */
static const double DIFFUSIVITY_CM2S_TO_M2YR = 3155.76;
static const double RHO_SEAWATER = 1026.0;
static const double CP_SEAWATER = 3985.0;
static const double SECONDS_PER_YEAR = 31557600.0;

/* Parameter vector layout shared with the GPU kind (include/rscm_gpu.h, RSCM_UDEB_P_*). */
enum {
    P_N_LAYERS = 0, P_MIXED_LAYER_DEPTH, P_LAYER_THICKNESS, P_KAPPA, P_KAPPA_MIN, P_KAPPA_DKDT,
    P_W_INITIAL, P_W_VARIABLE_FRACTION, P_W_THRESH_NH, P_W_THRESH_SH, P_ECS, P_RF_2XCO2, P_RLO,
    P_FEEDBACK_Q, P_FEEDBACK_CUMT, P_FEEDBACK_CUMT_PERIOD, P_K_LO, P_K_NS, P_AMPLIFY,
    P_NH_LAND, P_SH_LAND, P_DEPTH_DEPENDENT_AREA, P_ALPHA, P_GAMMA, P_POLAR_SINKING_RATIO,
    P_LAND_HC_ENABLED, P_K_LG, P_LAND_HC_THICKNESS, P_RF_REGION0, P_RF_REGION1, P_RF_REGION2,
    P_RF_REGION3, P_EFFICACY_APPLY, P_PRESCRIBED_EFFICACY, P_PROFILE_MODE, P_STEPS_PER_YEAR,
    P_MAX_TEMPERATURE, P_COUNT
};

ORC_API int32_t orc_udeb_n_params(void) { return P_COUNT; }

/* ClimateUDEBParameters::default(), climate_udeb.rs Default impl */
ORC_API void orc_udeb_default_params(double* p)
{
    p[P_N_LAYERS] = 50; p[P_MIXED_LAYER_DEPTH] = 60.0; p[P_LAYER_THICKNESS] = 100.0;
    p[P_KAPPA] = 0.75; p[P_KAPPA_MIN] = 0.1; p[P_KAPPA_DKDT] = -0.191;
    p[P_W_INITIAL] = 3.5; p[P_W_VARIABLE_FRACTION] = 0.7; p[P_W_THRESH_NH] = 8.0; p[P_W_THRESH_SH] = 8.0;
    p[P_ECS] = 3.0; p[P_RF_2XCO2] = 3.71; p[P_RLO] = 1.317;
    p[P_FEEDBACK_Q] = 7.84e-9; p[P_FEEDBACK_CUMT] = 0.08; p[P_FEEDBACK_CUMT_PERIOD] = 300.0;
    p[P_K_LO] = 1.44; p[P_K_NS] = 0.31; p[P_AMPLIFY] = 1.02;
    p[P_NH_LAND] = 0.42; p[P_SH_LAND] = 0.21; p[P_DEPTH_DEPENDENT_AREA] = 1.0;
    p[P_ALPHA] = 1.04; p[P_GAMMA] = -0.002; p[P_POLAR_SINKING_RATIO] = 0.2;
    p[P_LAND_HC_ENABLED] = 1.0; p[P_K_LG] = 0.1; p[P_LAND_HC_THICKNESS] = 300.0;
    p[P_RF_REGION0] = 1.4089; p[P_RF_REGION1] = 1.37045; p[P_RF_REGION2] = 1.43333; p[P_RF_REGION3] = 1.33257;
    p[P_EFFICACY_APPLY] = 0.0; p[P_PRESCRIBED_EFFICACY] = 1.0; p[P_PROFILE_MODE] = 2.0;
    p[P_STEPS_PER_YEAR] = 12.0; p[P_MAX_TEMPERATURE] = 25.0;
}

/* CMIP5 multi-model mean ocean temperature profiles (data: climate_udeb.rs CMIP5_PROFILE_NH/SH) */
static const double CMIP5_NH[50] = {
    1.89503822e+01, 1.58484640e+01, 1.27692938e+01, 1.11237631e+01, 9.93378544e+00, 8.89700890e+00,
    8.01173782e+00, 7.24060631e+00, 6.58022213e+00, 5.99888515e+00, 5.47700644e+00, 5.02416515e+00,
    4.62269211e+00, 4.27446032e+00, 3.95875454e+00, 3.70120311e+00, 3.47130036e+00, 3.26678157e+00,
    3.08187413e+00, 2.93045211e+00, 2.79141068e+00, 2.66952801e+00, 2.55478907e+00, 2.44816899e+00,
    2.35198379e+00, 2.26331019e+00, 2.18005610e+00, 2.10292435e+00, 2.02744699e+00, 1.95637441e+00,
    1.89118743e+00, 1.82867718e+00, 1.76954043e+00, 1.71074319e+00, 1.65469503e+00, 1.60236323e+00,
    1.55269921e+00, 1.50864816e+00, 1.47147048e+00, 1.44045138e+00, 1.41173756e+00, 1.38347185e+00,
    1.35783422e+00, 1.33539736e+00, 1.31498563e+00, 1.29516900e+00, 1.27472460e+00, 1.25263810e+00,
    1.22954643e+00, 1.20586693e+00};
static const double CMIP5_SH[50] = {
    1.62849369e+01, 1.35041571e+01, 1.10637445e+01, 9.45342350e+00, 8.30402851e+00, 7.37928152e+00,
    6.60113478e+00, 5.90550613e+00, 5.29829597e+00, 4.77080584e+00, 4.31242418e+00, 3.93976259e+00,
    3.62348270e+00, 3.35576391e+00, 3.11617875e+00, 2.93644977e+00, 2.77795982e+00, 2.63738632e+00,
    2.50925493e+00, 2.40222931e+00, 2.30221725e+00, 2.21322107e+00, 2.12794638e+00, 2.04543614e+00,
    1.96889246e+00, 1.89580762e+00, 1.82651293e+00, 1.75886285e+00, 1.69188118e+00, 1.62586987e+00,
    1.56049752e+00, 1.49373257e+00, 1.42720032e+00, 1.35796928e+00, 1.28947854e+00, 1.22542751e+00,
    1.16357803e+00, 1.10515058e+00, 1.05139232e+00, 1.00322735e+00, 9.58882809e-01, 9.15422320e-01,
    8.75476420e-01, 8.43416333e-01, 8.16016912e-01, 7.90101945e-01, 7.68699825e-01, 7.51805604e-01,
    7.36583769e-01, 7.25481987e-01};

typedef struct {
    int n, steps_per_year, land_hc, efficacy_apply, profile_mode;
    double dz_mix, dz, kappa, kappa_min, kappa_dkdt, w0, f_var, t_thresh_nh, t_thresh_sh;
    double ecs, rf_2x, rlo, fb_q, fb_cumt, fb_period, k_lo, k_ns, amplify, nh_land, sh_land, dda;
    double alpha, gamma, pi_ratio, k_lg, land_hc_thick, rf_regions[4], prescribed_eff, max_temp;
    /* derived at construction (ClimateUDEB::from_parameters) */
    double lambda_ocean, lambda_land, co2_internal_efficacy, co2_qfrac[4];
    double af_top[NL_MAX], af_bot[NL_MAX], af_diff[NL_MAX];
} udeb;

typedef struct {
    double ocean[2][NL_MAX];
    double upwelling[2], land[2], ground[2], alpha_eff[2], hemi_hx[2];
    double init_profile[2][NL_MAX];
    double polar_sinking_temp;
    double* hist_t;   /* temperature_history (T*dt) */
    double* hist_dt;
    int n_hist;
} udeb_state;

/* ---- linear algebra: rscm-core/src/utils/linear_algebra.rs ---------------------------------- */
static void thomas_solve(int n, const double* a, const double* b, const double* c, const double* d,
                         double* x)
{
    double cp[NL_MAX], dp[NL_MAX];
    cp[0] = c[0] / b[0];
    dp[0] = d[0] / b[0];
    for (int i = 1; i < n; ++i) {
        const double denom = b[i] - a[i] * cp[i - 1];
        if (i < n - 1) cp[i] = c[i] / denom;
        dp[i] = (d[i] - a[i] * dp[i - 1]) / denom;
    }
    x[n - 1] = dp[n - 1];
    for (int i = n - 2; i >= 0; --i) x[i] = dp[i] - cp[i] * x[i + 1];
}

static int invert_4x4(const double m[4][4], double inv[4][4])
{
    double aug[4][8];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 8; ++j) aug[i][j] = 0.0;
        for (int j = 0; j < 4; ++j) aug[i][j] = m[i][j];
        aug[i][i + 4] = 1.0;
    }
    for (int col = 0; col < 4; ++col) {
        int max_row = col;
        double max_val = fabs(aug[col][col]);
        for (int row = col + 1; row < 4; ++row) {
            const double val = fabs(aug[row][col]);
            if (val > max_val) { max_val = val; max_row = row; }
        }
        if (max_val < 1e-15) return 0;
        if (max_row != col)
            for (int j = 0; j < 8; ++j) { double t = aug[col][j]; aug[col][j] = aug[max_row][j]; aug[max_row][j] = t; }
        const double pivot = aug[col][col];
        for (int j = 0; j < 8; ++j) aug[col][j] /= pivot;
        for (int row = 0; row < 4; ++row) {
            if (row == col) continue;
            const double factor = aug[row][col];
            for (int j = 0; j < 8; ++j) aug[row][j] -= factor * aug[col][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i][j] = aug[i][j + 4];
    return 1;
}

/* ---- parameter helpers: parameters/climate_udeb.rs ------------------------------------------ */
static void box_fractions(const udeb* u, double* fgno, double* fgnl, double* fgso, double* fgsl)
{
    *fgnl = u->nh_land / 2.0;
    *fgno = 0.5 - *fgnl;
    *fgsl = u->sh_land / 2.0;
    *fgso = 0.5 - *fgsl;
}

static double heat_capacity_per_unit_area(double depth_m)
{
    return RHO_SEAWATER * CP_SEAWATER * depth_m / SECONDS_PER_YEAR;
}

static double ocean_area_at_depth(const udeb* u, double depth_m)
{
    static const double DEPTH[12] = {0.0, 200.0, 500.0, 1000.0, 1500.0, 2000.0, 2500.0, 3000.0, 3500.0, 4000.0, 4500.0, 5000.0};
    static const double AREA[12] = {1.0, 0.975, 0.95, 0.92, 0.91, 0.87, 0.81, 0.72, 0.55, 0.38, 0.18, 0.05};
    double hydro;
    if (depth_m <= DEPTH[0]) hydro = AREA[0];
    else if (depth_m >= DEPTH[11]) hydro = AREA[11];
    else {
        hydro = AREA[0];
        for (int i = 1; i < 12; ++i)
            if (depth_m <= DEPTH[i]) {
                const double frac = (depth_m - DEPTH[i - 1]) / (DEPTH[i] - DEPTH[i - 1]);
                hydro = AREA[i - 1] + frac * (AREA[i] - AREA[i - 1]);
                break;
            }
    }
    return 1.0 + u->dda * (hydro - 1.0);
}

static void compute_area_factors(udeb* u)
{
    for (int l = 0; l < u->n; ++l) {
        double z_top, z_bottom;
        if (l == 0) { z_top = 0.0; z_bottom = u->dz_mix; }
        else { z_top = u->dz_mix + ((double)l - 1.0) * u->dz; z_bottom = z_top + u->dz; }
        const double a_top = ocean_area_at_depth(u, z_top), a_bottom = ocean_area_at_depth(u, z_bottom);
        const double a_avg = (a_top + a_bottom) / 2.0;
        u->af_top[l] = a_top / a_avg;
        u->af_bot[l] = a_bottom / a_avg;
        u->af_diff[l] = (a_top - a_bottom) / a_avg;
    }
}

static void initial_ocean_profile(const udeb* u, int hemi, double* profile)
{
    if (u->profile_mode == 2) {
        const double* t = hemi == 0 ? CMIP5_NH : CMIP5_SH;
        for (int i = 0; i < u->n; ++i) profile[i] = i < 50 ? t[i] : t[49];
    } else {
        const double t_mix = 17.2, t_polar = 1.0;
        const double kappa = u->kappa * DIFFUSIVITY_CM2S_TO_M2YR;
        profile[0] = t_mix;
        for (int l = 1; l < u->n; ++l) {
            const double depth = ((double)l - 1.0) * u->dz + 0.5 * u->dz;
            profile[l] = t_polar + (t_mix - t_polar) * exp(-u->w0 * depth / kappa);
        }
    }
}

/* ---- LAMCALC: climate/lamcalc.rs ------------------------------------------------------------ */
static void compute_qfrac(const double rf[4], const double area[4], double qfrac[4])
{
    double rf_sum = 0.0;
    for (int i = 0; i < 4; ++i) rf_sum += rf[i] * area[i];
    if (fabs(rf_sum) <= 1e-15) { for (int i = 0; i < 4; ++i) qfrac[i] = 1.0; }
    else for (int i = 0; i < 4; ++i) qfrac[i] = rf[i] / rf_sum;
}

typedef struct { double lambda_ocean, lambda_land, inv[4][4], co2_internal_efficacy; } lam_result;

static int lamcalc(const udeb* u, double ecs, lam_result* out)
{
    double fgno, fgnl, fgso, fgsl;
    box_fractions(u, &fgno, &fgnl, &fgso, &fgsl);
    const double q = u->rf_2x, k_lo = u->k_lo, k_ns = u->k_ns, alpha = u->amplify;
    const double lam = q / ecs;
    const double fgosum = fgno + fgso, fglsum = fgnl + fgsl, fratio = fgosum / fglsum;
    const double area[4] = {fgno, fgnl, fgso, fgsl};
    double qfrac[4];
    compute_qfrac(u->rf_regions, area, qfrac);
    double lamo[42 + 2] = {0}, diff[42 + 2] = {0};
    lamo[1] = lam;
    lamo[2] = lam + 0.7;
    double dlamo = 0.7;
    int iflag = 0;
    for (int i = 2; i <= 40; ++i) {
        const double lam_l = lam + fratio * (lam - lamo[i]) / u->rlo;
        const double lam_o = lamo[i];
        const double m[4][4] = {
            {fgno * lam_o + k_lo * alpha + k_ns, -k_lo, -k_ns, 0.0},
            {-k_lo * alpha, fgnl * lam_l + k_lo, 0.0, 0.0},
            {-k_ns, 0.0, fgso * lam_o + k_lo * alpha + k_ns, -k_lo},
            {0.0, 0.0, -k_lo * alpha, fgsl * lam_l + k_lo}};
        double inv[4][4];
        if (!invert_4x4(m, inv)) return 0;
        double temps[4];
        for (int row = 0; row < 4; ++row) {
            double sum = 0.0;
            for (int col = 0; col < 4; ++col) sum += inv[row][col] * area[col] * qfrac[col];
            temps[row] = q * sum;
        }
        const double ocean_mean = (fgno * temps[0] + fgso * temps[2]) / (fgno + fgso);
        const double land_mean = (fgnl * temps[1] + fgsl * temps[3]) / (fgnl + fgsl);
        const double rlo_est = land_mean / ocean_mean;
        diff[i] = u->rlo - rlo_est;
        if (fabs(diff[i]) < 0.001) {
            out->lambda_ocean = lam_o;
            out->lambda_land = lam_l;
            memcpy(out->inv, inv, sizeof inv);
            /* calc_internal_efficacy */
            double rf_sum = 0.0;
            for (int k = 0; k < 4; ++k) rf_sum += u->rf_regions[k] * area[k];
            if (fabs(rf_sum) <= 1e-15) out->co2_internal_efficacy = 1.0;
            else {
                double t_global = 0.0;
                for (int row = 0; row < 4; ++row) {
                    double sum = 0.0;
                    for (int col = 0; col < 4; ++col) sum += inv[row][col] * area[col] * qfrac[col];
                    t_global += area[row] * (q * sum);
                }
                out->co2_internal_efficacy = t_global / ecs;
            }
            return 1;
        }
        if (diff[i] * diff[i - 1] < 0.0) iflag = 1;
        if (iflag == 0) {
            if (fabs(diff[i]) > fabs(diff[i - 1])) dlamo = -dlamo;
            lamo[i + 1] = lamo[i] + dlamo;
        } else if (diff[i] * diff[i - 1] < 0.0) {
            const double denom = diff[i] - diff[i - 1];
            if (fabs(denom) < 1e-30) lamo[i + 1] = lamo[i] + dlamo;
            else lamo[i + 1] = lamo[i] - diff[i] * (lamo[i] - lamo[i - 1]) / denom;
        } else {
            const int i2 = i - 2;
            const double denom = diff[i] - diff[i2];
            if (fabs(denom) < 1e-30) lamo[i + 1] = lamo[i] + dlamo;
            else lamo[i + 1] = lamo[i] - diff[i] * (lamo[i] - lamo[i2]) / denom;
        }
    }
    return 0;
}

/* ---- construction: ClimateUDEB::from_parameters (mod.rs:161-227) ---------------------------- */
static int udeb_init(udeb* u, const double* p)
{
    u->n = (int)p[P_N_LAYERS];
    if (u->n < 2 || u->n > NL_MAX) return 1;
    u->dz_mix = p[P_MIXED_LAYER_DEPTH]; u->dz = p[P_LAYER_THICKNESS];
    u->kappa = p[P_KAPPA]; u->kappa_min = p[P_KAPPA_MIN]; u->kappa_dkdt = p[P_KAPPA_DKDT];
    u->w0 = p[P_W_INITIAL]; u->f_var = p[P_W_VARIABLE_FRACTION];
    u->t_thresh_nh = p[P_W_THRESH_NH]; u->t_thresh_sh = p[P_W_THRESH_SH];
    u->ecs = p[P_ECS]; u->rf_2x = p[P_RF_2XCO2]; u->rlo = p[P_RLO];
    u->fb_q = p[P_FEEDBACK_Q]; u->fb_cumt = p[P_FEEDBACK_CUMT]; u->fb_period = p[P_FEEDBACK_CUMT_PERIOD];
    u->k_lo = p[P_K_LO]; u->k_ns = p[P_K_NS]; u->amplify = p[P_AMPLIFY];
    u->nh_land = p[P_NH_LAND]; u->sh_land = p[P_SH_LAND]; u->dda = p[P_DEPTH_DEPENDENT_AREA];
    u->alpha = p[P_ALPHA]; u->gamma = p[P_GAMMA]; u->pi_ratio = p[P_POLAR_SINKING_RATIO];
    u->land_hc = p[P_LAND_HC_ENABLED] != 0.0; u->k_lg = p[P_K_LG]; u->land_hc_thick = p[P_LAND_HC_THICKNESS];
    for (int i = 0; i < 4; ++i) u->rf_regions[i] = p[P_RF_REGION0 + i];
    u->efficacy_apply = (int)p[P_EFFICACY_APPLY]; u->prescribed_eff = p[P_PRESCRIBED_EFFICACY];
    u->profile_mode = (int)p[P_PROFILE_MODE]; u->steps_per_year = (int)p[P_STEPS_PER_YEAR];
    u->max_temp = p[P_MAX_TEMPERATURE];
    if (!isfinite(u->prescribed_eff) || u->prescribed_eff <= 0.0) return 2;
    if (u->steps_per_year < 1) return 3;
    lam_result r;
    if (!lamcalc(u, u->ecs, &r)) return 4; /* "LAMCALC iteration failed to converge" */
    u->lambda_ocean = r.lambda_ocean;
    u->lambda_land = r.lambda_land;
    u->co2_internal_efficacy = r.co2_internal_efficacy;
    double fgno, fgnl, fgso, fgsl;
    box_fractions(u, &fgno, &fgnl, &fgso, &fgsl);
    const double area[4] = {fgno, fgnl, fgso, fgsl};
    compute_qfrac(u->rf_regions, area, u->co2_qfrac);
    compute_area_factors(u);
    return 0;
}

/* ClimateUDEBState::new (state.rs) */
static void state_init(const udeb* u, udeb_state* s, int max_hist)
{
    memset(s, 0, sizeof *s);
    s->upwelling[0] = s->upwelling[1] = u->w0;
    s->alpha_eff[0] = s->alpha_eff[1] = u->alpha;
    initial_ocean_profile(u, 0, s->init_profile[0]);
    initial_ocean_profile(u, 1, s->init_profile[1]);
    s->polar_sinking_temp = 1.0;
    s->hist_t = (double*)malloc(sizeof(double) * (size_t)(max_hist > 0 ? max_hist : 1));
    s->hist_dt = (double*)malloc(sizeof(double) * (size_t)(max_hist > 0 ? max_hist : 1));
    s->n_hist = 0;
}

/* ---- physics helpers (mod.rs:253-397) -------------------------------------------------------- */
static double sst_to_air(const udeb* u, double sst)
{
    const double alpha = u->alpha, gamma = u->gamma;
    const double t_star = fabs(gamma) > 1e-15 ? -(alpha - 1.0) / (2.0 * gamma) : INFINITY;
    if (sst < t_star) return alpha * sst + gamma * sst * sst;
    const double delta_max = alpha * t_star + gamma * t_star * t_star - t_star;
    return sst + delta_max;
}

static double land_temperature(const udeb* u, double ocean_temp, double land_forcing,
                               double land_fraction, double lambda_land)
{
    const double numerator = land_forcing * land_fraction + u->k_lo * u->amplify * ocean_temp;
    const double denominator = lambda_land * land_fraction + u->k_lo;
    return fmin(numerator / denominator, u->max_temp);
}

static void efficacy_and_qfrac(const udeb* u, double erf, double co2_efficacy, double out[4])
{
    double adj = erf;
    if (u->efficacy_apply == 1) adj = erf * u->prescribed_eff;
    else if (u->efficacy_apply == 2 && isfinite(co2_efficacy) && co2_efficacy > 0.0)
        adj = erf * u->prescribed_eff / co2_efficacy;
    for (int i = 0; i < 4; ++i) out[i] = adj * u->co2_qfrac[i];
}

static double adjusted_ecs(const udeb* u, double global_forcing, const udeb_state* s)
{
    const double cumt_2x = u->ecs * u->fb_period;
    double cum_t = 0.0;
    if (s->n_hist > 0) {
        double years_remaining = u->fb_period, sum = 0.0;
        for (int i = s->n_hist - 1; i >= 0; --i) {
            if (years_remaining <= 0.0) break;
            const double dt = s->hist_dt[i];
            if (dt <= years_remaining) { sum += s->hist_t[i]; years_remaining -= dt; }
            else { sum += s->hist_t[i] * (years_remaining / dt); years_remaining = 0.0; }
        }
        cum_t = sum;
    }
    const double cumt_factor = fabs(cumt_2x) > 1e-15 ? 1.0 + u->fb_cumt * (cum_t - cumt_2x) / cumt_2x : 1.0;
    const double q_factor = 1.0 + u->fb_q * (fmax(global_forcing, 0.0) - u->rf_2x);
    return u->ecs * cumt_factor * q_factor;
}

/* ---- ocean column: ocean_column.rs ---------------------------------------------------------- */
static double step_hemisphere(const udeb* u, udeb_state* s, int hemi, double forcing, double dt,
                              double lambda_ocean, double lambda_land, double hemi_hx,
                              double ground_temp, double alpha_eff)
{
    const int n = u->n;
    double kappas[NL_MAX], a[NL_MAX], b[NL_MAX], c[NL_MAX], d[NL_MAX], x[NL_MAX];
    {
        const double total_depth = u->dz_mix + ((double)n - 1.0) * u->dz;
        const double t_top = s->ocean[hemi][0], t_bottom = s->ocean[hemi][n - 1];
        const double kappa_min_m2yr = u->kappa_min * DIFFUSIVITY_CM2S_TO_M2YR;
        for (int l = 0; l < n - 1; ++l) {
            const double depth = u->dz_mix + (double)l * u->dz;
            const double relative_depth = depth / total_depth;
            const double k = ((1.0 - relative_depth) * u->kappa_dkdt * (t_top - t_bottom) + u->kappa) * DIFFUSIVITY_CM2S_TO_M2YR;
            kappas[l] = fmax(k, kappa_min_m2yr);
        }
    }
    const double w = s->upwelling[hemi], dz = u->dz, dz_mix = u->dz_mix, pi_ratio = u->pi_ratio;
    const double* af_top = u->af_top; const double* af_bot = u->af_bot; const double* af_diff = u->af_diff;
    const double c_mix = heat_capacity_per_unit_area(u->dz_mix);
    for (int i = 0; i < n; ++i) a[i] = b[i] = c[i] = d[i] = 0.0;
    const double f_l_hemi = hemi == 0 ? u->nh_land / 2.0 : u->sh_land / 2.0;
    const double f_o_hemi = 0.5 - f_l_hemi;
    const double denominator = f_o_hemi * (u->k_lo + f_l_hemi * lambda_land);
    const double term_feedback = alpha_eff / c_mix * (lambda_ocean + lambda_land * u->k_lo * u->amplify * f_l_hemi / denominator);
    const double dz1 = dz / 2.0;
    const double term_diff = kappas[0] / (dz_mix * dz1) * dt;
    const double term_upwell = w / dz_mix * dt;
    const double forcing_amp = 1.0 + u->k_lo * f_l_hemi / denominator;
    b[0] = 1.0 + term_feedback * dt * af_top[0] + term_diff * af_bot[0] + term_upwell * pi_ratio * af_bot[0];
    c[0] = -(term_diff + term_upwell) * af_bot[0];
    d[0] = s->ocean[hemi][0] + (forcing * forcing_amp + hemi_hx) / c_mix * dt * af_top[0];
    if (u->land_hc) {
        const double land_temp = s->land[hemi];
        d[0] -= u->k_lg * (land_temp - ground_temp) / (c_mix * f_o_hemi) * dt * af_top[0];
    }
    for (int i = 1; i < n - 1; ++i) {
        const double dz_up = i == 1 ? dz1 : dz;
        const double term_diff_up = kappas[i - 1] / (dz * dz_up) * dt;
        const double term_diff_down = kappas[i] / (dz * dz) * dt;
        const double term_upwell_layer = w / dz * dt;
        a[i] = -term_diff_up * af_top[i];
        b[i] = 1.0 + term_diff_up * af_top[i] + term_diff_down * af_bot[i] + term_upwell_layer * af_top[i];
        c[i] = -(term_diff_down + term_upwell_layer) * af_bot[i];
        d[i] = s->ocean[hemi][i] + pi_ratio * term_upwell_layer * s->ocean[hemi][0] * af_diff[i];
    }
    {
        const double term_diff_up = kappas[n - 2] / (dz * dz) * dt;
        const double term_upwell_bottom = w / dz * dt;
        a[n - 1] = -term_diff_up * af_top[n - 1];
        b[n - 1] = 1.0 + (term_diff_up + term_upwell_bottom) * af_top[n - 1];
        d[n - 1] = s->ocean[hemi][n - 1] + pi_ratio * term_upwell_bottom * s->ocean[hemi][0] * af_top[n - 1];
    }
    const double delta_w = w - u->w0;
    if (fabs(delta_w) > 1e-15) {
        const double* init = s->init_profile[hemi];
        const double t_polar = s->polar_sinking_temp;
        const double dt_per_dz_mix = dt / dz_mix;
        d[0] += dt_per_dz_mix * delta_w * (init[1] - t_polar) * af_bot[0];
        const double dt_per_dz = dt / dz;
        for (int i = 1; i < n - 1; ++i) {
            d[i] += dt_per_dz * delta_w * (init[i + 1] * af_bot[i] - init[i] * af_top[i]);
            d[i] += dt_per_dz * delta_w * t_polar * af_diff[i];
        }
        d[n - 1] += dt_per_dz * delta_w * (t_polar - init[n - 1]) * af_top[n - 1];
    }
    thomas_solve(n, a, b, c, d, x);
    for (int i = 0; i < n; ++i) s->ocean[hemi][i] = fmin(x[i], u->max_temp);
    return s->ocean[hemi][0];
}

static void update_upwelling(const udeb* u, udeb_state* s, double global_temp)
{
    const double w_0 = u->w0, f_var = u->f_var, w_min = w_0 * (1.0 - f_var);
    const double w_nh = w_0 * (1.0 - f_var * fmin(global_temp / u->t_thresh_nh, 1.0));
    s->upwelling[0] = fmax(w_nh, w_min);
    const double w_sh = w_0 * (1.0 - f_var * fmin(global_temp / u->t_thresh_sh, 1.0));
    s->upwelling[1] = fmax(w_sh, w_min);
}

/* One ClimateUDEB::solve_impl (mod.rs:399-656).  out: surface temperature[4], heat uptake,
 * ocean heat content, sst. */
static void udeb_solve(const udeb* u, udeb_state* s, double t_current, double t_next,
                       double erf_start, double erf_end, const double prev_temp[4], double out[7])
{
    const double steps = (double)u->steps_per_year;
    if (s->ocean[0][0] == 0.0 && prev_temp[0] != 0.0) { /* warm start */
        s->ocean[0][0] = prev_temp[0];
        s->ocean[1][0] = prev_temp[2];
        s->land[0] = prev_temp[1];
        s->land[1] = prev_temp[3];
        s->ground[0] = s->land[0];
        s->ground[1] = s->land[1];
    }
    const double dt_year = t_next - t_current;
    const double dt_sub = dt_year / (double)u->steps_per_year;
    const double erf_mid = (erf_start + erf_end) / 2.0;
    const double adj_ecs = adjusted_ecs(u, erf_mid, s);
    double lam_o = u->lambda_ocean, lam_l = u->lambda_land, co2_eff = u->co2_internal_efficacy;
    if (fabs(adj_ecs - u->ecs) > 1e-10) {
        lam_result r;
        if (lamcalc(u, adj_ecs, &r)) { lam_o = r.lambda_ocean; lam_l = r.lambda_land; co2_eff = r.co2_internal_efficacy; }
    }
    double fgno, fgnl, fgso, fgsl;
    box_fractions(u, &fgno, &fgnl, &fgso, &fgsl);
    const double c_ground = u->land_hc ? heat_capacity_per_unit_area(u->land_hc_thick) : 0.0;
    const double alpha_eff_nh = s->alpha_eff[0], alpha_eff_sh = s->alpha_eff[1];
    for (int step_idx = 1; step_idx <= u->steps_per_year; ++step_idx) {
        const double frac = (double)step_idx / steps;
        const double erf = erf_start + frac * (erf_end - erf_start);
        double forcing[4];
        efficacy_and_qfrac(u, erf, co2_eff, forcing);
        if (u->land_hc) {
            const double fl[2] = {fgnl, fgsl};
            for (int hemi = 0; hemi < 2; ++hemi) {
                if (fl[hemi] < 1e-15) continue;
                const double flux = u->k_lg * (s->land[hemi] - s->ground[hemi]);
                s->ground[hemi] += flux / (fl[hemi] * c_ground) * dt_sub;
            }
        }
        const double nh_ground = s->ground[0], sh_ground = s->ground[1];
        const double sst_nh = step_hemisphere(u, s, 0, forcing[0], dt_sub, lam_o, lam_l, s->hemi_hx[0], nh_ground, alpha_eff_nh);
        const double sst_sh = step_hemisphere(u, s, 1, forcing[2], dt_sub, lam_o, lam_l, s->hemi_hx[1], sh_ground, alpha_eff_sh);
        const double t_air_nho = sst_to_air(u, sst_nh), t_air_sho = sst_to_air(u, sst_sh);
        s->land[0] = land_temperature(u, t_air_nho, forcing[1], fgnl, lam_l);
        s->land[1] = land_temperature(u, t_air_sho, forcing[3], fgsl, lam_l);
        if (fgno > 1e-15) s->hemi_hx[0] = u->k_ns / fgno * (t_air_sho - t_air_nho);
        if (fgso > 1e-15) s->hemi_hx[1] = u->k_ns / fgso * (t_air_nho - t_air_sho);
        const double global_temp = t_air_nho * fgno + s->land[0] * fgnl + t_air_sho * fgso + s->land[1] * fgsl;
        update_upwelling(u, s, global_temp);
    }
    const double sst_nh = s->ocean[0][0], sst_sh = s->ocean[1][0];
    s->alpha_eff[0] = fabs(sst_nh) < 1e-15 ? u->alpha : sst_to_air(u, sst_nh) / sst_nh;
    s->alpha_eff[1] = fabs(sst_sh) < 1e-15 ? u->alpha : sst_to_air(u, sst_sh) / sst_sh;
    const double st[4] = {sst_to_air(u, sst_nh), s->land[0], sst_to_air(u, sst_sh), s->land[1]};
    const double global_temp = st[0] * fgno + st[1] * fgnl + st[2] * fgso + st[3] * fgsl;
    s->hist_t[s->n_hist] = global_temp * dt_year;
    s->hist_dt[s->n_hist] = dt_year;
    s->n_hist++;
    double forcing_end[4];
    efficacy_and_qfrac(u, erf_end, co2_eff, forcing_end);
    {   /* calculate_heat_uptake */
        const double weights[4] = {fgno, fgnl, fgso, fgsl};
        const double lambdas[4] = {lam_o, lam_l, lam_o, lam_l};
        double q_global = 0.0, feedback_global = 0.0;
        for (int i = 0; i < 4; ++i) { q_global += weights[i] * forcing_end[i]; feedback_global += weights[i] * lambdas[i] * st[i]; }
        out[4] = q_global - feedback_global;
    }
    {   /* calculate_ocean_heat_content */
        const double rho_c = RHO_SEAWATER * CP_SEAWATER;
        double total = 0.0;
        for (int hemi = 0; hemi < 2; ++hemi) {
            total += rho_c * u->dz_mix * s->ocean[hemi][0];
            for (int l = 1; l < u->n; ++l) total += rho_c * u->dz * s->ocean[hemi][l];
        }
        out[5] = total / 2.0;
    }
    out[0] = st[0]; out[1] = st[1]; out[2] = st[2]; out[3] = st[3];
    out[6] = (sst_nh + sst_sh) / 2.0;
}

/*
 * Ensemble run: Model::run over the axis `bounds` (T+1 entries) with an exogenous scalar ERF
 * series on the model axis.  params [P_COUNT][N]; erf [S][T]; outputs, each [T][N]:
 *   st0..st3 (Surface Temperature FourBox: NH ocean, NH land, SH ocean, SH land; index 0 = the
 *   initial value given in st_init[4]), heat_uptake, ohc, sst (index 0 = NaN, no initial value).
 * status[i] = 0 ok, else the construction error (4 = LAMCALC did not converge).
 */
ORC_API int orc_udeb_run(int64_t N, int32_t T, const double* bounds, const double* params,
                         int32_t S, const double* erf, const int32_t* scen, const double st_init[4],
                         double* st0, double* st1, double* st2, double* st3, double* heat_uptake,
                         double* ohc, double* sst, int32_t* status, int64_t i0, int64_t i1)
{
    double* out[7] = {st0, st1, st2, st3, heat_uptake, ohc, sst};
    for (int64_t i = i0; i < i1; ++i) {
        double p[P_COUNT];
        for (int j = 0; j < P_COUNT; ++j) p[j] = params[(size_t)j * N + i];
        udeb u;
        const int rc = udeb_init(&u, p);
        status[i] = rc;
        for (int k = 0; k < 4; ++k) out[k][i] = st_init[k];
        for (int k = 4; k < 7; ++k) out[k][i] = NAN;
        if (rc) {
            for (int32_t n = 1; n < T; ++n)
                for (int k = 0; k < 7; ++k) out[k][(size_t)n * N + i] = NAN;
            continue;
        }
        const int32_t s_id = scen ? scen[i] : 0;
        if (s_id < 0 || s_id >= S) return 3;
        const double* F = erf + (size_t)s_id * T;
        udeb_state s;
        state_init(&u, &s, T);
        for (int32_t n = 0; n < T - 1; ++n) {
            const double prev[4] = {out[0][(size_t)n * N + i], out[1][(size_t)n * N + i],
                                    out[2][(size_t)n * N + i], out[3][(size_t)n * N + i]};
            double o[7];
            /* at_start = F[n]; at_end = F[n+1] (exists for every step of run()) */
            udeb_solve(&u, &s, bounds[n], bounds[n + 1], F[n], F[n + 1], prev, o);
            for (int k = 0; k < 7; ++k) out[k][(size_t)(n + 1) * N + i] = o[k];
        }
        free(s.hist_t);
        free(s.hist_dt);
    }
    return 0;
}

/* Diagnostics for the reference's unit-test properties. */
ORC_API int orc_udeb_lamcalc(const double* p, double ecs, double out[4])
{
    udeb u;
    const int rc = udeb_init(&u, p);
    if (rc) return rc;
    lam_result r;
    if (!lamcalc(&u, ecs, &r)) return 4;
    out[0] = r.lambda_ocean; out[1] = r.lambda_land; out[2] = r.co2_internal_efficacy;
    out[3] = u.co2_qfrac[0];
    return 0;
}

ORC_API int orc_udeb_area_factors(const double* p, double* af_top, double* af_bot, double* af_diff)
{
    udeb u;
    const int rc = udeb_init(&u, p);
    if (rc) return rc;
    for (int l = 0; l < u.n; ++l) { af_top[l] = u.af_top[l]; af_bot[l] = u.af_bot[l]; af_diff[l] = u.af_diff[l]; }
    return 0;
}

ORC_API double orc_udeb_sst_to_air(const double* p, double sst)
{
    udeb u;
    if (udeb_init(&u, p)) return NAN;
    return sst_to_air(&u, sst);
}
