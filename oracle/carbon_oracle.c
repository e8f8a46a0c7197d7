/*
 * CPU ORACLE for rscm-magicc's CO2Budget and TerrestrialCarbon -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of
 *   CO2Budget::solve / solve_budget               crates/rscm-magicc/src/carbon/budget.rs:96-190
 *   TerrestrialCarbon::solve / solve_pools        crates/rscm-magicc/src/carbon/terrestrial.rs:103-330
 *   their parameter structs and derived turnover times
 *                                                 crates/rscm-magicc/src/parameters/{co2_budget,terrestrial_carbon}.rs
 * under the stepper conventions of crates/rscm-core/src/model/runtime.rs: inputs are exogenous
 * (index n), the pools / the concentration are the component's own state (index n), everything is
 * written at index n+1, dt = bounds[n+1] - bounds[n].
 *
 * Parity pin: no golden vectors exist upstream for these components; the restatement is checked
 * against the known answers of the in-file unit tests and crates/rscm-magicc/tests/conservation.rs
 * (tests/test_oracle_carbon.py).  "Parity unpinned" beyond those.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stdint.h>

#define ORC_API __attribute__((visibility("default")))

enum { CARBON_BUDGET = 9, CARBON_TERRESTRIAL = 10 };

/* CO2BudgetParameters: gtc_per_ppm, co2_pi */
enum { B_GTC_PER_PPM = 0, B_CO2_PI, B_NPARAMS };
/* TerrestrialCarbonParameters field order (booleans as 0/1) */
enum { L_NPP_PI = 0, L_CO2_PI, L_BETA, L_NPP_TS, L_RESP_TS, L_DET_TS, L_SOIL_TS, L_HUM_TS, L_PLANT_PI,
       L_DET_PI, L_SOIL_PI, L_HUM_PI, L_RESP_PI, L_F_NPP_PLANT, L_F_NPP_DET, L_F_PLANT_DET, L_F_DET_SOIL,
       L_F_SOIL_HUM, L_ENABLE_FERT, L_ENABLE_TEMP, L_NPARAMS };

ORC_API int32_t orc_carbon_n_params(int32_t kind) { return kind == CARBON_BUDGET ? B_NPARAMS : kind == CARBON_TERRESTRIAL ? L_NPARAMS : -1; }
ORC_API int32_t orc_carbon_n_inputs(int32_t kind) { return kind == CARBON_BUDGET ? 4 : kind == CARBON_TERRESTRIAL ? 3 : -1; }
/* stored series: states first, then outputs */
ORC_API int32_t orc_carbon_n_states(int32_t kind) { return kind == CARBON_BUDGET ? 1 : kind == CARBON_TERRESTRIAL ? 4 : -1; }
ORC_API int32_t orc_carbon_n_outputs(int32_t kind) { return kind == CARBON_BUDGET ? 2 : kind == CARBON_TERRESTRIAL ? 1 : -1; }

ORC_API void orc_carbon_default_params(int32_t kind, double* p)
{
    if (kind == CARBON_BUDGET) {
        p[B_GTC_PER_PPM] = 2.123; p[B_CO2_PI] = 278.0;
    } else if (kind == CARBON_TERRESTRIAL) {
        static const double d[L_NPARAMS] = {66.27, 278.0, 0.6486, 0.0107, 0.0685, 0.1358, 0.1541, 0.05,
                                            884.86, 92.77, 1681.53, 836.0, 12.26, 0.4483, 0.3998, 0.9989, 0.3, 0.1,
                                            1.0, 1.0};
        for (int j = 0; j < L_NPARAMS; ++j) p[j] = d[j];
    }
}

/* carbon/budget.rs:96-127; in = {fossil, landuse, terrestrial flux, ocean flux};
 * out = {co2_next, net_emissions, airborne_fraction} */
ORC_API void orc_co2_budget_solve(const double* p, const double* in, double co2, double dt, double out[3])
{
    const double total_emissions = in[0] + in[1];
    const double total_uptake = in[2] + in[3];
    const double net_to_atm = total_emissions - total_uptake;
    const double delta = (net_to_atm * dt) / p[B_GTC_PER_PPM];
    out[0] = co2 + delta;
    out[1] = net_to_atm;
    out[2] = total_emissions > 0.0 ? net_to_atm / total_emissions : 0.0;
}

/* parameters/terrestrial_carbon.rs:103-168 */
static double frac_npp_to_soil(const double* p)
{
    const double f = 1.0 - p[L_F_NPP_PLANT] - p[L_F_NPP_DET];
    return fmax(f, 0.0);
}
static double net_flux_to_plant_pi(const double* p) { return p[L_F_NPP_PLANT] * p[L_NPP_PI] - p[L_RESP_PI]; }
static double tau_plant_pi(const double* p)
{
    const double nf = net_flux_to_plant_pi(p);
    return nf > 1e-10 ? p[L_PLANT_PI] / nf : 100.0;
}
static double tau_detritus_pi(const double* p)
{
    const double flux = p[L_F_NPP_DET] * p[L_NPP_PI] + p[L_F_PLANT_DET] * net_flux_to_plant_pi(p);
    return flux > 1e-10 ? p[L_DET_PI] / flux : 3.0;
}
static double tau_soil_pi(const double* p)
{
    const double nfp = net_flux_to_plant_pi(p);
    const double flux_detritus_out = p[L_DET_PI] / tau_detritus_pi(p);
    const double flux = frac_npp_to_soil(p) * p[L_NPP_PI] + (1.0 - p[L_F_PLANT_DET]) * nfp + p[L_F_DET_SOIL] * flux_detritus_out;
    return flux > 1e-10 ? p[L_SOIL_PI] / flux : 50.0;
}
static double tau_humus_pi(const double* p)
{
    const double flux_soil_out = p[L_SOIL_PI] / tau_soil_pi(p);
    const double flux = p[L_F_SOIL_HUM] * flux_soil_out;
    return flux > 1e-10 ? p[L_HUM_PI] / flux : 1000.0;
}
ORC_API void orc_terrestrial_taus(const double* p, double out[4])
{
    out[0] = tau_plant_pi(p); out[1] = tau_detritus_pi(p); out[2] = tau_soil_pi(p); out[3] = tau_humus_pi(p);
}

/* carbon/terrestrial.rs:44-63 */
static double fertilization_factor(const double* p, double co2)
{
    if (p[L_ENABLE_FERT] == 0.0 || co2 <= 0.0) return 1.0;
    return fmax(1.0 + p[L_BETA] * log(co2 / p[L_CO2_PI]), 0.1);
}
static double temperature_factor(const double* p, double temperature, double sensitivity)
{
    if (p[L_ENABLE_TEMP] == 0.0) return 1.0;
    return exp(sensitivity * temperature);
}
/* carbon/terrestrial.rs:82-100 */
static void implicit_pool_step(double pool, double tau, double flux_in, double temp_factor, double dt,
                               double* new_pool, double* turnover)
{
    const double k_eff = temp_factor / tau;
    const double half_k = 0.5 * k_eff * dt;
    double np_ = ((1.0 - half_k) * pool + flux_in * dt) / (1.0 + half_k);
    np_ = fmax(np_, 0.0);
    *new_pool = np_;
    *turnover = 0.5 * k_eff * (pool + np_);
}

/* carbon/terrestrial.rs:103-190 solve_pools; in = {co2, temperature, landuse};
 * out = {plant, detritus, soil, humus, net_flux} */
ORC_API void orc_terrestrial_solve_pools(const double* p, const double* in, const double pools[4], double dt, double out[5])
{
    const double co2 = in[0], temperature = in[1], landuse = in[2];
    const double npp = p[L_NPP_PI] * fertilization_factor(p, co2) * temperature_factor(p, temperature, p[L_NPP_TS]);
    const double respiration = p[L_RESP_PI] * fertilization_factor(p, co2) * temperature_factor(p, temperature, p[L_RESP_TS]);
    const double tf_det = temperature_factor(p, temperature, p[L_DET_TS]);
    const double tf_soil = temperature_factor(p, temperature, p[L_SOIL_TS]);
    const double tf_hum = temperature_factor(p, temperature, p[L_HUM_TS]);
    double np_, to_plant, nd, to_det, ns, to_soil, nh, to_hum;
    const double flux_in_plant = npp * p[L_F_NPP_PLANT] - respiration - landuse;
    implicit_pool_step(pools[0], tau_plant_pi(p), flux_in_plant, 1.0, dt, &np_, &to_plant);
    const double flux_in_det = npp * p[L_F_NPP_DET] + p[L_F_PLANT_DET] * to_plant;
    implicit_pool_step(pools[1], tau_detritus_pi(p), flux_in_det, tf_det, dt, &nd, &to_det);
    const double npp_to_soil = npp * frac_npp_to_soil(p);
    const double plant_to_soil = (1.0 - p[L_F_PLANT_DET]) * to_plant;
    const double det_to_soil = p[L_F_DET_SOIL] * to_det;
    implicit_pool_step(pools[2], tau_soil_pi(p), npp_to_soil + plant_to_soil + det_to_soil, tf_soil, dt, &ns, &to_soil);
    implicit_pool_step(pools[3], tau_humus_pi(p), p[L_F_SOIL_HUM] * to_soil, tf_hum, dt, &nh, &to_hum);
    const double det_to_atm = (1.0 - p[L_F_DET_SOIL]) * to_det;
    const double soil_to_atm = (1.0 - p[L_F_SOIL_HUM]) * to_soil;
    const double total_resp = respiration + det_to_atm + soil_to_atm + to_hum;
    out[0] = np_; out[1] = nd; out[2] = ns; out[3] = nh;
    out[4] = npp - total_resp - landuse;
}

/*
 * Ensemble run: params [P][N]; inputs [S][n_inputs][T]; bounds [T+1]; scen[N] or NULL;
 * series [n_states + n_outputs][T][N]: state rows 0 hold the initial values on entry, output rows 0
 * are set to NaN; members [m0, m1).
 */
ORC_API int32_t orc_carbon_run(int32_t kind, int64_t n_members, int32_t n_times, const double* bounds,
                               const double* params, const double* inputs, const int32_t* scen, double* series,
                               int64_t m0, int64_t m1)
{
    const int P = orc_carbon_n_params(kind), NI = orc_carbon_n_inputs(kind);
    const int NS = orc_carbon_n_states(kind), NO = orc_carbon_n_outputs(kind);
    if (P < 0) return 1;
    const int64_t vs = (int64_t)n_times * n_members;
    for (int64_t i = m0; i < m1; ++i) {
        double p[L_NPARAMS], in[4], st[4], out[5];
        for (int j = 0; j < P; ++j) p[j] = params[(int64_t)j * n_members + i];
        const double* c = inputs + (int64_t)(scen ? scen[i] : 0) * NI * n_times;
        for (int o = 0; o < NO; ++o) series[(NS + o) * vs + i] = NAN;
        for (int32_t n = 0; n + 1 < n_times; ++n) {
            const double dt = bounds[n + 1] - bounds[n];
            for (int k = 0; k < NI; ++k) in[k] = c[(int64_t)k * n_times + n];
            for (int k = 0; k < NS; ++k) st[k] = series[k * vs + (int64_t)n * n_members + i];
            if (kind == CARBON_BUDGET) orc_co2_budget_solve(p, in, st[0], dt, out);
            else orc_terrestrial_solve_pools(p, in, st, dt, out);
            for (int k = 0; k < NS + NO; ++k) series[k * vs + (int64_t)(n + 1) * n_members + i] = out[k];
        }
    }
    return 0;
}
