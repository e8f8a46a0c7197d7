/*
 * CPU ORACLE for rscm-magicc's OceanCarbon -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of
 *   OceanCarbon::solve_impl / solve_ocean / calculate_delta_dic / calculate_flux
 *                                    crates/rscm-magicc/src/carbon/ocean.rs:73-215
 *   OceanCarbonParameters (gfdl_3d / bern_2d / hilda presets, irf, scale_irf,
 *   delta_pco2_from_dic, ocean_pco2, dic_conversion_factor, gas_exchange_rate)
 *                                    crates/rscm-magicc/src/parameters/ocean_carbon.rs:88-300
 * under the stepper conventions of crates/rscm-core/src/model/runtime.rs: CO2 and SST are
 * exogenous (index n), pCO2 and the cumulative uptake are the component's own state (index n),
 * the flux history is the component's internal state (solve_with_state) and persists across
 * steps; outputs at index n+1.
 *
 * The impulse response is evaluated afresh for every (pulse, time) pair exactly as the reference
 * does -- O(history) exp-sums per monthly sub-step -- so keep ensembles small here.
 *
 * Parity pin: no golden vectors exist upstream for this component; the restatement is checked
 * against the known answers of the in-file unit tests, tests/conservation.rs and
 * tests/carbon_cycle_physics.rs (tests/test_oracle_ocean.py).  "Parity unpinned" beyond those
 * (powi(k) is restated as the square-and-multiply product LLVM expands it to).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define ORC_API __attribute__((visibility("default")))

#define PPM_TO_GTC 2.124                    /* carbon/ocean.rs:26 */
#define MICROMOL_PER_PPM_M3_PER_KG 1.72e17  /* parameters/ocean_carbon.rs:4 */

/* parameter vector; [u] rows select tables or loop bounds and are uniform over an ensemble */
enum { OC_MODEL = 0 /* [u] 0 GFDL3D, 1 BERN2D, 2 HILDA */, OC_CO2_PI, OC_PCO2_PI, OC_GAS_SCALE, OC_GAS_TAU,
       OC_TEMP_SENS, OC_IRF_SCALE /* [u] */, OC_MLD, OC_AREA, OC_SST_PI, OC_STEPS /* [u] */, OC_MAX_HIST /* [u] */,
       OC_SWITCH /* [u] */, OC_OFF0, OC_COEF0 = OC_OFF0 + 5, OC_ENABLE_TEMP = OC_COEF0 + 5, OC_NPARAMS };

ORC_API int32_t orc_ocean_n_params(void) { return OC_NPARAMS; }

typedef struct { int poly; int n; double c[8]; double tau[8]; } irf_form;

static void irf_forms(int model, irf_form* early, irf_form* late)
{
    static const irf_form gfdl_e = {1, 7, {1.0, -2.2617, 14.002, -48.770, 82.986, -67.527, 21.037}, {0}};
    static const irf_form gfdl_l = {0, 6, {0.01481, 0.019439, 0.038344, 0.066485, 0.24966, 0.70367},
                                    {1.0e10, 347.55, 65.359, 15.281, 2.3488, 0.70177}};
    static const irf_form bern_e = {0, 6, {0.058648, 0.07515, 0.079338, 0.41413, 0.24845, 0.12429},
                                    {1.0e10, 9.6218, 9.2364, 0.7603, 0.16294, 0.0032825}};
    static const irf_form bern_l = {0, 6, {0.01369, 0.012456, 0.026933, 0.026994, 0.036608, 0.06738},
                                    {1.0e10, 331.54, 107.57, 38.946, 11.677, 10.515}};
    static const irf_form hilda_e = {0, 5, {0.12935, 0.24093, 0.24071, 0.17003, 0.21898},
                                     {1.0e10, 4.9792, 0.96083, 0.26936, 0.034569}};
    static const irf_form hilda_l = {0, 6, {0.022936, 0.035549, 0.037820, 0.089318, 0.13963, 0.24278},
                                     {1.0e10, 232.30, 68.736, 18.601, 5.2528, 1.2679}};
    *early = model == 1 ? bern_e : model == 2 ? hilda_e : gfdl_e;
    *late = model == 1 ? bern_l : model == 2 ? hilda_l : gfdl_l;
}

/* presets: parameters/ocean_carbon.rs:88-196 */
ORC_API void orc_ocean_default_params(int32_t model, double* p)
{
    static const double off[5] = {1.5568, 7.4706, 1.2748, 2.4491, 1.5468};
    static const double coef[5] = {-0.013993, -0.20207, -0.12015, -0.12639, -0.15326};
    p[OC_MODEL] = model; p[OC_CO2_PI] = 278.0; p[OC_PCO2_PI] = 278.0; p[OC_GAS_SCALE] = 1.833492;
    p[OC_GAS_TAU] = model == 1 ? 7.46 : model == 2 ? 9.06 : 7.66;
    p[OC_TEMP_SENS] = 0.03717879; p[OC_IRF_SCALE] = 0.9492864;
    p[OC_MLD] = model == 1 ? 50.0 : model == 2 ? 75.0 : 50.9;
    p[OC_AREA] = model == 1 ? 3.5375e14 : model == 2 ? 3.62e14 : 3.55e14;
    p[OC_SST_PI] = model == 1 ? 18.2997 : model == 2 ? 18.1716 : 17.7;
    p[OC_STEPS] = 12.0; p[OC_MAX_HIST] = 6000.0;
    p[OC_SWITCH] = model == 1 ? 9.9 : model == 2 ? 2.0 : 1.0;
    for (int i = 0; i < 5; ++i) { p[OC_OFF0 + i] = off[i]; p[OC_COEF0 + i] = coef[i]; }
    p[OC_ENABLE_TEMP] = 1.0;
}

/* IrfForm::evaluate, parameters/ocean_carbon.rs:41-71 */
static double irf_eval(const irf_form* f, double t)
{
    if (f->poly) {
        double r = 0.0;
        for (int i = f->n - 1; i >= 0; --i) r = r * t + f->c[i];
        return r;
    }
    double s = 0.0;
    for (int i = 0; i < f->n; ++i) s += f->c[i] * exp(-t / f->tau[i]);
    return s;
}

/* OceanCarbonParameters::irf + scale_irf, parameters/ocean_carbon.rs:202-216 */
ORC_API double orc_ocean_irf(const double* p, double t)
{
    irf_form e, l;
    irf_forms((int)p[OC_MODEL], &e, &l);
    const double raw = t < p[OC_SWITCH] ? irf_eval(&e, t) : irf_eval(&l, t);
    const double f = p[OC_IRF_SCALE];
    return (raw * f) / (raw * f + 1.0 - raw);
}

/* delta_pco2_from_dic, parameters/ocean_carbon.rs:218-235 */
ORC_API double orc_ocean_delta_pco2_from_dic(const double* p, double d)
{
    const double d2 = d * d, d3 = d * d2, d4 = d2 * d2, d5 = d * d4;
    const double g[5] = {d, d2 * 1e-3, -d3 * 1e-5, d4 * 1e-7, -d5 * 1e-10};
    double s = 0.0;
    for (int i = 0; i < 5; ++i) s += (p[OC_OFF0 + i] + p[OC_COEF0 + i] * p[OC_SST_PI]) * g[i];
    return s;
}

/* ocean_pco2, parameters/ocean_carbon.rs:237-245 */
ORC_API double orc_ocean_pco2(const double* p, double delta_pco2_dic, double delta_sst)
{
    const double tf = p[OC_ENABLE_TEMP] != 0.0 ? exp(p[OC_TEMP_SENS] * delta_sst) : 1.0;
    return (p[OC_PCO2_PI] + delta_pco2_dic) * tf;
}

typedef struct { double* buf; int64_t cap, head, len; } ring; /* the VecDeque flux_history */

/* solve_ocean, carbon/ocean.rs:116-160; out = {pco2, cumulative, flux} */
static void solve_ocean(const double* p, ring* h, double co2_atm, double delta_sst, double pco2_initial,
                        double cumulative_initial, double dt, double out[3])
{
    const int64_t steps = (int64_t)p[OC_STEPS], max_hist = (int64_t)p[OC_MAX_HIST];
    const double dt_month = dt / (double)steps;
    const double k = p[OC_GAS_SCALE] / (p[OC_GAS_TAU] * 12.0);
    const double dic_conv = MICROMOL_PER_PPM_M3_PER_KG / (p[OC_MLD] * p[OC_AREA]);
    double pco2 = pco2_initial, cumulative = cumulative_initial, total = 0.0;
    for (int64_t s = 0; s < steps; ++s) {
        const double flux_ppm = k * (co2_atm - pco2);
        if (h->len == h->cap) { h->head = (h->head + 1) % h->cap; h->len -= 1; } /* room for the push */
        h->buf[(h->head + h->len) % h->cap] = flux_ppm;
        h->len += 1;
        if (h->len > max_hist) { h->head = (h->head + 1) % h->cap; h->len -= 1; } /* pop_front */
        const double flux_gtc_yr = flux_ppm * 12.0 * PPM_TO_GTC;
        total += flux_gtc_yr / (double)steps;
        cumulative += flux_gtc_yr * dt_month;
        /* calculate_delta_dic: oldest to newest, pulse i happened (n-1-i) months ago */
        double integral = 0.0;
        const int64_t n = h->len;
        for (int64_t i = 0; i < n; ++i) {
            const double t_since = (double)(n - 1 - i) * (1.0 / 12.0);
            integral += h->buf[(h->head + i) % h->cap] * orc_ocean_irf(p, t_since) * 1.0;
        }
        const double delta_dic = n == 0 ? 0.0 : integral * dic_conv;
        pco2 = orc_ocean_pco2(p, orc_ocean_delta_pco2_from_dic(p, delta_dic), delta_sst);
    }
    out[0] = pco2; out[1] = cumulative; out[2] = total;
}

/*
 * Ensemble run: params [OC_NPARAMS][N]; inputs [S][2][T] (CO2, SST anomaly); bounds [T+1];
 * series [3][T][N] = pCO2, cumulative uptake, flux -- rows 0 of the two states hold the initial
 * values on entry, flux row 0 is set to NaN; members [m0, m1).
 */
ORC_API int32_t orc_ocean_run(int64_t n_members, int32_t n_times, const double* bounds, const double* params,
                              const double* inputs, const int32_t* scen, double* series, int64_t m0, int64_t m1)
{
    const int64_t vs = (int64_t)n_times * n_members;
    for (int64_t i = m0; i < m1; ++i) {
        double p[OC_NPARAMS], out[3];
        for (int j = 0; j < OC_NPARAMS; ++j) p[j] = params[(int64_t)j * n_members + i];
        const int64_t max_hist = (int64_t)p[OC_MAX_HIST];
        if (max_hist < 0 || (int64_t)p[OC_STEPS] < 0) return 1;
        ring h = {malloc((size_t)(max_hist + 1) * sizeof(double)), max_hist + 1, 0, 0};
        if (!h.buf) return 2;
        const double* c = inputs + (int64_t)(scen ? scen[i] : 0) * 2 * n_times;
        series[2 * vs + i] = NAN;
        for (int32_t n = 0; n + 1 < n_times; ++n) {
            const int64_t r = (int64_t)n * n_members + i;
            solve_ocean(p, &h, c[n], c[(int64_t)n_times + n], series[r], series[vs + r], bounds[n + 1] - bounds[n], out);
            for (int k = 0; k < 3; ++k) series[k * vs + r + n_members] = out[k];
        }
        free(h.buf);
    }
    return 0;
}

/* one stand-alone solve_ocean call chain for the unit tests: runs `n_calls` calls with constant
 * inputs starting from an empty history, out[n_calls][3] */
ORC_API int32_t orc_ocean_solve_repeated(const double* p, double co2_atm, double delta_sst, double pco2_initial,
                                         double cumulative_initial, double dt, int32_t n_calls, double* out)
{
    const int64_t max_hist = (int64_t)p[OC_MAX_HIST];
    ring h = {malloc((size_t)(max_hist + 1) * sizeof(double)), max_hist + 1, 0, 0};
    if (!h.buf) return 2;
    double pco2 = pco2_initial, cum = cumulative_initial;
    for (int32_t c = 0; c < n_calls; ++c) {
        solve_ocean(p, &h, co2_atm, delta_sst, pco2, cum, dt, out + 3 * c);
        pco2 = out[3 * c];
        cum = out[3 * c + 1];
    }
    free(h.buf);
    return 0;
}
