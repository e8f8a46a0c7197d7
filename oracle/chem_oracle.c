/*
 * CPU ORACLE for rscm-magicc's CH4Chemistry and N2OChemistry -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of
 *   CH4Chemistry::solve / solve_concentration / prather_iteration
 *                                   crates/rscm-magicc/src/chemistry/ch4.rs:121-330
 *   N2OChemistry::solve / solve_concentration / iteration
 *                                   crates/rscm-magicc/src/chemistry/n2o.rs:96-260
 *   their parameter structs          crates/rscm-magicc/src/parameters/{ch4,n2o}_chemistry.rs
 * under the stepper conventions of crates/rscm-core/src/model/runtime.rs: emissions and
 * temperature are exogenous (index n); the concentration is the component's own state (at_start =
 * index n, previous() = index n-1, at_offset(-k) = index n-k, each falling back as the reference
 * does when the index would be negative); concentration and lifetime are written at index n+1.
 *
 * Parity pin: the reference holds no golden vectors for these components (its only full-chain
 * regression scenario is xfail upstream); the restatement is checked against the known answers
 * of the in-file unit tests (tests/test_oracle_chem.py).  "Parity unpinned" beyond those.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stdint.h>

#define ORC_API __attribute__((visibility("default")))

enum { CHEM_CH4 = 7, CHEM_N2O = 8 };
#define PRATHER_ITERATIONS 4

/* CH4ChemistryParameters field order (booleans as 0/1) */
enum { M_PI_ = 0, M_NATURAL, M_TAU_OH, M_TAU_SOIL, M_TAU_STRAT, M_TAU_TROP_CL, M_SELF_FB, M_OH_SCALE,
       M_OH_NOX, M_OH_CO, M_OH_NMVOC, M_TEMP_SENS, M_INCL_TEMP, M_INCL_EMIS, M_PPB_TO_TG, M_NOX_REF,
       M_CO_REF, M_NMVOC_REF, M_NPARAMS };
/* N2OChemistryParameters field order (strat_delay as a double holding an integer) */
enum { N_PI_ = 0, N_NATURAL, N_TAU, N_LIFETIME_FB, N_STRAT_DELAY, N_PPB_TO_TG, N_NPARAMS };

ORC_API int32_t orc_chem_n_params(int32_t kind) { return kind == CHEM_CH4 ? M_NPARAMS : kind == CHEM_N2O ? N_NPARAMS : -1; }
ORC_API int32_t orc_chem_n_inputs(int32_t kind) { return kind == CHEM_CH4 ? 5 : kind == CHEM_N2O ? 1 : -1; }

ORC_API void orc_chem_default_params(int32_t kind, double* p)
{
    if (kind == CHEM_CH4) { /* parameters/ch4_chemistry.rs Default */
        static const double d[M_NPARAMS] = {722.0, 209.0, 9.3, 150.0, 120.0, 200.0, -0.32, 0.72, 0.0042, -0.000105,
                                            -0.000315, 0.0316, 1.0, 1.0, 2.75, 0.0, 0.0, 0.0};
        for (int j = 0; j < M_NPARAMS; ++j) p[j] = d[j];
    } else if (kind == CHEM_N2O) { /* parameters/n2o_chemistry.rs Default */
        static const double d[N_NPARAMS] = {270.0, 11.0, 139.275, -0.04, 1.0, 4.79};
        for (int j = 0; j < N_NPARAMS; ++j) p[j] = d[j];
    }
}

/* parameters/ch4_chemistry.rs tau_other */
static double ch4_tau_other(const double* p)
{
    return 1.0 / (1.0 / p[M_TAU_SOIL] + 1.0 / p[M_TAU_STRAT] + 1.0 / p[M_TAU_TROP_CL]);
}

/* chemistry/ch4.rs:121-215 solve_concentration; out = {new concentration, total lifetime} */
ORC_API void orc_ch4_solve_concentration(const double* p, double ch4_prev, double ch4_current, double emissions,
                                         double temperature, double nox, double co, double nmvoc, double out[2])
{
    const double total_emissions = emissions + p[M_NATURAL];
    const double burden_prev = ch4_prev * p[M_PPB_TO_TG];
    const double burden_reference = p[M_PI_] * p[M_PPB_TO_TG];
    const double delta_nox = nox - p[M_NOX_REF], delta_co = co - p[M_CO_REF], delta_nmvoc = nmvoc - p[M_NMVOC_REF];
    const double gamma = p[M_OH_SCALE];
    double base; /* calculate_base_lifetime_factor */
    if (p[M_INCL_EMIS] == 0.0) {
        base = p[M_TAU_OH];
    } else {
        const double exponent = -gamma * (p[M_OH_NOX] * delta_nox + p[M_OH_CO] * delta_co + p[M_OH_NMVOC] * delta_nmvoc);
        base = p[M_TAU_OH] * exp(exponent);
    }
    const double x = -gamma * p[M_SELF_FB];
    const double tau_other = ch4_tau_other(p);
    double burden = ch4_current * p[M_PPB_TO_TG];
    double delta_burden = 0.0, tau_oh = p[M_TAU_OH];
    int have_delta = 0;
    for (int it = 0; it < PRATHER_ITERATIONS; ++it) {
        const double burden_mean = (burden + burden_prev) / 2.0;
        const double ratio = fmax(burden_mean / burden_reference, 1.0);
        tau_oh = base * pow(ratio, x);
        if (have_delta && !(fabs(burden_prev) < 1e-10)) /* apply_iteration_correction(tau, db_prev, burden_previous) */
            tau_oh = tau_oh * (1.0 - 0.5 * x * delta_burden / burden_prev);
        if (!(p[M_INCL_TEMP] == 0.0 || fabs(temperature) < 1e-10)) { /* apply_temperature_feedback */
            const double delta_t = fmax(temperature, 0.0);
            tau_oh = p[M_TAU_OH] / (p[M_TAU_OH] / tau_oh + p[M_TEMP_SENS] * delta_t);
        }
        delta_burden = total_emissions - burden_mean / tau_oh - burden_mean / tau_other;
        have_delta = 1;
        burden = burden_prev + delta_burden;
    }
    out[0] = burden / p[M_PPB_TO_TG];
    out[1] = 1.0 / (1.0 / tau_oh + 1.0 / tau_other);
}

/* chemistry/n2o.rs:96-146 solve_concentration; out = {new concentration, effective lifetime} */
ORC_API void orc_n2o_solve_concentration(const double* p, double n2o_prev, double n2o_current, double n2o_lagged,
                                         double emissions, double dt, double out[2])
{
    const double total_emissions = emissions + p[N_NATURAL];
    const double burden_prev = n2o_prev * p[N_PPB_TO_TG];
    const double burden_lagged = n2o_lagged * p[N_PPB_TO_TG];
    const double burden_reference = p[N_PI_] * p[N_PPB_TO_TG];
    double burden = n2o_current * p[N_PPB_TO_TG];
    double tau_eff = p[N_TAU];
    for (int it = 0; it < PRATHER_ITERATIONS; ++it) {
        const double burden_mid = (burden_prev + burden) / 2.0;
        const double ratio = fmax(burden_mid / burden_reference, 1.0);
        tau_eff = p[N_TAU] * pow(ratio, p[N_LIFETIME_FB]);
        const double rate = total_emissions - burden_lagged / tau_eff;
        burden = burden_prev + rate * dt;
    }
    out[0] = burden / p[N_PPB_TO_TG];
    out[1] = tau_eff;
}

/*
 * Ensemble run: params [P][N]; inputs [S][n_inputs][T]; bounds [T+1]; scen[N] or NULL; conc and
 * lifetime [T][N] with conc row 0 holding the initial concentration on entry (lifetime row 0 is
 * set to NaN); members [m0, m1).
 */
ORC_API int32_t orc_chem_run(int32_t kind, int64_t n_members, int32_t n_times, const double* bounds,
                             const double* params, const double* inputs, const int32_t* scen, double* conc,
                             double* lifetime, int64_t m0, int64_t m1)
{
    const int P = orc_chem_n_params(kind), NI = orc_chem_n_inputs(kind);
    if (P < 0) return 1;
    for (int64_t i = m0; i < m1; ++i) {
        double p[M_NPARAMS], out[2];
        for (int j = 0; j < P; ++j) p[j] = params[(int64_t)j * n_members + i];
        const double* c = inputs + (int64_t)(scen ? scen[i] : 0) * NI * n_times;
        lifetime[i] = NAN;
#define C(n) conc[(int64_t)(n) * n_members + i]
        for (int32_t n = 0; n + 1 < n_times; ++n) {
            const double cur = C(n);
            const double prev = n == 0 ? cur : C(n - 1); /* previous().unwrap_or(current) */
            if (kind == CHEM_CH4) {
                orc_ch4_solve_concentration(p, prev, cur, c[n], c[(int64_t)n_times + n], c[(int64_t)2 * n_times + n],
                                            c[(int64_t)3 * n_times + n], c[(int64_t)4 * n_times + n], out);
            } else {
                /* n2o.rs:203-218: delay = strat_delay.max(1); at_offset(-delay) else prev;
                 * at_offset(-(delay+1)) else the former */
                int64_t delay = (int64_t)p[N_STRAT_DELAY];
                if (delay < 1) delay = 1;
                const double t_delay = (int64_t)n - delay >= 0 ? C(n - delay) : prev;
                const double t_delay_m1 = (int64_t)n - delay - 1 >= 0 ? C(n - delay - 1) : t_delay;
                const double lagged = (t_delay + t_delay_m1) / 2.0;
                orc_n2o_solve_concentration(p, prev, cur, lagged, c[n], bounds[n + 1] - bounds[n], out);
            }
            C(n + 1) = out[0];
            lifetime[(int64_t)(n + 1) * n_members + i] = out[1];
        }
#undef C
    }
    return 0;
}
