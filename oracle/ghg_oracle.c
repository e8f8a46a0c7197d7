/*
 * CPU ORACLE for rscm-magicc's GhgForcing component -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of
 *   GhgForcing::calculate_forcings / solve        crates/rscm-magicc/src/forcing/ghg.rs:119-345
 *   GhgForcingParameters (+ Default)              crates/rscm-magicc/src/parameters/ghg_forcing.rs
 * under the stepper conventions of crates/rscm-core/src/model/runtime.rs: the three
 * concentrations are exogenous (read at index n), the three ERFs are written at index n+1 and
 * index 0 stays NaN.
 *
 * Parity pin: the MAGICC7 outputs the reference's regression tests hold for this component
 * (tests/regression/data/ghg_forcing/{01,02}*.csv -> tests/golden/ghg_forcing_magicc7.json), at
 * the reference's own tolerance (rtol 1e-5, atol 1e-6; tests/regression/test_ghg_forcing.py), and
 * the known answers of the in-file unit tests (forcing/ghg.rs:368-727).  Bit-level agreement
 * with the Rust binary is unpinned (no Rust toolchain in the image); powf/ln/sqrt are libm's
 * pow/log/sqrt here, which is what rustc lowers them to on Linux.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stdint.h>

#define ORC_API __attribute__((visibility("default")))

/* parameter vector, GhgForcingParameters field order; method: 0 = Ipcctar, 1 = Olbl */
enum {
    G_METHOD = 0, G_CO2_PI, G_CH4_PI, G_N2O_PI, G_DELQ2X, G_CH4_RADEFF, G_N2O_RADEFF,
    G_CO2_A1, G_CO2_B1, G_CO2_C1, G_CO2_D1, G_CH4_A3, G_CH4_B3, G_CH4_D3,
    G_N2O_A2, G_N2O_B2, G_N2O_C2, G_N2O_D2, G_ADJ_CO2, G_ADJ_CH4, G_ADJ_N2O, G_NPARAMS
};

ORC_API int32_t orc_ghg_n_params(void) { return G_NPARAMS; }

/* parameters/ghg_forcing.rs Default */
ORC_API void orc_ghg_default_params(double* p)
{
    p[G_METHOD] = 1.0;
    p[G_CO2_PI] = 278.0; p[G_CH4_PI] = 722.0; p[G_N2O_PI] = 270.0;
    p[G_DELQ2X] = 3.71; p[G_CH4_RADEFF] = 0.036; p[G_N2O_RADEFF] = 0.12;
    p[G_CO2_A1] = -2.4785e-7; p[G_CO2_B1] = 7.5906e-4; p[G_CO2_C1] = -2.1492e-3; p[G_CO2_D1] = 5.2;
    p[G_CH4_A3] = -8.9603e-5; p[G_CH4_B3] = -1.2462e-4; p[G_CH4_D3] = 0.045;
    p[G_N2O_A2] = -3.4197e-4; p[G_N2O_B2] = 2.5455e-4; p[G_N2O_C2] = -2.4357e-4; p[G_N2O_D2] = 0.14;
    p[G_ADJ_CO2] = 1.05; p[G_ADJ_CH4] = 0.86; p[G_ADJ_N2O] = 1.0;
}

/* forcing/ghg.rs:119-129 overlap_f (Myhre et al. 1998) */
static double overlap_f(double ch4_ppb, double n2o_ppb)
{
    const double mn = ch4_ppb * n2o_ppb;
    return 0.47 * log(1.0 + 2.01e-5 * pow(mn, 0.75) + 5.31e-15 * ch4_ppb * pow(mn, 1.52));
}

/* forcing/ghg.rs:164-170 */
static double co2_ipcctar(const double* p, double co2)
{
    const double alpha = p[G_DELQ2X] / log(2.0);
    return alpha * log(co2 / p[G_CO2_PI]);
}

/* forcing/ghg.rs:172-185 */
static double ch4_ipcctar(const double* p, double ch4)
{
    const double direct = p[G_CH4_RADEFF] * (sqrt(ch4) - sqrt(p[G_CH4_PI]));
    const double overlap = overlap_f(ch4, p[G_N2O_PI]) - overlap_f(p[G_CH4_PI], p[G_N2O_PI]);
    return direct - overlap;
}

/* forcing/ghg.rs:187-200 */
static double n2o_ipcctar(const double* p, double n2o)
{
    const double direct = p[G_N2O_RADEFF] * (sqrt(n2o) - sqrt(p[G_N2O_PI]));
    const double overlap = overlap_f(p[G_CH4_PI], n2o) - overlap_f(p[G_CH4_PI], p[G_N2O_PI]);
    return direct - overlap;
}

/* forcing/ghg.rs:210-240 */
static double co2_olbl(const double* p, double co2, double n2o)
{
    const double co2_pi = p[G_CO2_PI];
    const double delta_co2 = co2 - co2_pi;
    const double n2o_overlap = p[G_CO2_C1] * sqrt(n2o);
    const double c_max = co2_pi - p[G_CO2_B1] / (2.0 * p[G_CO2_A1]);
    double alpha;
    if (co2 >= c_max)
        alpha = -p[G_CO2_B1] * p[G_CO2_B1] / (4.0 * p[G_CO2_A1]) + p[G_CO2_D1] + n2o_overlap;
    else if (co2 <= co2_pi)
        alpha = p[G_CO2_D1] + n2o_overlap;
    else
        alpha = p[G_CO2_A1] * delta_co2 * delta_co2 + p[G_CO2_B1] * delta_co2 + p[G_CO2_D1] + n2o_overlap;
    return alpha * log(co2 / co2_pi);
}

/* forcing/ghg.rs:248-254 */
static double ch4_olbl(const double* p, double ch4, double n2o)
{
    const double coeff = p[G_CH4_A3] * sqrt(ch4) + p[G_CH4_B3] * sqrt(n2o) + p[G_CH4_D3];
    return coeff * (sqrt(ch4) - sqrt(p[G_CH4_PI]));
}

/* forcing/ghg.rs:260-269 */
static double n2o_olbl(const double* p, double co2, double ch4, double n2o)
{
    const double coeff = p[G_N2O_A2] * sqrt(co2) + p[G_N2O_B2] * sqrt(n2o) + p[G_N2O_C2] * sqrt(ch4) + p[G_N2O_D2];
    return coeff * (sqrt(n2o) - sqrt(p[G_N2O_PI]));
}

/* forcing/ghg.rs:272-290 calculate_forcings: out = {co2_erf, ch4_erf, n2o_erf} */
ORC_API void orc_ghg_forcings(const double* p, double co2, double ch4, double n2o, double out[3])
{
    const int olbl = p[G_METHOD] != 0.0;
    const double co2_raw = olbl ? co2_olbl(p, co2, n2o) : co2_ipcctar(p, co2);
    const double ch4_raw = olbl ? ch4_olbl(p, ch4, n2o) : ch4_ipcctar(p, ch4);
    const double n2o_raw = olbl ? n2o_olbl(p, co2, ch4, n2o) : n2o_ipcctar(p, n2o);
    out[0] = co2_raw * p[G_ADJ_CO2];
    out[1] = ch4_raw * p[G_ADJ_CH4];
    out[2] = n2o_raw * p[G_ADJ_N2O];
}

/*
 * Ensemble run: params [G_NPARAMS][N] (SoA), conc [S][3][T] (CO2, CH4, N2O on the model axis),
 * scen[N] or NULL, outputs [T][N] each (row 0 is left NaN by the caller's initialisation, rows
 * 1..T-1 are written), members [m0, m1).
 */
ORC_API void orc_ghg_run(int64_t n_members, int32_t n_times, const double* params, int32_t n_scen,
                         const double* conc, const int32_t* scen, double* co2_erf, double* ch4_erf,
                         double* n2o_erf, int64_t m0, int64_t m1)
{
    (void)n_scen;
    for (int64_t i = m0; i < m1; ++i) {
        double p[G_NPARAMS];
        for (int j = 0; j < G_NPARAMS; ++j) p[j] = params[(int64_t)j * n_members + i];
        const double* c = conc + (int64_t)(scen ? scen[i] : 0) * 3 * n_times;
        co2_erf[i] = NAN; ch4_erf[i] = NAN; n2o_erf[i] = NAN;
        for (int32_t n = 0; n + 1 < n_times; ++n) {
            double out[3];
            orc_ghg_forcings(p, c[n], c[n_times + n], c[2 * n_times + n], out);
            const int64_t r = (int64_t)(n + 1) * n_members + i;
            co2_erf[r] = out[0];
            ch4_erf[r] = out[1];
            n2o_erf[r] = out[2];
        }
    }
}
