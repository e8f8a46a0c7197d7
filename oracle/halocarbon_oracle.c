/*
 * CPU ORACLE for rscm-magicc's HalocarbonChemistry -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of
 *   HalocarbonChemistry::solve / step_concentrations / decay_species / species_forcing /
 *   calculate_{total,fgas,montreal}_forcing / calculate_eesc
 *                                    crates/rscm-magicc/src/chemistry/halocarbon.rs:79-350
 *   HalocarbonParameters (+ Default: 23 F-gases, 18 Montreal gases), emission_to_concentration_factor
 *                                    crates/rscm-magicc/src/parameters/halocarbon.rs:46-160
 * under the stepper conventions of crates/rscm-core/src/model/runtime.rs: emissions are exogenous
 * (index n), the 41 concentrations are the component's own states (index n), the new
 * concentrations and the four aggregates (from the NEW concentrations) are written at index n+1.
 *
 * Parity pin: no golden vectors exist upstream for this component; the restatement is checked
 * against the known answers of the in-file unit tests (tests/test_oracle_halocarbon.py).  "Parity
 * unpinned" beyond those.  Sums run in species order from 0.0 (f64::sum).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stdint.h>

#define ORC_API __attribute__((visibility("default")))

#define N_FGAS 23
#define N_MONTREAL 18
#define N_SPECIES (N_FGAS + N_MONTREAL)
/* parameter vector: six globals, then seven fields per species in list order */
enum { H_BR_MULT = 0, H_CFC11_NORM, H_EESC_DELAY, H_AIR_MOLAR, H_ATM_MASS_TG, H_MIX_FRAC, H_SPECIES0 };
enum { S_LIFETIME = 0, S_RADEFF, S_CONC_PI, S_MOL_WEIGHT, S_N_CL, S_N_BR, S_FRAC_RELEASE, S_NFIELDS };
#define H_NPARAMS (H_SPECIES0 + N_SPECIES * S_NFIELDS)

ORC_API int32_t orc_halo_n_params(void) { return H_NPARAMS; }
ORC_API int32_t orc_halo_n_species(void) { return N_SPECIES; }
ORC_API int32_t orc_halo_n_fgases(void) { return N_FGAS; }

/* parameters/halocarbon.rs:95-160: lifetime, radiative efficiency, PI conc, mol. weight, nCl, nBr, release */
static const double DEFAULT_SPECIES[N_SPECIES][S_NFIELDS] = {
    {50000.0, 0.09, 0.0, 88.0, 0, 0, 0.0},  {10000.0, 0.25, 0.0, 138.0, 0, 0, 0.0}, {2600.0, 0.28, 0.0, 188.0, 0, 0, 0.0},
    {2600.0, 0.36, 0.0, 238.0, 0, 0, 0.0},  {4100.0, 0.41, 0.0, 288.0, 0, 0, 0.0},  {3100.0, 0.44, 0.0, 338.0, 0, 0, 0.0},
    {3000.0, 0.50, 0.0, 388.0, 0, 0, 0.0},  {3000.0, 0.55, 0.0, 438.0, 0, 0, 0.0},  {3200.0, 0.32, 0.0, 200.0, 0, 0, 0.0},
    {228.0, 0.18, 0.0, 70.0, 0, 0, 0.0},    {5.4, 0.11, 0.0, 52.0, 0, 0, 0.0},      {17.0, 0.359, 0.0, 252.0, 0, 0, 0.0},
    {31.0, 0.23, 0.0, 120.0, 0, 0, 0.0},    {14.0, 0.16, 0.0, 102.0, 0, 0, 0.0},    {51.0, 0.16, 0.0, 84.0, 0, 0, 0.0},
    {1.6, 0.10, 0.0, 66.0, 0, 0, 0.0},      {36.0, 0.26, 0.0, 170.0, 0, 0, 0.0},    {213.0, 0.24, 0.0, 152.0, 0, 0, 0.0},
    {7.9, 0.24, 0.0, 134.0, 0, 0, 0.0},     {8.9, 0.22, 0.0, 148.0, 0, 0, 0.0},     {569.0, 0.20, 0.0, 71.0, 0, 0, 0.0},
    {850.0, 0.57, 0.0, 146.0, 0, 0, 0.0},   {36.0, 0.20, 0.0, 102.0, 0, 0, 0.0},
    {52.0, 0.295, 0.0, 137.4, 3, 0, 0.47},  {102.0, 0.364, 0.0, 120.9, 2, 0, 0.23}, {93.0, 0.30, 0.0, 187.4, 3, 0, 0.29},
    {189.0, 0.31, 0.0, 170.9, 2, 0, 0.12},  {540.0, 0.20, 0.0, 154.5, 1, 0, 0.04},  {11.9, 0.21, 0.0, 86.5, 1, 0, 0.13},
    {9.4, 0.16, 0.0, 116.9, 2, 0, 0.34},    {18.0, 0.19, 0.0, 100.5, 1, 0, 0.17},   {5.0, 0.07, 0.0, 133.4, 3, 0, 0.67},
    {32.0, 0.174, 0.0, 153.8, 4, 0, 0.56},  {0.9, 0.004, 500.0, 50.5, 1, 0, 0.44},  {0.5, 0.028, 0.0, 84.9, 2, 0, 0.0},
    {0.5, 0.07, 0.0, 119.4, 3, 0, 0.0},     {0.8, 0.004, 5.0, 94.9, 0, 1, 0.60},    {16.0, 0.29, 0.0, 165.4, 1, 1, 0.62},
    {72.0, 0.30, 0.0, 148.9, 0, 1, 0.28},   {28.0, 0.31, 0.0, 259.8, 0, 2, 0.65},   {2.5, 0.27, 0.0, 209.8, 0, 2, 0.62},
};

ORC_API void orc_halo_default_params(double* p)
{
    p[H_BR_MULT] = 60.0; p[H_CFC11_NORM] = 0.47; p[H_EESC_DELAY] = 3.0; p[H_AIR_MOLAR] = 28.97;
    p[H_ATM_MASS_TG] = 5.133e9; p[H_MIX_FRAC] = 0.949;
    for (int s = 0; s < N_SPECIES; ++s)
        for (int f = 0; f < S_NFIELDS; ++f) p[H_SPECIES0 + s * S_NFIELDS + f] = DEFAULT_SPECIES[s][f];
}

/* emission_to_concentration_factor, parameters/halocarbon.rs:66-77 */
static double emission_factor(const double* p, double molecular_weight)
{
    const double atm_mass_g = p[H_ATM_MASS_TG] * 1e12;
    return (p[H_AIR_MOLAR] / molecular_weight) * (1e9 / atm_mass_g) * 1e12 / p[H_MIX_FRAC];
}

/* decay_species, chemistry/halocarbon.rs:79-98 */
ORC_API double orc_halo_decay_species(const double* p, int32_t s, double concentration, double emissions, double dt)
{
    const double* sp = p + H_SPECIES0 + s * S_NFIELDS;
    const double decay = exp(-dt / sp[S_LIFETIME]);
    const double emissions_ppt = emissions * emission_factor(p, sp[S_MOL_WEIGHT]);
    return concentration * decay + emissions_ppt * sp[S_LIFETIME] * (1.0 - decay);
}

/* species_forcing :100-109; aggregates :111-228 (sums in species order); out = {total, fgas, montreal, eesc} */
ORC_API void orc_halo_aggregates(const double* p, const double* conc, double out[4])
{
    double total = 0.0, fgas = 0.0, montreal = 0.0, eesc = 0.0;
    for (int s = 0; s < N_SPECIES; ++s) {
        const double* sp = p + H_SPECIES0 + s * S_NFIELDS;
        const double f = (conc[s] - sp[S_CONC_PI]) * sp[S_RADEFF] / 1000.0;
        total += f;
        if (s < N_FGAS) fgas += f; else montreal += f;
        if (sp[S_FRAC_RELEASE] > 0.0) {
            const double halogen_loading = sp[S_N_CL] + p[H_BR_MULT] * sp[S_N_BR];
            const double normalised_release = sp[S_FRAC_RELEASE] / p[H_CFC11_NORM];
            eesc += conc[s] * halogen_loading * normalised_release;
        }
    }
    out[0] = total; out[1] = fgas; out[2] = montreal; out[3] = eesc;
}

/*
 * Ensemble run: params [H_NPARAMS][N]; emissions [S][N_SPECIES][T]; bounds [T+1]; scen[N] or NULL;
 * series [N_SPECIES + 4][T][N]: concentration rows 0 hold the initial values on entry, aggregate
 * rows 0 are set to NaN; members [m0, m1).
 */
ORC_API int32_t orc_halo_run(int64_t n_members, int32_t n_times, const double* bounds, const double* params,
                             const double* emissions, const int32_t* scen, double* series, int64_t m0, int64_t m1)
{
    const int64_t vs = (int64_t)n_times * n_members;
    for (int64_t i = m0; i < m1; ++i) {
        double p[H_NPARAMS], conc[N_SPECIES], agg[4];
        for (int j = 0; j < H_NPARAMS; ++j) p[j] = params[(int64_t)j * n_members + i];
        const double* e = emissions + (int64_t)(scen ? scen[i] : 0) * N_SPECIES * n_times;
        for (int k = 0; k < 4; ++k) series[(N_SPECIES + k) * vs + i] = NAN;
        for (int s = 0; s < N_SPECIES; ++s) conc[s] = series[s * vs + i];
        for (int32_t n = 0; n + 1 < n_times; ++n) {
            const double dt = bounds[n + 1] - bounds[n];
            for (int s = 0; s < N_SPECIES; ++s) {
                conc[s] = orc_halo_decay_species(p, s, conc[s], e[(int64_t)s * n_times + n], dt);
                series[s * vs + (int64_t)(n + 1) * n_members + i] = conc[s];
            }
            orc_halo_aggregates(p, conc, agg);
            for (int k = 0; k < 4; ++k) series[(N_SPECIES + k) * vs + (int64_t)(n + 1) * n_members + i] = agg[k];
        }
    }
    return 0;
}
