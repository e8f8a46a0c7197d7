/* Sanitizer driver for the CPU oracle (test infrastructure): exercises every exported function
 * on small inputs under -fsanitize=address,undefined.  GPU ASan is not available on the pool, so
 * memory-safety checking happens here, on the code the GPU results are compared against. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int orc_bounds_from_values(const double*, int32_t, double*);
int32_t orc_rk4_nsteps(double, double, double);
int orc_rk4_endtime_ok(double, double, double);
void orc_two_layer_solve(const double*, double, double, double, double, double*, double*, double*);
int orc_two_layer_run(int64_t, int32_t, const double*, const double*, int32_t, const double*,
                      const int32_t*, int, double, int32_t, int32_t, double*, double*, int64_t, int64_t);
void orc_carbon_cycle_solve(const double*, double, double, double, double, double, double*);
double orc_co2_erf(double, double, double);
double orc_aggregate_sum(const double*, int32_t);
int orc_coupled_run(int64_t, int32_t, const double*, const double*, int32_t, const double*,
                    const int32_t*, double, double, int32_t, int32_t, double*, double*, double*,
                    double*, double*, double*, double*, int64_t, int64_t);
int orc_gaussian_loglik(int64_t, int32_t, const double* const*, int32_t, const int32_t*,
                        const int32_t*, const double*, const double*, int, double*, int64_t, int64_t);

int32_t orc_udeb_n_params(void);
void orc_udeb_default_params(double*);
int orc_udeb_run(int64_t, int32_t, const double*, const double*, int32_t, const double*, const int32_t*,
                 const double*, double*, double*, double*, double*, double*, double*, double*, int32_t*,
                 int64_t, int64_t);

/* rscm-magicc / rscm-components restatements */
int32_t orc_ghg_n_params(void);
void orc_ghg_default_params(double*);
void orc_ghg_run(int64_t, int32_t, const double*, int32_t, const double*, const int32_t*, double*, double*, double*,
                 int64_t, int64_t);
int32_t orc_pointwise_n_params(int32_t), orc_pointwise_n_inputs(int32_t), orc_pointwise_n_outputs(int32_t);
void orc_pointwise_default_params(int32_t, double*);
int32_t orc_pointwise_run(int32_t, int64_t, int32_t, const double*, const double*, const int32_t*, double*, int64_t, int64_t);
int32_t orc_chem_n_params(int32_t), orc_chem_n_inputs(int32_t);
void orc_chem_default_params(int32_t, double*);
int32_t orc_chem_run(int32_t, int64_t, int32_t, const double*, const double*, const double*, const int32_t*, double*,
                     double*, int64_t, int64_t);
int32_t orc_carbon_n_params(int32_t), orc_carbon_n_inputs(int32_t), orc_carbon_n_states(int32_t), orc_carbon_n_outputs(int32_t);
void orc_carbon_default_params(int32_t, double*);
int32_t orc_carbon_run(int32_t, int64_t, int32_t, const double*, const double*, const double*, const int32_t*, double*,
                       int64_t, int64_t);
int32_t orc_ocean_n_params(void);
void orc_ocean_default_params(int32_t, double*);
int32_t orc_ocean_run(int64_t, int32_t, const double*, const double*, const double*, const int32_t*, double*, int64_t, int64_t);
int32_t orc_halo_n_params(void), orc_halo_n_species(void);
void orc_halo_default_params(double*);
int32_t orc_halo_run(int64_t, int32_t, const double*, const double*, const double*, const int32_t*, double*, int64_t, int64_t);

/* params [P][n] from one default vector, member i scaled by (1 + 0.01 i) in row `vary` */
static double* spread(const double* d, int P, int n, int vary)
{
    double* p = malloc(sizeof(double) * P * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < P; ++j) p[j * n + i] = d[j] * (j == vary ? 1.0 + 0.01 * i : 1.0);
    return p;
}

static int magicc_kinds(void)
{
    enum { NM = 5, TM = 16 };
    double b[TM + 1], d[320];
    int32_t scen[NM];
    for (int i = 0; i <= TM; ++i) b[i] = 1900.0 + i;
    for (int i = 0; i < NM; ++i) scen[i] = i % 2;
    double* in = malloc(sizeof(double) * 2 * 41 * TM);  /* the widest input block: 2 scenarios x 41 rows */
    for (int i = 0; i < 2 * 41 * TM; ++i) in[i] = 300.0 + (i % 97);
    double* out = malloc(sizeof(double) * 45 * TM * NM);  /* the widest series block */
    double* p;
    /* GhgForcing, both methods */
    for (int method = 0; method < 2; ++method) {
        orc_ghg_default_params(d);
        d[0] = method;
        p = spread(d, orc_ghg_n_params(), NM, 1);
        orc_ghg_run(NM, TM, p, 2, in, scen, out, out + TM * NM, out + 2 * TM * NM, 0, NM);
        if (!isnan(out[0]) || !isfinite(out[(TM - 1) * NM + NM - 1])) return 20 + method;
        free(p);
    }
    /* the five pointwise kinds */
    const int pw[5] = {4, 5, 6, 13, 14};
    for (int k = 0; k < 5; ++k) {
        orc_pointwise_default_params(pw[k], d);
        p = spread(d, orc_pointwise_n_params(pw[k]), NM, 0);
        if (orc_pointwise_run(pw[k], NM, TM, p, in, scen, out, 0, NM)) return 30 + k;
        free(p);
    }
    /* chemistry: concentration state in out, lifetime behind it */
    for (int kind = 7; kind <= 8; ++kind) {
        orc_chem_default_params(kind, d);
        p = spread(d, orc_chem_n_params(kind), NM, 2);
        for (int i = 0; i < NM; ++i) out[i] = kind == 7 ? 800.0 : 275.0;
        if (orc_chem_run(kind, NM, TM, b, p, in, scen, out, out + TM * NM, 0, NM)) return 40 + kind;
        free(p);
    }
    /* carbon: states then outputs */
    for (int kind = 9; kind <= 10; ++kind) {
        orc_carbon_default_params(kind, d);
        p = spread(d, orc_carbon_n_params(kind), NM, 0);
        for (int s_ = 0; s_ < orc_carbon_n_states(kind); ++s_)
            for (int i = 0; i < NM; ++i) out[s_ * TM * NM + i] = 300.0 + 100.0 * s_;
        if (orc_carbon_run(kind, NM, TM, b, p, in, scen, out, 0, NM)) return 50 + kind;
        free(p);
    }
    /* ocean: the history ring at its bound (max_history_months = 30 < 15 x 12 months) and unbounded */
    for (int sc = 0; sc < 2; ++sc)  /* the second input row is an SST anomaly in K */
        for (int t_ = 0; t_ < TM; ++t_) in[(sc * 2 + 1) * TM + t_] = 0.05 * t_;
    for (int bounded = 0; bounded < 2; ++bounded) {
        orc_ocean_default_params(bounded ? 2 : 0, d);
        if (bounded) d[11] = 30.0;
        p = spread(d, orc_ocean_n_params(), NM, 4);
        for (int i = 0; i < NM; ++i) { out[i] = 290.0; out[TM * NM + i] = 0.0; }
        if (orc_ocean_run(NM, TM, b, p, in, scen, out, 0, NM)) return 60 + bounded;
        if (!isfinite(out[(TM - 1) * NM])) return 62;
        free(p);
    }
    /* halocarbon: 41 concentration states then four aggregates */
    orc_halo_default_params(d);
    p = spread(d, orc_halo_n_params(), NM, 6);
    for (int s_ = 0; s_ < orc_halo_n_species(); ++s_)
        for (int i = 0; i < NM; ++i) out[s_ * TM * NM + i] = 5.0;
    if (orc_halo_run(NM, TM, b, p, in, scen, out, 0, NM)) return 70;
    if (!isfinite(out[44 * TM * NM + (TM - 1) * NM])) return 71;
    free(p);
    free(in);
    free(out);
    return 0;
}

#define N 37
#define T 41
int main(void)
{
    double v[T], b[T + 1], F[2 * T], *p = malloc(10 * N * sizeof(double));
    int32_t scen[N];
    for (int i = 0; i < T; ++i) { v[i] = 1750.0 + i; F[i] = 0.1 * i; F[T + i] = -0.05 * i; }
    if (orc_bounds_from_values(v, T, b)) return 1;
    for (int i = 0; i < N; ++i) {
        const double base[10] = {1.0 + 0.01 * i, 0.001 * i, 1.2, 0.7, 8.0, 100.0, 25.0, 278.0, 0.05, 3.7};
        for (int j = 0; j < 10; ++j) p[j * N + i] = base[j];
        scen[i] = i % 2;
    }
    double* s[7];
    for (int k = 0; k < 7; ++k) {
        s[k] = malloc(sizeof(double) * T * N);
        for (int i = 0; i < T * N; ++i) s[k][i] = NAN;
        for (int i = 0; i < N; ++i) s[k][i] = k == 2 ? 278.0 : 0.0;
    }
    if (orc_two_layer_run(N, T, b, p, 2, F, scen, 0, 0.1, 0, T - 1, s[0], s[1], 0, N)) return 2;
    if (orc_two_layer_run(N, T, b, p, 2, F, scen, 1, 0.1, 0, T - 1, s[0], s[1], 3, N - 3)) return 3;
    if (orc_coupled_run(N, T, b, p, 2, F, scen, 0.1, 0.1, 0, T - 1, s[0], s[1], s[2], s[3], s[4], s[5], s[6], 0, N)) return 4;
    int32_t ov[3] = {0, 0, 1}, ot[3] = {5, 10, 20};
    double val[3] = {0.1, 0.2, 0.05}, sig[3] = {0.1, 0.1, 0.2}, ll[N];
    const double* ser[2] = {s[0], s[1]};
    if (orc_gaussian_loglik(N, T, ser, 3, ov, ot, val, sig, 1, ll, 0, N)) return 5;
    double ts = 0, td = 0, heat = 0, y[3] = {280, 0, 0};
    orc_two_layer_solve(p, 4.0, 2000, 2001, 0.1, &ts, &td, &heat);
    orc_carbon_cycle_solve((double[]){20.3, 280.0, 0.0}, 10.0, 1.0, 1850, 1851, 1.0 / 120.0, y);
    double agg[3] = {1.0, NAN, 3.0};
    if (orc_aggregate_sum(agg, 3) != 4.0) return 6;
    if (fabs(orc_co2_erf(3.7, 278.0, 556.0) - 3.7) > 1e-10) return 7;
    if (orc_rk4_nsteps(1750, 1751, 0.1) != 10 || !orc_rk4_endtime_ok(1750, 1751, 0.1)) return 8;
    {   /* ClimateUDEB: 3 members x 20 years */
        const int P = orc_udeb_n_params(), NU = 3, TU = 21;
        double* up = malloc(sizeof(double) * P * NU), ub[22], uf[21], init[4] = {0, 0, 0, 0};
        double* uo[7];
        int32_t ust[3];
        for (int i = 0; i < NU; ++i) {
            double d[64];
            orc_udeb_default_params(d);
            d[10] = 2.0 + i; /* ecs */
            for (int j = 0; j < P; ++j) up[j * NU + i] = d[j];
        }
        for (int i = 0; i <= TU; ++i) ub[i] = 1850.0 + i;
        for (int i = 0; i < TU; ++i) uf[i] = i ? 3.71 : 0.0;
        for (int k = 0; k < 7; ++k) uo[k] = malloc(sizeof(double) * TU * NU);
        if (orc_udeb_run(NU, TU, ub, up, 1, uf, NULL, init, uo[0], uo[1], uo[2], uo[3], uo[4], uo[5], uo[6], ust, 0, NU)) return 9;
        if (ust[0] || !(uo[0][(TU - 1) * NU + 2] > uo[0][(TU - 1) * NU + 0])) return 10;
        for (int k = 0; k < 7; ++k) free(uo[k]);
        free(up);
    }
    {
        const int rc = magicc_kinds();
        if (rc) return rc;
    }
    printf("oracle selftest ok: Ts=%.17g lnL[0]=%.17g conc=%.17g\n", ts, ll[0], y[0]);
    for (int k = 0; k < 7; ++k) free(s[k]);
    free(p);
    return 0;
}
