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
    printf("oracle selftest ok: Ts=%.17g lnL[0]=%.17g conc=%.17g\n", ts, ll[0], y[0]);
    for (int k = 0; k < 7; ++k) free(s[k]);
    free(p);
    return 0;
}
