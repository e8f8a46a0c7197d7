/*
 * rscm_oracle.c -- CPU restatement of the rscm hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared library.  The product (rscm_amd/, include/) never
 * links, imports or calls it; the product path fails loudly without its HIP
 * library instead of falling back to this code.
 *
 * What it restates (all f64, no FMA contraction: build with
 * -O2 -ffp-contract=off -fno-fast-math; evaluation order exactly as written):
 *
 *   time axis        crates/rscm-core/src/timeseries.rs:45-212
 *   stepper indices  crates/rscm-core/src/model/runtime.rs:368-527
 *                    crates/rscm-core/src/state/windows.rs:155-234
 *   RK4 driver       crates/rscm-core/src/ivp/mod.rs:73-102,245-253 and the
 *                    classical fixed-step scheme of ode_solvers 0.6.1 `Rk4`
 *                    (Cargo.lock:623-626; crate source NOT in the reference tree)
 *   TwoLayer         crates/rscm-two-layer/src/component.rs:159-251
 *   CarbonCycle      crates/rscm-components/src/components/carbon_cycle.rs:102-159
 *   CO2ERF           crates/rscm-components/src/components/co2_erf.rs:57-80
 *   Sum aggregate    crates/rscm-core/src/schema.rs:760-773,886-901
 *   Gaussian lnL     crates/rscm-calibrate/src/likelihood.rs:167-250
 *
 * PARITY PIN STATUS.  The reference is Rust and cannot be compiled or imported
 * in the build container (no cargo/rustc; `rscm._lib` is an unbuilt pyo3
 * cdylib), so there is no reference-produced numeric vector for TwoLayer.
 * Pinned by the reference's own known-answer tests: stepper index conventions,
 * time axis, CO2ERF exact points, CarbonCycle analytic solution (<1 %),
 * TwoLayer qualitative properties, likelihood values (tests/test_oracle_*.py).
 * UNPINNED ("parity unpinned"): the bit-level stage/sum association inside
 * ode_solvers::Rk4, restated here from the crate's published classical scheme
 *   k1=f(y) k2=f(y+k1*(h/2)) k3=f(y+k2*(h/2)) k4=f(y+k3*h)
 *   y' = y + (((k1 + k2*2) + k3*2) + k4) * (h/6),  n = ceil((t1-t0)/h)
 * and additionally pinned by build-supplied analytic tests (closed-form linear
 * solution, O(h^4) convergence, energy identity).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* crates/rscm-components/src/constants.rs:37 */
static const double GTC_PER_PPM = 2.13;
/* crates/rscm-core/src/ivp/mod.rs:73 */
static const double T_THRESHOLD = 5e-3;

/* ---- time axis: timeseries.rs:66-77 (from_values) -------------------------- */
ORC_API int orc_bounds_from_values(const double* v, int32_t T, double* bounds)
{
    if (T < 2) return 1;
    for (int32_t i = 1; i < T; ++i)
        if (!(v[i] > v[i - 1])) return 2; /* strictly increasing */
    for (int32_t i = 0; i < T; ++i) bounds[i] = v[i];
    bounds[T] = v[T - 1] + (v[T - 1] - v[T - 2]);
    return 0;
}

/* ---- RK4 step count and end-time acceptance: ode_solvers Rk4::integrate +
 *      ivp/mod.rs:90-102 (get_last_step asserts |t_last - t1| < 5e-3) --------- */
ORC_API int32_t orc_rk4_nsteps(double t0, double t1, double h)
{
    return (int32_t)ceil((t1 - t0) / h);
}

ORC_API int orc_rk4_endtime_ok(double t0, double t1, double h)
{
    int32_t m = orc_rk4_nsteps(t0, t1, h);
    double t = t0;
    if (m < 1) return 0; /* results would hold 1 entry: assert!(y.len() > 1) fails */
    for (int32_t i = 0; i < m; ++i) t = t + h;
    return fabs(t - t1) < T_THRESHOLD;
}

/* ---- generic 3-state classical RK4 (ode_solvers 0.6.1 Rk4::step) ------------ */
typedef void (*rhs3_fn)(const void* ctx, const double y[3], double dy[3]);

static void rk4_integrate3(rhs3_fn f, const void* ctx, double t0, double t1, double h, double y[3])
{
    const double half_step = h / 2.0;
    const double sixth = h / 6.0;
    const int32_t m = (int32_t)ceil((t1 - t0) / h);
    double k1[3], k2[3], k3[3], k4[3], yt[3];
    for (int32_t s = 0; s < m; ++s) {
        f(ctx, y, k1);
        for (int c = 0; c < 3; ++c) yt[c] = y[c] + k1[c] * half_step;
        f(ctx, yt, k2);
        for (int c = 0; c < 3; ++c) yt[c] = y[c] + k2[c] * half_step;
        f(ctx, yt, k3);
        for (int c = 0; c < 3; ++c) yt[c] = y[c] + k3[c] * h;
        f(ctx, yt, k4);
        for (int c = 0; c < 3; ++c)
            y[c] = y[c] + (((k1[c] + k2[c] * 2.0) + k3[c] * 2.0) + k4[c]) * sixth;
    }
}

/* ---- TwoLayer: rscm-two-layer/src/component.rs:159-189 ---------------------- */
typedef struct {
    double lambda0, a, efficacy, eta, cs, cd; /* component.rs:38-90 order */
    double erf;                               /* inputs.erf.get(): constant over the step */
} two_layer_ctx;

static void two_layer_rhs(const void* vctx, const double y[3], double dy[3])
{
    const two_layer_ctx* p = (const two_layer_ctx*)vctx;
    const double temperature_surface = y[0];
    const double temperature_deep = y[1];
    const double erf = p->erf;
    const double temperature_difference = temperature_surface - temperature_deep;
    const double lambda_eff = p->lambda0 - p->a * temperature_surface;
    const double heat_exchange_surface = p->efficacy * p->eta * temperature_difference;
    const double dts = (erf - lambda_eff * temperature_surface - heat_exchange_surface) / p->cs;
    const double heat_exchange_deep = p->eta * temperature_difference;
    const double dtd = heat_exchange_deep / p->cd;
    dy[0] = dts;
    dy[1] = dtd;
    dy[2] = p->cs * dts + p->cd * dtd; /* integrated from 0 and dropped (component.rs:236,245-248) */
}

/* One TwoLayer::solve (component.rs:223-251): y=(Ts,Td) in/out, h=0.1 in the reference. */
ORC_API void orc_two_layer_solve(const double params[6], double erf, double t0, double t1, double h,
                                 double* ts, double* td, double* heat)
{
    two_layer_ctx c = {params[0], params[1], params[2], params[3], params[4], params[5], erf};
    double y[3] = {*ts, *td, 0.0};
    rk4_integrate3(two_layer_rhs, &c, t0, t1, h, y);
    *ts = y[0];
    *td = y[1];
    if (heat) *heat = y[2];
}

/*
 * Ensemble run of the stand-alone two-layer model, steps [step_begin, step_end).
 *   params   [6][N]  SoA, rows in TwoLayerParameters field order
 *   forcing  [S][T]  ERF already on the model axis
 *   scen     [N] scenario of member, or NULL (all 0)
 *   source   0 = Exogenous/OwnState -> F[n]; 1 = UpstreamOutput -> F[n+1]
 *            (windows.rs:229-234; n+1 <= T-1 always holds inside run())
 *   ts, td   [T][N]  index step_begin holds the state; outputs go to n+1
 *            (runtime.rs:480).  member i starts at i0 and ends before i1 so
 *            threads can split the ensemble.
 */
ORC_API int orc_two_layer_run(int64_t N, int32_t T, const double* bounds, const double* params,
                              int32_t S, const double* forcing, const int32_t* scen, int source,
                              double h, int32_t step_begin, int32_t step_end, double* ts,
                              double* td, int64_t i0, int64_t i1)
{
    if (step_begin < 0 || step_end > T - 1 || step_begin > step_end) return 1;
    if (source != 0 && source != 1) return 2;
    for (int64_t i = i0; i < i1; ++i) {
        two_layer_ctx c = {params[0 * N + i], params[1 * N + i], params[2 * N + i],
                           params[3 * N + i], params[4 * N + i], params[5 * N + i], 0.0};
        const int32_t s = scen ? scen[i] : 0;
        if (s < 0 || s >= S) return 3;
        const double* F = forcing + (size_t)s * (size_t)T;
        for (int32_t n = step_begin; n < step_end; ++n) {
            double y[3] = {ts[(size_t)n * N + i], td[(size_t)n * N + i], 0.0};
            c.erf = F[n + source];
            rk4_integrate3(two_layer_rhs, &c, bounds[n], bounds[n + 1], h, y);
            ts[(size_t)(n + 1) * N + i] = y[0];
            td[(size_t)(n + 1) * N + i] = y[1];
        }
    }
    return 0;
}

/* ---- CarbonCycle: carbon_cycle.rs:133-159 ----------------------------------- */
typedef struct {
    double tau, conc_pi, alpha_temperature;
    double emissions, temperature; /* get(): constant over the step */
} carbon_ctx;

static void carbon_rhs(const void* vctx, const double y[3], double dy[3])
{
    const carbon_ctx* p = (const carbon_ctx*)vctx;
    const double conc = y[0];
    const double lifetime = p->tau * exp(p->alpha_temperature * p->temperature);
    const double uptake = (conc - p->conc_pi) / lifetime;
    dy[0] = p->emissions / GTC_PER_PPM - uptake;
    dy[1] = uptake * GTC_PER_PPM;
    dy[2] = p->emissions;
}

/* One CarbonCycle::solve (carbon_cycle.rs:102-131); y = (conc, cum_uptake, cum_emissions). */
ORC_API void orc_carbon_cycle_solve(const double params[3], double emissions, double temperature,
                                    double t0, double t1, double h, double y[3])
{
    carbon_ctx c = {params[0], params[1], params[2], emissions, temperature};
    rk4_integrate3(carbon_rhs, &c, t0, t1, h, y);
}

/* ---- CO2ERF: co2_erf.rs:57-60 ----------------------------------------------- */
ORC_API double orc_co2_erf(double erf_2xco2, double conc_pi, double concentration)
{
    return erf_2xco2 / log(2.0) * log(1.0 + (concentration - conc_pi) / conc_pi);
}

/* ---- scalar Sum aggregate: schema.rs:760-773 -------------------------------- */
ORC_API double orc_aggregate_sum(const double* values, int32_t k)
{
    double sum = 0.0;
    int32_t valid = 0;
    for (int32_t i = 0; i < k; ++i)
        if (!isnan(values[i])) {
            sum += values[i];
            ++valid;
        }
    return valid ? sum : NAN;
}

/*
 * Coupled chain, registration order CarbonCycle, CO2ERF, TwoLayer + schema
 * aggregate "Effective Radiative Forcing" = Sum(["Effective Radiative
 * Forcing|CO2"]) (docs/notebooks/coupled_model.py:435-483).  Per step n:
 *   CarbonCycle(E[n] exogenous, Ts[n] -- classified Exogenous because TwoLayer
 *     is registered later, builder.rs:470-482 -- own states C,U,S [n])
 *     -> C,U,S [n+1]
 *   CO2ERF(C[n+1] upstream)            -> ERF|CO2[n+1]
 *   Sum aggregate (at_end contributors) -> ERF[n+1]
 *   TwoLayer(ERF[n+1] upstream, Ts[n], Td[n]) -> Ts,Td [n+1]
 *
 *   params [10][N]: lambda0,a,efficacy,eta,cs,cd, tau,conc_pi_cc,alpha_temperature,
 *                   erf_2xco2   (CO2ERF conc_pi == params[7] as in the notebook)
 *   emissions [S][T]; series pointers are [T][N] each.
 */
ORC_API int orc_coupled_run(int64_t N, int32_t T, const double* bounds, const double* params,
                            int32_t S, const double* emissions, const int32_t* scen, double h_tl,
                            double h_cc, int32_t step_begin, int32_t step_end, double* ts,
                            double* td, double* conc, double* cum_uptake, double* cum_emis,
                            double* erf_co2, double* erf_total, int64_t i0, int64_t i1)
{
    if (step_begin < 0 || step_end > T - 1 || step_begin > step_end) return 1;
    for (int64_t i = i0; i < i1; ++i) {
        two_layer_ctx tl = {params[0 * N + i], params[1 * N + i], params[2 * N + i],
                            params[3 * N + i], params[4 * N + i], params[5 * N + i], 0.0};
        carbon_ctx cc = {params[6 * N + i], params[7 * N + i], params[8 * N + i], 0.0, 0.0};
        const double erf_2x = params[9 * N + i];
        const int32_t s = scen ? scen[i] : 0;
        if (s < 0 || s >= S) return 3;
        const double* E = emissions + (size_t)s * (size_t)T;
        for (int32_t n = step_begin; n < step_end; ++n) {
            const size_t a = (size_t)n * N + i, b = (size_t)(n + 1) * N + i;
            double yc[3] = {conc[a], cum_uptake[a], cum_emis[a]};
            cc.emissions = E[n];
            cc.temperature = ts[a];
            rk4_integrate3(carbon_rhs, &cc, bounds[n], bounds[n + 1], h_cc, yc);
            conc[b] = yc[0];
            cum_uptake[b] = yc[1];
            cum_emis[b] = yc[2];
            erf_co2[b] = orc_co2_erf(erf_2x, cc.conc_pi, conc[b]);
            erf_total[b] = orc_aggregate_sum(&erf_co2[b], 1);
            double yt[3] = {ts[a], td[a], 0.0};
            tl.erf = erf_total[b];
            rk4_integrate3(two_layer_rhs, &tl, bounds[n], bounds[n + 1], h_tl, yt);
            ts[b] = yt[0];
            td[b] = yt[1];
        }
    }
    return 0;
}

/*
 * Gaussian log-likelihood per member: likelihood.rs:186-250.
 *   per observation (:186-198): residual = obs - model;
 *       chi = (residual*residual)/(sigma*sigma); l = -0.5*chi;
 *       if normalize: l -= 0.5*ln(2*pi); l -= ln(sigma)
 *   per variable (:206-226): partial sum over its observations, in order
 *   total (:238-248): sum of the per-variable partials
 * Observations must be grouped by variable (obs_series non-interleaved); the
 * reference iterates a HashMap of variables, so the order of the groups is
 * arbitrary there and ours is the caller's.  Observation time matching is by
 * integer index (the reference matches "{:.6}" strings of the same axis values,
 * likelihood.rs:40-42).  A non-finite model value makes the member an error
 * (:216-221) -> log-posterior -inf (sampler/ensemble.rs:163-172).
 *   series [n_series] pointers to [T][N]; obs_series[j] picks one.
 */
ORC_API int orc_gaussian_loglik(int64_t N, int32_t T, const double* const* series,
                                int32_t n_obs, const int32_t* obs_series, const int32_t* obs_tidx,
                                const double* obs_value, const double* obs_sigma, int normalize,
                                double* out, int64_t i0, int64_t i1)
{
    const double ln_2pi = log(2.0 * M_PI);
    for (int32_t j = 0; j < n_obs; ++j)
        if (obs_tidx[j] < 0 || obs_tidx[j] >= T) return 1;
    for (int64_t i = i0; i < i1; ++i) {
        double total = 0.0, partial = 0.0;
        int bad = 0;
        for (int32_t j = 0; j < n_obs; ++j) {
            if (j > 0 && obs_series[j] != obs_series[j - 1]) {
                total += partial;
                partial = 0.0;
            }
            const double m = series[obs_series[j]][(size_t)obs_tidx[j] * N + i];
            if (!isfinite(m)) {
                bad = 1;
                break;
            }
            const double sigma = obs_sigma[j];
            const double residual = obs_value[j] - m;
            const double chi_squared = (residual * residual) / (sigma * sigma);
            double ln_l = -0.5 * chi_squared;
            if (normalize) {
                ln_l -= 0.5 * ln_2pi;
                ln_l -= log(sigma);
            }
            partial += ln_l;
        }
        total += partial;
        out[i] = bad ? -INFINITY : total;
    }
    return 0;
}
