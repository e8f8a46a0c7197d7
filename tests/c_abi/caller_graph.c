/* The component-graph side of the C-ABI from a compiled caller (strict C11, include/rscm_gpu.h only): what a host that keeps the
 * reference's Component trait and ModelBuilder (crates/rscm-core/src/model/builder.rs:464-518, runtime.rs:368-527) does with the
 * coupled chain of docs/notebooks/coupled_model.py -- CarbonCycle -> CO2ERF -> Sum aggregate -> TwoLayer, the temperature fed back
 * to the carbon cycle one step late --
 *   (1) as FOUR linked ensembles stepped in lock-step (rscm_ens_link_input with each consumer's VariableSource,
 *       rscm_ens_run_lockstep), on a stream of the caller's (rscm_gpu_stream_create / rscm_ens_set_stream);
 *   (2) as the fused RSCM_KIND_COUPLED kind;
 * checks inside C that (1) and (2) carry the same bits on every series, scores both against the same observations
 * (rscm_ens_loglik) and writes the fused kind's series for the test to hold against the oracle.
 *
 *     caller_graph <out.bin> [n_members]
 *
 * out.bin: int64 N, int64 T, then Ts[T][N], CO2[T][N], loglik[N] as doubles. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rscm_gpu.h"

#define T0 1750
#define T1 1950
#define NT (T1 - T0 + 1)
#define N_OBS 6

static int check(int rc, const char* what)
{
    if (rc != RSCM_OK) {
        fprintf(stderr, "caller_graph: %s failed (%d): %s\n", what, rc, rscm_gpu_last_error());
        exit(2);
    }
    return rc;
}

static double* rows(size_t n) { double* p = malloc(sizeof(double) * n); if (!p) exit(5); return p; }

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: caller_graph <out.bin> [n_members]\n");
        return 64;
    }
    const int64_t n = argc > 2 ? (int64_t)atoll(argv[2]) : 3000;
    int32_t n_dev = 0;
    check(rscm_gpu_device_count(&n_dev), "rscm_gpu_device_count");
    if (n_dev < 1) return 4;

    double bounds[NT + 1], emissions[NT], nan_row[NT];
    for (int t = 0; t <= NT; ++t) bounds[t] = (double)(T0 + t);
    for (int t = 0; t < NT; ++t) {
        emissions[t] = 0.02 * (double)t;             /* GtC / yr, linear ramp: no libm on either side of the test */
        nan_row[t] = 0.0 / 0.0;
    }
    /* member parameters (SoA rows): the six two-layer rows, tau, conc_pi, alpha_temperature, erf_2xco2 */
    static const double lo[RSCM_CP_NPARAMS] = {0.9, 0.0, 1.0, 0.5, 5.0, 50.0, 15.0, 278.0, 0.0, 3.7};
    static const double hi[RSCM_CP_NPARAMS] = {1.5, 0.05, 1.8, 1.0, 15.0, 200.0, 40.0, 278.0, 0.1, 3.7};
    double* P = rows((size_t)RSCM_CP_NPARAMS * (size_t)n);
    for (int j = 0; j < RSCM_CP_NPARAMS; ++j)
        for (int64_t i = 0; i < n; ++i) {
            const uint32_t hash = (uint32_t)((uint64_t)i * 2246822519u + (uint64_t)j * 374761393u + 7u);
            P[(size_t)j * (size_t)n + (size_t)i] = lo[j] + (hi[j] - lo[j]) * ((double)hash / 4294967296.0);
        }

    /* ---- (2) the fused kind */
    rscm_ens* fused = NULL;
    check(rscm_ens_create(RSCM_KIND_COUPLED, n, NT, bounds, 0, &fused), "rscm_ens_create(coupled)");
    check(rscm_ens_set_params(fused, P), "rscm_ens_set_params(coupled)");
    check(rscm_ens_set_forcing(fused, RSCM_CP_VAR_EMISSIONS, 1, emissions, NULL, RSCM_SRC_EXOGENOUS), "rscm_ens_set_forcing(coupled)");
    const double zero = 0.0, c0 = 278.0;
    check(rscm_ens_set_initial(fused, RSCM_CP_VAR_TS, &zero, 1), "initial Ts");
    check(rscm_ens_set_initial(fused, RSCM_CP_VAR_TD, &zero, 1), "initial Td");
    check(rscm_ens_set_initial(fused, RSCM_CP_VAR_CONC, &c0, 1), "initial CO2");
    check(rscm_ens_set_initial(fused, RSCM_CP_VAR_CUM_UPTAKE, &zero, 1), "initial uptake");
    check(rscm_ens_set_initial(fused, RSCM_CP_VAR_CUM_EMIS, &zero, 1), "initial emissions");
    check(rscm_ens_run(fused, 0, NT - 1), "rscm_ens_run(coupled)");

    /* ---- (1) four linked ensembles on one stream of the caller's */
    void* stream = NULL;
    check(rscm_gpu_stream_create(0, &stream), "rscm_gpu_stream_create");
    rscm_ens *cc = NULL, *ce = NULL, *ag = NULL, *tl = NULL;
    check(rscm_ens_create(RSCM_KIND_CARBON_CYCLE, n, NT, bounds, 0, &cc), "create CarbonCycle");
    check(rscm_ens_create(RSCM_KIND_CO2_ERF, n, NT, bounds, 0, &ce), "create CO2ERF");
    check(rscm_ens_create(RSCM_KIND_AGGREGATE, n, NT, bounds, 0, &ag), "create aggregate");
    check(rscm_ens_create(RSCM_KIND_TWO_LAYER, n, NT, bounds, 0, &tl), "create TwoLayer");
    rscm_ens* order[4];
    order[0] = cc; order[1] = ce; order[2] = ag; order[3] = tl;
    for (int k = 0; k < 4; ++k) check(rscm_ens_set_stream(order[k], stream), "rscm_ens_set_stream");
    double* Pcc = rows((size_t)RSCM_CC_NPARAMS * (size_t)n);
    double* Pce = rows((size_t)RSCM_CE_NPARAMS * (size_t)n);
    double* Pag = rows((size_t)RSCM_AG_NPARAMS * (size_t)n);
    memcpy(Pcc, P + (size_t)6 * (size_t)n, sizeof(double) * 3 * (size_t)n);                 /* tau, conc_pi, alpha_temperature */
    memcpy(Pce, P + (size_t)9 * (size_t)n, sizeof(double) * (size_t)n);                     /* erf_2xco2 */
    memcpy(Pce + (size_t)n, P + (size_t)7 * (size_t)n, sizeof(double) * (size_t)n);         /* conc_pi */
    memset(Pag, 0, sizeof(double) * (size_t)RSCM_AG_NPARAMS * (size_t)n);                   /* operation 0: Sum */
    check(rscm_ens_set_params(cc, Pcc), "params CarbonCycle");
    check(rscm_ens_set_params(ce, Pce), "params CO2ERF");
    check(rscm_ens_set_params(ag, Pag), "params aggregate");
    check(rscm_ens_set_params(tl, P), "params TwoLayer");                                   /* rows 0..5 */
    double cc_inputs[2 * NT];                                                               /* [n_inputs][T]: emissions; the temperature row is linked */
    memcpy(cc_inputs, emissions, sizeof emissions);
    memcpy(cc_inputs + NT, nan_row, sizeof nan_row);
    check(rscm_ens_set_forcing(cc, 0, 1, cc_inputs, NULL, RSCM_SRC_EXOGENOUS), "inputs CarbonCycle");
    check(rscm_ens_set_initial(cc, 1, &c0, 1), "initial CO2 (linked)");
    check(rscm_ens_set_initial(cc, 2, &zero, 1), "initial uptake (linked)");
    check(rscm_ens_set_initial(cc, 3, &zero, 1), "initial emissions (linked)");
    check(rscm_ens_set_initial(tl, RSCM_TL_VAR_TS, &zero, 1), "initial Ts (linked)");
    check(rscm_ens_set_initial(tl, RSCM_TL_VAR_TD, &zero, 1), "initial Td (linked)");
    /* the edges, each with its consumer's VariableSource (builder.rs:470-482): TwoLayer is registered after CarbonCycle, so the
     * carbon cycle reads the temperature at index n (a lagged feedback); everything downstream reads its producer at n + 1 */
    check(rscm_ens_link_input(cc, 1, tl, RSCM_TL_VAR_TS, RSCM_SRC_EXOGENOUS), "link CarbonCycle <- Ts");
    check(rscm_ens_link_input(ce, 0, cc, 1, RSCM_SRC_UPSTREAM), "link CO2ERF <- CO2");
    check(rscm_ens_link_input(ag, 0, ce, 1, RSCM_SRC_UPSTREAM), "link Sum <- ERF|CO2");
    check(rscm_ens_link_input(tl, 0, ag, 1, RSCM_SRC_UPSTREAM), "link TwoLayer <- ERF");
    check(rscm_ens_run_lockstep((rscm_ens* const*)order, 4, 0, NT - 1), "rscm_ens_run_lockstep");
    check(rscm_ens_sync(tl), "rscm_ens_sync");

    /* ---- the same bits */
    double* a = rows((size_t)NT * (size_t)n);
    double* b = rows((size_t)NT * (size_t)n);
    double* co2 = rows((size_t)NT * (size_t)n);
    struct { rscm_ens* h; int32_t var; int32_t fused_var; const char* name; } pairs[4] = {
        {tl, RSCM_TL_VAR_TS, RSCM_CP_VAR_TS, "Surface Temperature"}, {tl, RSCM_TL_VAR_TD, RSCM_CP_VAR_TD, "Deep Ocean Temperature"},
        {cc, 1, RSCM_CP_VAR_CONC, "Atmospheric Concentration|CO2"}, {ag, 1, RSCM_CP_VAR_ERF, "Effective Radiative Forcing"}};
    int same = 1;
    for (int k = 0; k < 4; ++k) {
        check(rscm_ens_get_series(pairs[k].h, pairs[k].var, 0, NT, 1, 0, n, a), "rscm_ens_get_series(linked)");
        check(rscm_ens_get_series(fused, pairs[k].fused_var, 0, NT, 1, 0, n, b), "rscm_ens_get_series(fused)");
        if (memcmp(a, b, sizeof(double) * (size_t)NT * (size_t)n) != 0) {
            fprintf(stderr, "caller_graph: the linked graph and the fused kind differ in %s\n", pairs[k].name);
            same = 0;
        }
        if (k == 2) memcpy(co2, b, sizeof(double) * (size_t)NT * (size_t)n);
    }
    check(rscm_ens_get_series(fused, RSCM_CP_VAR_TS, 0, NT, 1, 0, n, a), "Ts");

    /* ---- one likelihood, two evaluators: observations of Ts every 30 years, sigma 0.2 K */
    int32_t obs_var_f[N_OBS], obs_var_l[N_OBS], obs_tidx[N_OBS];
    double obs_value[N_OBS], obs_sigma[N_OBS];
    for (int k = 0; k < N_OBS; ++k) {
        obs_var_f[k] = RSCM_CP_VAR_TS;
        obs_var_l[k] = RSCM_TL_VAR_TS;
        obs_tidx[k] = 30 * (k + 1);
        obs_value[k] = 0.004 * (double)obs_tidx[k];
        obs_sigma[k] = 0.2;
    }
    double* ll_f = rows((size_t)n);
    double* ll_l = rows((size_t)n);
    check(rscm_ens_loglik(fused, N_OBS, obs_var_f, obs_tidx, obs_value, obs_sigma, 0, ll_f), "rscm_ens_loglik(fused)");
    check(rscm_ens_loglik(tl, N_OBS, obs_var_l, obs_tidx, obs_value, obs_sigma, 0, ll_l), "rscm_ens_loglik(linked)");
    if (memcmp(ll_f, ll_l, sizeof(double) * (size_t)n) != 0) {
        fprintf(stderr, "caller_graph: the two evaluators score differently\n");
        same = 0;
    }

    /* a source must outlive its links (rscm_ens_destroy refuses), then everything goes in consumer-first order */
    const int refused = rscm_ens_destroy(ag) != RSCM_OK;
    check(rscm_ens_unlink_input(cc, 1), "rscm_ens_unlink_input");
    check(rscm_ens_destroy(tl), "destroy TwoLayer");
    check(rscm_ens_destroy(ag), "destroy aggregate");
    check(rscm_ens_destroy(ce), "destroy CO2ERF");
    check(rscm_ens_destroy(cc), "destroy CarbonCycle");
    check(rscm_ens_destroy(fused), "destroy coupled");
    check(rscm_gpu_stream_destroy(0, stream), "rscm_gpu_stream_destroy");

    FILE* f = fopen(argv[1], "wb");
    if (!f) return 8;
    const int64_t head[2] = {n, NT};
    int ok = fwrite(head, sizeof head, 1, f) == 1;
    ok = ok && fwrite(a, sizeof(double), (size_t)NT * (size_t)n, f) == (size_t)NT * (size_t)n;
    ok = ok && fwrite(co2, sizeof(double), (size_t)NT * (size_t)n, f) == (size_t)NT * (size_t)n;
    ok = ok && fwrite(ll_f, sizeof(double), (size_t)n, f) == (size_t)n;
    if (fclose(f) != 0 || !ok) return 9;
    printf("{\"members\": %lld, \"steps\": %d, \"linked_equals_fused\": %s, \"destroy_of_a_linked_source_refused\": %s}\n", (long long)n, NT - 1,
           same ? "true" : "false", refused ? "true" : "false");
    return same && refused ? 0 : 10;
}
