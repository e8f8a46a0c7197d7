/* A compiled caller of the C-ABI: strict C11, nothing included but <stdio.h>/<stdlib.h>/<stdint.h> and include/rscm_gpu.h,
 * linked against rscm_amd/librscm_gpu.so -- what a Rust `extern "C"` block, a cgo stub or any other non-Python host binds
 * (INTEGRATION.md section 1).  It does what ModelRunner::run_batch does for the two-layer model
 * (crates/rscm-calibrate/src/model_runner.rs:161-266): parameter rows [N][P] in, every member stepped over the whole axis,
 * per-member series and status out.
 *
 *     caller <out.bin> [n_members]
 *
 * Inputs are formed with +, * and / only, from integers (no libm), so that the test re-forms them bit for bit in numpy and
 * holds the output to the oracle's: members' parameters from a multiplicative hash of (member, parameter), forcing
 * F[t] = 4 x / (1 + x), x = (t - 1750) / 120.  out.bin: int64 N, int64 K (kept rows), int32 rows[K], then Ts[K][N], Td[K][N]
 * as doubles, then status[N] as bytes.  tests/test_gpu_c_caller.py builds, runs and checks it. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rscm_gpu.h"

#define T0 1750
#define T1 2500
#define NT (T1 - T0 + 1)

static const double LOW[RSCM_TL_NPARAMS] = {0.8, 0.0, 1.0, 0.5, 5.0, 50.0};
static const double HIGH[RSCM_TL_NPARAMS] = {1.5, 0.1, 1.8, 1.0, 15.0, 200.0};

static int check(int rc, const char* what)
{
    if (rc != RSCM_OK) {
        fprintf(stderr, "caller: %s failed (%d): %s\n", what, rc, rscm_gpu_last_error());
        exit(2);
    }
    return rc;
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: caller <out.bin> [n_members]\n");
        return 64;
    }
    const int64_t n = argc > 2 ? (int64_t)atoll(argv[2]) : 70000;
    if (rscm_gpu_abi_version() != RSCM_GPU_ABI_VERSION || rscm_gpu_abi_minor() < RSCM_GPU_ABI_MINOR) {
        fprintf(stderr, "caller: library ABI %d.%d, header %d.%d\n", rscm_gpu_abi_version(), rscm_gpu_abi_minor(),
                RSCM_GPU_ABI_VERSION, RSCM_GPU_ABI_MINOR);
        return 3;
    }
    int32_t n_dev = 0;
    check(rscm_gpu_device_count(&n_dev), "rscm_gpu_device_count");
    if (n_dev < 1) {
        fprintf(stderr, "caller: no GPU\n");
        return 4;
    }

    double* bounds = malloc(sizeof(double) * (NT + 1));
    double* forcing = malloc(sizeof(double) * NT);
    double* params = malloc(sizeof(double) * (size_t)n * RSCM_TL_NPARAMS);   /* [N][P], the run_batch shape */
    if (!bounds || !forcing || !params) return 5;
    for (int t = 0; t <= NT; ++t) bounds[t] = (double)(T0 + t);
    for (int t = 0; t < NT; ++t) {
        const double x = (double)t / 120.0;
        forcing[t] = 4.0 * x / (1.0 + x);
    }
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < RSCM_TL_NPARAMS; ++j) {
            const uint32_t hash = (uint32_t)((uint64_t)i * 2654435761u + (uint64_t)j * 40503u + 12345u);
            const double u = (double)hash / 4294967296.0;
            params[(size_t)i * RSCM_TL_NPARAMS + j] = LOW[j] + (HIGH[j] - LOW[j]) * u;
        }

    rscm_ens* ens = NULL;
    check(rscm_ens_create(RSCM_KIND_TWO_LAYER, n, NT, bounds, 0, &ens), "rscm_ens_create");
    int32_t n_params = 0, n_times = 0;
    int64_t n_members = 0;
    check(rscm_ens_n_params(ens, &n_params), "rscm_ens_n_params");
    check(rscm_ens_n_times(ens, &n_times), "rscm_ens_n_times");
    check(rscm_ens_n_members(ens, &n_members), "rscm_ens_n_members");
    if (n_params != RSCM_TL_NPARAMS || n_times != NT || n_members != n) return 6;
    check(rscm_ens_set_params_aos(ens, params), "rscm_ens_set_params_aos");
    check(rscm_ens_set_forcing(ens, RSCM_TL_VAR_ERF, 1, forcing, NULL, RSCM_SRC_EXOGENOUS), "rscm_ens_set_forcing");
    const double zero = 0.0;
    check(rscm_ens_set_initial(ens, RSCM_TL_VAR_TS, &zero, 1), "rscm_ens_set_initial(Ts)");
    check(rscm_ens_set_initial(ens, RSCM_TL_VAR_TD, &zero, 1), "rscm_ens_set_initial(Td)");
    check(rscm_ens_run(ens, 0, NT - 1), "rscm_ens_run");

    int32_t tidx = -1, blocks = 0, chunks = 0;
    float ms = 0.0f;
    check(rscm_ens_time_index(ens, &tidx), "rscm_ens_time_index");
    check(rscm_ens_last_run_plan(ens, &blocks, &chunks), "rscm_ens_last_run_plan");
    check(rscm_ens_last_run_ms(ens, &ms), "rscm_ens_last_run_ms");

    const int32_t stride = 50;
    const int64_t kept = (NT + stride - 1) / stride;
    double* ts = malloc(sizeof(double) * (size_t)kept * (size_t)n);
    double* td = malloc(sizeof(double) * (size_t)kept * (size_t)n);
    uint8_t* status = malloc((size_t)n);
    if (!ts || !td || !status) return 5;
    check(rscm_ens_get_series(ens, RSCM_TL_VAR_TS, 0, NT, stride, 0, n, ts), "rscm_ens_get_series(Ts)");
    check(rscm_ens_get_series(ens, RSCM_TL_VAR_TD, 0, NT, stride, 0, n, td), "rscm_ens_get_series(Td)");
    check(rscm_ens_status(ens, status), "rscm_ens_status");
    double summary[4];
    check(rscm_ens_summary(ens, RSCM_TL_VAR_TS, 270, summary), "rscm_ens_summary");

    /* an invalid call comes back as a code and a message, not as a crash */
    if (rscm_ens_run(ens, 0, NT - 1) != RSCM_ERR_STATE) {
        fprintf(stderr, "caller: a second whole run without a rewind was not refused\n");
        return 7;
    }
    check(rscm_ens_destroy(ens), "rscm_ens_destroy");

    FILE* f = fopen(argv[1], "wb");
    if (!f) return 8;
    int64_t head[2] = {n, kept};
    int ok = fwrite(head, sizeof head, 1, f) == 1;
    for (int64_t k = 0; k < kept && ok; ++k) {
        const int32_t row = (int32_t)(k * stride);
        ok = fwrite(&row, sizeof row, 1, f) == 1;
    }
    ok = ok && fwrite(ts, sizeof(double), (size_t)kept * (size_t)n, f) == (size_t)kept * (size_t)n;
    ok = ok && fwrite(td, sizeof(double), (size_t)kept * (size_t)n, f) == (size_t)kept * (size_t)n;
    ok = ok && fwrite(status, 1, (size_t)n, f) == (size_t)n;
    if (fclose(f) != 0 || !ok) return 9;

    int64_t failed = 0;
    for (int64_t i = 0; i < n; ++i) failed += status[i] != 0;
    printf("{\"members\": %lld, \"time_index\": %d, \"member_blocks\": %d, \"step_chunks\": %d, \"run_ms\": %.3f, "
           "\"failed_members\": %lld, \"ts_2020_count\": %.0f, \"ts_2020_mean\": %.17g}\n",
           (long long)n, tidx, blocks, chunks, (double)ms, (long long)failed, summary[0], summary[0] > 0.0 ? summary[1] / summary[0] : 0.0);   /* count, sum, min, max */
    free(ts); free(td); free(status); free(params); free(forcing); free(bounds);
    return 0;
}
