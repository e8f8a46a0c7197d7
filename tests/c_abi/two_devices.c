/* The multi-device recipe of INTEGRATION.md section 4 as a compiled program: ONE process, one thread and one handle per
 * device -- the shape of a Rust host that replaces ModelRunner::run_batch (crates/rscm-calibrate/src/model_runner.rs:261-266)
 * with member blocks on the GPUs of a node; no torchrun, no torch.distributed, no Python.
 *
 *     two_devices <out.bin> <n_total> <n_threads>
 *
 * Thread k owns the contiguous members [offset_k, offset_k + count_k) of one global ensemble (blocks differ by at most one
 * member), on device k % n_devices: it creates its handle, draws ITS rows of the global Latin hypercube on the device
 * (rscm_ens_sample_lhs with member_offset / n_total: no scatter), runs the axis, scores its members on the device
 * (rscm_ens_loglik_device) and takes part in the gather of the per-member losses:
 *   - every thread on its own device (n_devices >= n_threads): ncclCommInitAll + ncclAllGather on the library's device
 *     pointers, RCCL over xGMI, 8 B per member;
 *   - threads sharing a device (a one-GPU box: RCCL refuses two ranks on one device): each copies its block into the global
 *     host vector (rscm_gpu_copy_to_host), and thread 0 additionally passes its block through a ONE-rank RCCL communicator
 *     -- the same ncclAllGather call on the same kind of pointer -- and checks that it comes back unchanged.
 * out.bin: loglik[n_total] as doubles, then status[n_total] as bytes.  tests/test_c_caller.py checks it against one handle
 * of n_total members, bit for bit. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "rscm_gpu.h"

#define T0 1750
#define T1 2500
#define NT (T1 - T0 + 1)
#define N_OBS 18
#define MAX_THREADS 16

static const double LOW[RSCM_TL_NPARAMS] = {0.8, 0.0, 1.0, 0.5, 5.0, 50.0};
static const double HIGH[RSCM_TL_NPARAMS] = {1.5, 0.1, 1.8, 1.0, 15.0, 200.0};

typedef struct {
    int rank, n_threads, n_devices, device, use_rccl;
    int64_t n_total, offset, count, max_count;
    const double* bounds;
    const double* forcing;
    ncclComm_t comm;          /* use_rccl: this thread's communicator of the node-wide group */
    double* host_loglik;      /* [n_total], written at this thread's offset (shared-device path) or by rank 0 (RCCL path) */
    uint8_t* host_status;     /* [n_total] */
    int one_rank_rccl_ok;     /* thread 0, shared-device path: its block survived a one-rank ncclAllGather */
    double run_ms;
    int failed;
    char error[512];
} Shard;

#define FAIL(s, ...) do { snprintf((s)->error, sizeof (s)->error, __VA_ARGS__); (s)->failed = 1; return NULL; } while (0)
#define RSCM(s, call) do { if ((call) != RSCM_OK) FAIL(s, "rank %d: %s: %s", (s)->rank, #call, rscm_gpu_last_error()); } while (0)
#define HIP(s, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) FAIL(s, "rank %d: %s: %s", (s)->rank, #call, hipGetErrorString(e_)); } while (0)
#define NCCL(s, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) FAIL(s, "rank %d: %s: %s", (s)->rank, #call, ncclGetErrorString(r_)); } while (0)

static void* shard_main(void* arg)
{
    Shard* s = arg;
    HIP(s, hipSetDevice(s->device));
    rscm_ens* ens = NULL;
    RSCM(s, rscm_ens_create(RSCM_KIND_TWO_LAYER, s->count, NT, s->bounds, s->device, &ens));
    RSCM(s, rscm_ens_sample_lhs(ens, 20260327u, LOW, HIGH, s->offset, s->n_total));
    RSCM(s, rscm_ens_set_forcing(ens, RSCM_TL_VAR_ERF, 1, s->forcing, NULL, RSCM_SRC_EXOGENOUS));
    const double zero = 0.0;
    RSCM(s, rscm_ens_set_initial(ens, RSCM_TL_VAR_TS, &zero, 1));
    RSCM(s, rscm_ens_set_initial(ens, RSCM_TL_VAR_TD, &zero, 1));
    RSCM(s, rscm_ens_run(ens, 0, NT - 1));
    float ms = 0.0f;
    RSCM(s, rscm_ens_last_run_ms(ens, &ms));
    s->run_ms = ms;
    int32_t obs_var[N_OBS], obs_tidx[N_OBS];
    double obs_value[N_OBS], obs_sigma[N_OBS];
    for (int k = 0; k < N_OBS; ++k) {   /* Surface Temperature 1850, 1860, ..., 2020, sigma 0.1 K */
        obs_var[k] = RSCM_TL_VAR_TS;
        obs_tidx[k] = 100 + 10 * k;
        obs_value[k] = 1.0 + 0.004 * (double)obs_tidx[k];
        obs_sigma[k] = 0.1;
    }
    void* d_loglik = NULL;
    RSCM(s, rscm_ens_loglik_device(ens, N_OBS, obs_var, obs_tidx, obs_value, obs_sigma, 0, &d_loglik));
    RSCM(s, rscm_ens_status(ens, s->host_status + s->offset));

    if (s->use_rccl) {   /* one rank per device: all-gather of equal (padded) blocks, device to device */
        hipStream_t stream;
        double *d_send = NULL, *d_recv = NULL;
        HIP(s, hipStreamCreate(&stream));
        HIP(s, hipMalloc((void**)&d_send, sizeof(double) * (size_t)s->max_count));
        HIP(s, hipMalloc((void**)&d_recv, sizeof(double) * (size_t)s->max_count * (size_t)s->n_threads));
        HIP(s, hipMemsetAsync(d_send, 0, sizeof(double) * (size_t)s->max_count, stream));
        HIP(s, hipMemcpyAsync(d_send, d_loglik, sizeof(double) * (size_t)s->count, hipMemcpyDeviceToDevice, stream));
        NCCL(s, ncclAllGather(d_send, d_recv, (size_t)s->max_count, ncclDouble, s->comm, stream));
        HIP(s, hipStreamSynchronize(stream));
        if (s->rank == 0) {   /* every rank holds the whole vector; rank 0 unpads it for the host */
            double* padded = malloc(sizeof(double) * (size_t)s->max_count * (size_t)s->n_threads);
            if (!padded) FAIL(s, "rank 0: host allocation");
            HIP(s, hipMemcpy(padded, d_recv, sizeof(double) * (size_t)s->max_count * (size_t)s->n_threads, hipMemcpyDeviceToHost));
            const int64_t base = s->n_total / s->n_threads, rem = s->n_total % s->n_threads;
            for (int r = 0; r < s->n_threads; ++r) {
                const int64_t cnt = base + (r < rem ? 1 : 0), off = r * base + (r < rem ? r : rem);
                memcpy(s->host_loglik + off, padded + (size_t)r * (size_t)s->max_count, sizeof(double) * (size_t)cnt);
            }
            free(padded);
        }
        HIP(s, hipFree(d_send));
        HIP(s, hipFree(d_recv));
        HIP(s, hipStreamDestroy(stream));
    } else {
        RSCM(s, rscm_gpu_copy_to_host(s->device, s->host_loglik + s->offset, d_loglik, (int64_t)sizeof(double) * s->count));
        if (s->rank == 0) {   /* the library's pointer through RCCL all the same: a communicator of one rank */
            ncclComm_t one;
            int dev = s->device;
            hipStream_t stream;
            double* d_recv = NULL;
            double* back = malloc(sizeof(double) * (size_t)s->count);
            if (!back) FAIL(s, "rank 0: host allocation");
            NCCL(s, ncclCommInitAll(&one, 1, &dev));
            HIP(s, hipStreamCreate(&stream));
            HIP(s, hipMalloc((void**)&d_recv, sizeof(double) * (size_t)s->count));
            NCCL(s, ncclAllGather(d_loglik, d_recv, (size_t)s->count, ncclDouble, one, stream));
            HIP(s, hipStreamSynchronize(stream));
            HIP(s, hipMemcpy(back, d_recv, sizeof(double) * (size_t)s->count, hipMemcpyDeviceToHost));
            s->one_rank_rccl_ok = memcmp(back, s->host_loglik + s->offset, sizeof(double) * (size_t)s->count) == 0;
            free(back);
            HIP(s, hipFree(d_recv));
            HIP(s, hipStreamDestroy(stream));
            NCCL(s, ncclCommDestroy(one));
        }
    }
    RSCM(s, rscm_ens_destroy(ens));
    return NULL;
}

int main(int argc, char** argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: two_devices <out.bin> <n_total> <n_threads>\n");
        return 64;
    }
    const int64_t n_total = (int64_t)atoll(argv[2]);
    const int n_threads = atoi(argv[3]);
    if (n_total < n_threads || n_threads < 1 || n_threads > MAX_THREADS) return 64;
    int32_t n_dev = 0;
    if (rscm_gpu_device_count(&n_dev) != RSCM_OK || n_dev < 1) {
        fprintf(stderr, "two_devices: no GPU: %s\n", rscm_gpu_last_error());
        return 2;
    }
    static double bounds[NT + 1], forcing[NT];
    for (int t = 0; t <= NT; ++t) bounds[t] = (double)(T0 + t);
    for (int t = 0; t < NT; ++t) {
        const double x = (double)t / 120.0;
        forcing[t] = 4.0 * x / (1.0 + x);
    }
    double* loglik = malloc(sizeof(double) * (size_t)n_total);
    uint8_t* status = malloc((size_t)n_total);
    if (!loglik || !status) return 5;
    const int use_rccl = n_dev >= n_threads && n_threads > 1;
    ncclComm_t comms[MAX_THREADS];
    if (use_rccl) {
        int devs[MAX_THREADS];
        for (int k = 0; k < n_threads; ++k) devs[k] = k;
        const ncclResult_t r = ncclCommInitAll(comms, n_threads, devs);
        if (r != ncclSuccess) {
            fprintf(stderr, "two_devices: ncclCommInitAll: %s\n", ncclGetErrorString(r));
            return 3;
        }
    }
    Shard shards[MAX_THREADS];
    pthread_t threads[MAX_THREADS];
    const int64_t base = n_total / n_threads, rem = n_total % n_threads;
    for (int k = 0; k < n_threads; ++k) {
        Shard* s = &shards[k];
        memset(s, 0, sizeof *s);
        s->rank = k; s->n_threads = n_threads; s->n_devices = n_dev; s->device = k % n_dev; s->use_rccl = use_rccl;
        s->n_total = n_total;
        s->count = base + (k < rem ? 1 : 0);
        s->offset = k * base + (k < rem ? k : rem);
        s->max_count = base + (rem ? 1 : 0);
        s->bounds = bounds; s->forcing = forcing;
        s->host_loglik = loglik; s->host_status = status;
        if (use_rccl) s->comm = comms[k];
        if (pthread_create(&threads[k], NULL, shard_main, s) != 0) return 6;
    }
    int failed = 0;
    for (int k = 0; k < n_threads; ++k) {
        pthread_join(threads[k], NULL);
        if (shards[k].failed) {
            fprintf(stderr, "two_devices: %s\n", shards[k].error);
            failed = 1;
        }
    }
    if (use_rccl)
        for (int k = 0; k < n_threads; ++k) ncclCommDestroy(comms[k]);
    if (failed) return 7;
    FILE* f = fopen(argv[1], "wb");
    if (!f) return 8;
    int ok = fwrite(loglik, sizeof(double), (size_t)n_total, f) == (size_t)n_total;
    ok = ok && fwrite(status, 1, (size_t)n_total, f) == (size_t)n_total;
    if (fclose(f) != 0 || !ok) return 9;
    printf("{\"members\": %lld, \"threads\": %d, \"devices\": %d, \"gather\": \"%s\", \"one_rank_rccl_ok\": %s, \"run_ms\": [",
           (long long)n_total, n_threads, n_dev, use_rccl ? "ncclAllGather" : "rscm_gpu_copy_to_host",
           use_rccl ? "null" : (shards[0].one_rank_rccl_ok ? "true" : "false"));
    for (int k = 0; k < n_threads; ++k) printf("%s%.3f", k ? ", " : "", shards[k].run_ms);
    printf("]}\n");
    free(loglik);
    free(status);
    return 0;
}
