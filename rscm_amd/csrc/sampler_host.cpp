// The device stretch-move sampler behind rscm_sampler_* (kernels: csrc/sampler.hip): what the reference does on the host per
// iteration in EnsembleSampler::run / update_group (crates/rscm-calibrate/src/sampler/ensemble.rs:496-547) driven as launches on
// the evaluator's stream -- one evaluating ensemble (the fused two-layer run+likelihood, or any kind scored from its stored
// series) or a graph of linked ensembles (rscm_sampler_create_graph).
#include "ens.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------
// Device stretch-move sampler (csrc/sampler.hip)
// ---------------------------------------------------------------------------------------------
struct rscm_sampler {
    rscm_ens* ev = nullptr;  // evaluates one half-ensemble per launch; not owned
    int32_t W = 0, D = 0, groups = 1;
    bool sharded = false;            // driven half-step by half-step with an exchange in between
    int32_t rank = 0, n_ranks = 1;   // sharded: this rank owns half-walkers [rank * n_local, (rank + 1) * n_local) of both halves
    int32_t n_local = 0;             // = members of the evaluator
    double* d_send = nullptr;        // [D + 1][n_local]
    double* d_recv = nullptr;        // [n_ranks][D + 1][n_local]
    double stretch_a = 2.0;
    uint64_t seed = 0;
    int32_t iteration = 0;
    bool positions_set = false;
    int32_t* d_rows = nullptr;
    int32_t* d_kind = nullptr;
    double* d_base = nullptr;
    double* d_pa = nullptr;
    double* d_pb = nullptr;
    double* d_plo = nullptr;
    double* d_phi = nullptr;
    double* d_pos = nullptr;
    double* d_logp = nullptr;
    double* d_prop = nullptr;
    double* d_z = nullptr;
    double* d_lp = nullptr;
    int64_t* d_nacc = nullptr;
    int64_t* d_nprop = nullptr;
    // fused: the two-layer run+likelihood kernel scores a half; otherwise the half is run through
    // rscm_ens_run_async (any kind, stored series) and scored by the likelihood kernel
    bool fused = true;
    void* d_sobs = nullptr;          // stored path: observation rows, values, sigmas, groups
    rscm::LoglikArgs lik{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // graph evaluator (rscm_sampler_create_graph): every half-step rewinds these handles, runs them in lock-step up
    // to the last observed index and scores the stored series; ev is the first of them
    std::vector<rscm_ens*> graph;
    std::vector<uint64_t> graph_sampled_rows;  // per handle: bit j = parameter row j is written by the proposal kernel
    bool graph_clear = false;                  // NaN the stored rows before every run (graphs in which a consumer runs ahead of its producer)
    int32_t graph_last_step = 0;
    double** d_param_ptr = nullptr;            // [D] device addresses of the sampled rows
};

namespace {

rscm::SamplerArgs sampler_args(const rscm_sampler* s, int32_t half, int32_t identity)
{
    rscm::SamplerArgs a{};
    a.n_walkers = s->W;
    a.n_dims = s->D;
    a.n_params = s->ev->P;
    a.n_groups = s->groups;
    a.half = half;
    a.k_offset = s->rank * s->n_local;
    a.n_local = s->n_local;
    a.n_ranks = s->n_ranks;
    a.exchange = nullptr;
    a.iteration = s->iteration;
    a.identity = identity;
    a.seed = s->seed;
    a.stretch_a = s->stretch_a;
    a.param_rows = s->d_rows;
    a.base_params = s->d_base;
    a.prior_kind = s->d_kind;
    a.prior_a = s->d_pa;
    a.prior_b = s->d_pb;
    a.prior_lo = s->d_plo;
    a.prior_hi = s->d_phi;
    a.pos = s->d_pos;
    a.logp = s->d_logp;
    a.proposal = s->d_prop;
    a.z = s->d_z;
    a.lp = s->d_lp;
    a.loglik = s->ev->d_loglik;
    a.eval_params = s->ev->d_params;
    a.param_ptr = s->d_param_ptr;
    a.n_accepted = s->d_nacc;
    a.n_proposed = s->d_nprop;
    return a;
}

// propose (or re-score) one half, evaluate it, accept: all on the evaluator's stream
int sampler_half_step(rscm_sampler* s, int32_t half, int32_t identity)
{
    const rscm::SamplerArgs a = sampler_args(s, half, identity);
    if (!s->graph.empty()) {
        // Model::run of the whole graph for this half's proposals: a fresh collection, the proposal kernel writes each
        // sampled parameter into its owner's block, the graph steps to the last observed index (later steps cannot
        // change ln L), the likelihood kernel reads the observation rows where the owners store them
        for (size_t k = 0; k < s->graph.size(); ++k) {
            rscm_ens* g = s->graph[k];
            g->uniform_rows &= ~s->graph_sampled_rows[k];
            if (s->graph_sampled_rows[k]) g->derived_dirty = true;   // the proposal kernel is about to write its parameter block
            if (int rc = s->graph_clear ? rscm_ens_clear_series(g) : rscm_ens_rewind(g)) return rc;
        }
        HIPCHK(rscm::launch_sampler_propose(a, s->ev->stream));
        if (s->graph_last_step > 0)
            if (int rc = rscm_ens_run_lockstep(s->graph.data(), (int32_t)s->graph.size(), 0, s->graph_last_step)) return rc;
        HIPCHK(rscm::launch_loglik(s->lik, s->ev->stream));
        HIPCHK(rscm::launch_sampler_accept(a, s->ev->stream));
        if (s->sharded) {
            rscm::SamplerArgs p = a;
            p.exchange = s->d_send;
            HIPCHK(rscm::launch_sampler_pack(p, s->ev->stream));
        }
        return RSCM_OK;
    }
    s->ev->uniform_rows = 0;  // the proposal kernel writes the evaluator's parameter block
    s->ev->derived_dirty = true;
    HIPCHK(rscm::launch_sampler_propose(a, s->ev->stream));
    if (s->fused) {
        HIPCHK(launch_loglik(s->ev));
    } else {
        s->ev->time_index = 0;  // every evaluation is a fresh Model::run of the half
        if (int rc = rscm_ens_run_async(s->ev, 0, s->ev->T - 1)) return rc;
        s->ev->time_index = 0;
        HIPCHK(rscm::launch_loglik(s->lik, s->ev->stream));
    }
    HIPCHK(rscm::launch_sampler_accept(a, s->ev->stream));
    if (s->sharded) {  // this rank's block of the updated half, ready for the all-gather
        rscm::SamplerArgs p = a;
        p.exchange = s->d_send;
        HIPCHK(rscm::launch_sampler_pack(p, s->ev->stream));
    }
    return RSCM_OK;
}

}  // namespace

int rscm_sampler_create(rscm_ens* evaluator, int32_t n_walkers, int32_t n_dims, const int32_t* param_rows,
                        const double* base_params, const int32_t* prior_kind, const double* prior_a,
                        const double* prior_b, const double* prior_low, const double* prior_high,
                        int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                        const double* obs_value, const double* obs_sigma, int32_t normalize, double stretch_a,
                        uint64_t seed, rscm_sampler** out)
{
    return rscm_sampler_create_sharded(evaluator, n_walkers, n_dims, param_rows, base_params, prior_kind, prior_a, prior_b,
                                       prior_low, prior_high, n_obs, obs_var, obs_tidx, obs_value, obs_sigma, normalize,
                                       stretch_a, seed, 0, 1, out);
}

int rscm_sampler_create_sharded(rscm_ens* evaluator, int32_t n_walkers, int32_t n_dims, const int32_t* param_rows,
                        const double* base_params, const int32_t* prior_kind, const double* prior_a,
                        const double* prior_b, const double* prior_low, const double* prior_high,
                        int32_t n_obs, const int32_t* obs_var, const int32_t* obs_tidx,
                        const double* obs_value, const double* obs_sigma, int32_t normalize, double stretch_a,
                        uint64_t seed, int32_t rank, int32_t n_ranks, rscm_sampler** out)
{
    GUARD_BEGIN
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    rscm_ens* h = evaluator;
    NEED(h);
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RSCM_ERR_INVALID, "bad rank %d of %d", rank, n_ranks);
    if (h->windowed) return fail(RSCM_ERR_INVALID, "the sampler's evaluator must not be a windowed ensemble");
    if (n_walkers < 2) return fail(RSCM_ERR_INVALID, "Must have at least 2 walkers");          // ensemble.rs:120-127
    if (n_walkers % 2) return fail(RSCM_ERR_INVALID, "Number of walkers must be even");
    if ((n_walkers / 2) % n_ranks)
        return fail(RSCM_ERR_INVALID, "half the walkers (%d) must split evenly over %d ranks", n_walkers / 2, n_ranks);
    if (h->N != n_walkers / 2 / n_ranks)
        return fail(RSCM_ERR_INVALID, "the evaluating ensemble must have n_walkers / 2 / n_ranks = %d members, it has %lld",
                    n_walkers / 2 / n_ranks, (long long)h->N);
    if (!(stretch_a > 1.0)) return fail(RSCM_ERR_INVALID, "Stretch move scale parameter must be > 1.0, got %g", stretch_a);  // moves.rs:40-48
    if (n_dims < 1 || n_dims > h->P || !param_rows || !base_params || !prior_kind || !prior_a || !prior_b)
        return fail(RSCM_ERR_INVALID, "bad parameter description");
    for (int32_t d = 0; d < n_dims; ++d) {
        if (param_rows[d] < 0 || param_rows[d] >= h->P) return fail(RSCM_ERR_INVALID, "dimension %d: parameter row %d out of range", d, param_rows[d]);
        if (const char* name = structural_row(h->kind, param_rows[d]))
            return fail(RSCM_ERR_INVALID, "dimension %d: parameter row %d (%s) is structural -- the host derives tables from it when the "
                        "parameters are set -- and cannot be sampled on the device", d, param_rows[d], name);
        for (int32_t e2 = 0; e2 < d; ++e2)
            if (param_rows[e2] == param_rows[d]) return fail(RSCM_ERR_INVALID, "parameter row %d sampled twice", param_rows[d]);
        if (prior_kind[d] < 0 || prior_kind[d] > 2) return fail(RSCM_ERR_INVALID, "dimension %d: unknown prior kind %d", d, prior_kind[d]);
        if (prior_kind[d] == 0 && !(prior_b[d] > prior_a[d])) return fail(RSCM_ERR_INVALID, "dimension %d: Uniform needs high > low", d);
        if (prior_kind[d] != 0 && !(prior_b[d] > 0.0)) return fail(RSCM_ERR_INVALID, "dimension %d: the scale parameter must be > 0", d);
        if ((prior_low && prior_high) && !(prior_low[d] < prior_high[d])) return fail(RSCM_ERR_INVALID, "dimension %d: Bound needs low < high", d);
    }
    // the fused kernel takes two-layer observations with ascending time indices inside a group
    bool fused = h->kind == RSCM_KIND_TWO_LAYER;
    for (int32_t j = 0; j < n_obs && fused; ++j) {
        if (!obs_var || !obs_tidx) return fail(RSCM_ERR_INVALID, "bad observation arrays");
        if (j > 0 && obs_var[j] == obs_var[j - 1] && obs_tidx[j] < obs_tidx[j - 1]) fused = false;
    }
    if (fused) {
        if (int rc = prepare_obs(h, n_obs, obs_var, obs_tidx, obs_value, obs_sigma, normalize)) return rc;
        h->loglik_stop_at_last_obs = true;   // cleared again by rscm_sampler_destroy
    } else {
        if (h->rows != h->T)
            return fail(RSCM_ERR_INVALID, "this evaluator stores no series: only the fused two-layer likelihood "
                                          "(ascending observation times) is available for it");
        if (n_obs < 0 || (n_obs > 0 && (!obs_var || !obs_tidx || !obs_value || !obs_sigma)))
            return fail(RSCM_ERR_INVALID, "bad observation arrays");
        for (int32_t j = 0; j < n_obs; ++j) {
            if (obs_var[j] < 1 || obs_var[j] >= h->V) return fail(RSCM_ERR_INVALID, "observation %d: variable %d has no stored series", j, obs_var[j]);
            if (obs_tidx[j] < 0 || obs_tidx[j] >= h->T) return fail(RSCM_ERR_INVALID, "observation %d: time index %d out of range", j, obs_tidx[j]);
            if (j > 0 && obs_var[j] != obs_var[j - 1])
                for (int32_t k = 0; k < j; ++k)
                    if (obs_var[k] == obs_var[j]) return fail(RSCM_ERR_INVALID, "observations must be grouped by variable");
        }
        if (int rc = set_device(h)) return rc;
        if (!h->d_loglik) HIPCHK(rscm::dev_malloc(&h->d_loglik, (size_t)h->N * sizeof(double)));
    }
    rscm_sampler* s = new rscm_sampler();
    s->ev = h;
    s->fused = fused;
    s->W = n_walkers;
    s->D = n_dims;
    s->rank = rank;
    s->n_ranks = n_ranks;
    s->sharded = n_ranks > 1;
    s->n_local = (int32_t)h->N;
    s->stretch_a = stretch_a;
    s->seed = seed;
    auto cleanup = [&](int rc) {
        rscm_sampler_destroy(s);
        return rc;
    };
    const size_t W = (size_t)n_walkers, H = (size_t)h->N, D = (size_t)n_dims;  // H: this rank's block of a half
#define CK(expr)                                                                               \
    do {                                                                                       \
        hipError_t e2_ = (expr);                                                               \
        if (e2_ != hipSuccess)                                                                 \
            return cleanup(fail(e2_ == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, \
                                "%s failed: %s", #expr, hipGetErrorString(e2_)));              \
    } while (0)
    CK(rscm::dev_malloc(&s->d_rows, D * sizeof(int32_t)));
    CK(rscm::dev_malloc(&s->d_kind, D * sizeof(int32_t)));
    CK(rscm::dev_malloc(&s->d_base, (size_t)h->P * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_pa, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_pb, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_plo, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_phi, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_pos, D * W * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_logp, W * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_prop, D * H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_z, H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_lp, H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_send, (D + 1) * H * sizeof(double)));   // 2 x (D + 1) x H doubles: also for one rank (the exchange of
    CK(rscm::dev_malloc(&s->d_recv, (size_t)n_ranks * (D + 1) * H * sizeof(double)));   // a one-rank group is a copy)
    CK(rscm::dev_malloc(&s->d_nacc, W * sizeof(int64_t)));
    CK(rscm::dev_malloc(&s->d_nprop, W * sizeof(int64_t)));
    CK(hipMemcpy(s->d_rows, param_rows, D * sizeof(int32_t), hipMemcpyHostToDevice));
    CK(hipMemcpy(s->d_kind, prior_kind, D * sizeof(int32_t), hipMemcpyHostToDevice));
    CK(hipMemcpy(s->d_base, base_params, (size_t)h->P * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(s->d_pa, prior_a, D * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(s->d_pb, prior_b, D * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<double> lo(D, -std::numeric_limits<double>::infinity()), hi(D, std::numeric_limits<double>::infinity());
        if (prior_low && prior_high)
            for (size_t d = 0; d < D; ++d) { lo[d] = prior_low[d]; hi[d] = prior_high[d]; }
        CK(hipMemcpy(s->d_plo, lo.data(), D * sizeof(double), hipMemcpyHostToDevice));
        CK(hipMemcpy(s->d_phi, hi.data(), D * sizeof(double), hipMemcpyHostToDevice));
    }
    CK(hipMemset(s->d_nacc, 0, W * sizeof(int64_t)));
    CK(hipMemset(s->d_nprop, 0, W * sizeof(int64_t)));
    CK(hipEventCreate(&s->ev0));
    CK(hipEventCreate(&s->ev1));
    if (!fused) {  // the observation rows of the stored series, once
        const size_t sz_ptr = (size_t)n_obs * sizeof(double*), sz_i = (size_t)n_obs * sizeof(int32_t),
                     sz_d = (size_t)n_obs * sizeof(double);
        const size_t off_val = sz_ptr, off_sig = off_val + sz_d, off_grp = off_sig + sz_d;
        std::vector<unsigned char> blob(off_grp + sz_i + 8);
        std::vector<const double*> ptrs(n_obs);
        for (int32_t j = 0; j < n_obs; ++j) ptrs[j] = h->series(obs_var[j]) + (size_t)obs_tidx[j] * h->N;
        if (n_obs > 0) {
            memcpy(blob.data(), ptrs.data(), sz_ptr);
            memcpy(blob.data() + off_val, obs_value, sz_d);
            memcpy(blob.data() + off_sig, obs_sigma, sz_d);
            memcpy(blob.data() + off_grp, obs_var, sz_i);
        }
        CK(rscm::dev_malloc(&s->d_sobs, blob.size()));
        CK(hipMemcpy(s->d_sobs, blob.data(), blob.size(), hipMemcpyHostToDevice));
        s->lik.n_members = h->N;
        s->lik.n_obs = n_obs;
        s->lik.normalize = normalize ? 1 : 0;
        s->lik.obs_series = (const double* const*)s->d_sobs;
        s->lik.obs_value = (const double*)((char*)s->d_sobs + off_val);
        s->lik.obs_sigma = (const double*)((char*)s->d_sobs + off_sig);
        s->lik.obs_group = (const int32_t*)((char*)s->d_sobs + off_grp);
        s->lik.out = h->d_loglik;
    }
#undef CK
    *out = s;
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_create_graph(rscm_ens* const* handles, int32_t n_handles, int32_t clear_between_runs, int32_t n_walkers, int32_t n_dims,
                              const int32_t* param_owner, const int32_t* param_rows, const int32_t* prior_kind, const double* prior_a,
                              const double* prior_b, const double* prior_low, const double* prior_high, int32_t n_obs,
                              const int32_t* obs_owner, const int32_t* obs_var, const int32_t* obs_tidx, const double* obs_value,
                              const double* obs_sigma, int32_t normalize, double stretch_a, uint64_t seed, int32_t rank, int32_t n_ranks,
                              rscm_sampler** out)
{
    GUARD_BEGIN
    if (!out) return fail(RSCM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!handles || n_handles < 1) return fail(RSCM_ERR_INVALID, "need at least one handle");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RSCM_ERR_INVALID, "bad rank %d of %d", rank, n_ranks);
    if (n_walkers < 2) return fail(RSCM_ERR_INVALID, "Must have at least 2 walkers");          // ensemble.rs:120-127
    if (n_walkers % 2) return fail(RSCM_ERR_INVALID, "Number of walkers must be even");
    if ((n_walkers / 2) % n_ranks)
        return fail(RSCM_ERR_INVALID, "half the walkers (%d) must split evenly over %d ranks", n_walkers / 2, n_ranks);
    if (!(stretch_a > 1.0)) return fail(RSCM_ERR_INVALID, "Stretch move scale parameter must be > 1.0, got %g", stretch_a);  // moves.rs:40-48
    if (n_dims < 1 || !param_owner || !param_rows || !prior_kind || !prior_a || !prior_b) return fail(RSCM_ERR_INVALID, "bad parameter description");
    rscm_ens* lead = handles[0];
    for (int32_t k = 0; k < n_handles; ++k) {
        rscm_ens* h = handles[k];
        NEED(h);
        if (h->N != n_walkers / 2 / n_ranks)
            return fail(RSCM_ERR_INVALID, "handle %d: every ensemble of the graph must have n_walkers / 2 / n_ranks = %d members, it has %lld", k,
                        n_walkers / 2 / n_ranks, (long long)h->N);
        if (h->windowed || h->rows != h->T) return fail(RSCM_ERR_INVALID, "handle %d: the sampler's evaluators store their whole series (no window, no RSCM_FLAG_NO_SERIES)", k);
        if (h->stream != lead->stream || h->device != lead->device) return fail(RSCM_ERR_STATE, "handle %d runs on another stream or device than handle 0", k);
        if (!h->params_set) return fail(RSCM_ERR_STATE, "handle %d: set the parameters once (rscm_ens_set_params) before sampling", k);
    }
    std::vector<uint64_t> sampled((size_t)n_handles, 0);
    for (int32_t d = 0; d < n_dims; ++d) {
        if (param_owner[d] < 0 || param_owner[d] >= n_handles) return fail(RSCM_ERR_INVALID, "dimension %d: owner %d out of range", d, param_owner[d]);
        const rscm_ens* h = handles[param_owner[d]];
        if (param_rows[d] < 0 || param_rows[d] >= h->P) return fail(RSCM_ERR_INVALID, "dimension %d: parameter row %d out of range", d, param_rows[d]);
        if (const char* name = structural_row(h->kind, param_rows[d]))
            return fail(RSCM_ERR_INVALID, "dimension %d: parameter row %d (%s) of handle %d is structural -- the host derives tables from it "
                        "when the parameters are set -- and cannot be sampled on the device", d, param_rows[d], name, param_owner[d]);
        for (int32_t e2 = 0; e2 < d; ++e2)
            if (param_owner[e2] == param_owner[d] && param_rows[e2] == param_rows[d])
                return fail(RSCM_ERR_INVALID, "parameter row %d of handle %d sampled twice", param_rows[d], param_owner[d]);
        if (prior_kind[d] < 0 || prior_kind[d] > 2) return fail(RSCM_ERR_INVALID, "dimension %d: unknown prior kind %d", d, prior_kind[d]);
        if (prior_kind[d] == 0 && !(prior_b[d] > prior_a[d])) return fail(RSCM_ERR_INVALID, "dimension %d: Uniform needs high > low", d);
        if (prior_kind[d] != 0 && !(prior_b[d] > 0.0)) return fail(RSCM_ERR_INVALID, "dimension %d: the scale parameter must be > 0", d);
        if ((prior_low && prior_high) && !(prior_low[d] < prior_high[d])) return fail(RSCM_ERR_INVALID, "dimension %d: Bound needs low < high", d);
        if (param_rows[d] < 64) sampled[(size_t)param_owner[d]] |= 1ull << param_rows[d];
    }
    if (n_obs < 0 || (n_obs > 0 && (!obs_owner || !obs_var || !obs_tidx || !obs_value || !obs_sigma))) return fail(RSCM_ERR_INVALID, "bad observation arrays");
    int32_t last_step = 0;
    for (int32_t j = 0; j < n_obs; ++j) {
        if (obs_owner[j] < 0 || obs_owner[j] >= n_handles) return fail(RSCM_ERR_INVALID, "observation %d: owner %d out of range", j, obs_owner[j]);
        const rscm_ens* h = handles[obs_owner[j]];
        if (obs_var[j] < 1 || obs_var[j] >= h->V) return fail(RSCM_ERR_INVALID, "observation %d: variable %d has no stored series", j, obs_var[j]);
        if (obs_tidx[j] < 0 || obs_tidx[j] >= h->T) return fail(RSCM_ERR_INVALID, "observation %d: time index %d out of range", j, obs_tidx[j]);
        last_step = std::max(last_step, obs_tidx[j]);
        // one contiguous run per (owner, variable), as rscm_sampler_create_sharded and the host likelihood require: the kernel closes a
        // partial sum at every change of group, so an interleaved order would change the association of the sum (likelihood.rs:206-248)
        if (j > 0 && (obs_owner[j] != obs_owner[j - 1] || obs_var[j] != obs_var[j - 1]))
            for (int32_t e2 = 0; e2 < j - 1; ++e2)
                if (obs_owner[e2] == obs_owner[j] && obs_var[e2] == obs_var[j])
                    return fail(RSCM_ERR_INVALID, "observation %d: the observations of (handle %d, variable %d) must be contiguous", j, obs_owner[j], obs_var[j]);
    }
    if (int rc = set_device(lead)) return rc;
    if (!lead->d_loglik) HIPCHK(rscm::dev_malloc(&lead->d_loglik, (size_t)lead->N * sizeof(double)));
    rscm_sampler* s = new rscm_sampler();
    s->ev = lead;
    s->fused = false;
    s->graph.assign(handles, handles + n_handles);
    s->graph_sampled_rows = sampled;
    s->graph_clear = clear_between_runs != 0;
    s->graph_last_step = last_step;
    s->W = n_walkers;
    s->D = n_dims;
    s->rank = rank;
    s->n_ranks = n_ranks;
    s->sharded = n_ranks > 1;
    s->n_local = (int32_t)lead->N;
    s->stretch_a = stretch_a;
    s->seed = seed;
    auto cleanup = [&](int rc) {
        rscm_sampler_destroy(s);
        return rc;
    };
    const size_t W = (size_t)n_walkers, H = (size_t)lead->N, D = (size_t)n_dims;
#define CK(expr)                                                                               \
    do {                                                                                       \
        hipError_t e2_ = (expr);                                                               \
        if (e2_ != hipSuccess)                                                                 \
            return cleanup(fail(e2_ == hipErrorOutOfMemory ? RSCM_ERR_NOMEM : RSCM_ERR_DEVICE, \
                                "%s failed: %s", #expr, hipGetErrorString(e2_)));              \
    } while (0)
    CK(rscm::dev_malloc(&s->d_kind, D * sizeof(int32_t)));
    CK(rscm::dev_malloc(&s->d_pa, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_pb, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_plo, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_phi, D * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_pos, D * W * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_logp, W * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_prop, D * H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_z, H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_lp, H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_send, (D + 1) * H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_recv, (size_t)n_ranks * (D + 1) * H * sizeof(double)));
    CK(rscm::dev_malloc(&s->d_nacc, W * sizeof(int64_t)));
    CK(rscm::dev_malloc(&s->d_nprop, W * sizeof(int64_t)));
    CK(rscm::dev_malloc((void**)&s->d_param_ptr, D * sizeof(double*)));
    {
        std::vector<double*> ptrs(D);
        for (size_t d = 0; d < D; ++d) ptrs[d] = handles[param_owner[d]]->d_params + (size_t)param_rows[d] * H;
        CK(hipMemcpy(s->d_param_ptr, ptrs.data(), D * sizeof(double*), hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(s->d_kind, prior_kind, D * sizeof(int32_t), hipMemcpyHostToDevice));
    CK(hipMemcpy(s->d_pa, prior_a, D * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(s->d_pb, prior_b, D * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<double> lo(D, -std::numeric_limits<double>::infinity()), hi(D, std::numeric_limits<double>::infinity());
        if (prior_low && prior_high)
            for (size_t d = 0; d < D; ++d) { lo[d] = prior_low[d]; hi[d] = prior_high[d]; }
        CK(hipMemcpy(s->d_plo, lo.data(), D * sizeof(double), hipMemcpyHostToDevice));
        CK(hipMemcpy(s->d_phi, hi.data(), D * sizeof(double), hipMemcpyHostToDevice));
    }
    CK(hipMemset(s->d_nacc, 0, W * sizeof(int64_t)));
    CK(hipMemset(s->d_nprop, 0, W * sizeof(int64_t)));
    CK(hipEventCreate(&s->ev0));
    CK(hipEventCreate(&s->ev1));
    {   // the observation rows where their owners store them; one partial sum per (owner, variable) in the caller's order
        const size_t sz_ptr = (size_t)n_obs * sizeof(double*), sz_i = (size_t)n_obs * sizeof(int32_t), sz_d = (size_t)n_obs * sizeof(double);
        const size_t off_val = sz_ptr, off_sig = off_val + sz_d, off_grp = off_sig + sz_d;
        std::vector<unsigned char> blob(off_grp + sz_i + 8);
        std::vector<const double*> ptrs((size_t)n_obs);
        std::vector<int32_t> grp((size_t)n_obs);
        for (int32_t j = 0; j < n_obs; ++j) {
            const rscm_ens* h = handles[obs_owner[j]];
            ptrs[(size_t)j] = h->series(obs_var[j]) + (size_t)obs_tidx[j] * h->N;
            // the group id: the index of the run's first observation (unique per (owner, variable) whatever the variable count)
            grp[(size_t)j] = (j > 0 && obs_owner[j] == obs_owner[j - 1] && obs_var[j] == obs_var[j - 1]) ? grp[(size_t)j - 1] : j;
        }
        if (n_obs > 0) {
            memcpy(blob.data(), ptrs.data(), sz_ptr);
            memcpy(blob.data() + off_val, obs_value, sz_d);
            memcpy(blob.data() + off_sig, obs_sigma, sz_d);
            memcpy(blob.data() + off_grp, grp.data(), sz_i);
        }
        CK(rscm::dev_malloc(&s->d_sobs, blob.size()));
        CK(hipMemcpy(s->d_sobs, blob.data(), blob.size(), hipMemcpyHostToDevice));
        s->lik.n_members = lead->N;
        s->lik.n_obs = n_obs;
        s->lik.normalize = normalize ? 1 : 0;
        s->lik.obs_series = (const double* const*)s->d_sobs;
        s->lik.obs_value = (const double*)((char*)s->d_sobs + off_val);
        s->lik.obs_sigma = (const double*)((char*)s->d_sobs + off_sig);
        s->lik.obs_group = (const int32_t*)((char*)s->d_sobs + off_grp);
        s->lik.out = lead->d_loglik;
    }
#undef CK
    *out = s;
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_destroy(rscm_sampler* s)
{
    if (!s) return RSCM_OK;
    if (s->ev) {
        (void)hipStreamSynchronize(s->ev->stream);
        s->ev->loglik_stop_at_last_obs = false;
    }
    (void)hipFree(s->d_rows); (void)hipFree(s->d_kind); (void)hipFree(s->d_base); (void)hipFree(s->d_pa);
    (void)hipFree(s->d_pb); (void)hipFree(s->d_plo); (void)hipFree(s->d_phi); (void)hipFree(s->d_pos); (void)hipFree(s->d_logp); (void)hipFree(s->d_prop);
    (void)hipFree(s->d_z); (void)hipFree(s->d_lp); (void)hipFree(s->d_nacc); (void)hipFree(s->d_nprop);
    (void)hipFree(s->d_sobs);
    (void)hipFree((void*)s->d_param_ptr);
    (void)hipFree(s->d_send);
    (void)hipFree(s->d_recv);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
    return RSCM_OK;
}

// what must hold before a half-step is enqueued
static int sampler_ready(rscm_sampler* s)
{
    rscm_ens* h = s->ev;
    if (!s->graph.empty()) return set_device(h);   // rscm_ens_run_lockstep checks every handle of the graph when it runs
    h->time_index = 0;
    return check_loglik_ready(h);
}

int rscm_sampler_set_groups(rscm_sampler* s, int32_t n_groups)
{
    GUARD_BEGIN
    if (!s) return fail(RSCM_ERR_INVALID, "sampler is NULL");
    if (n_groups < 1 || s->W % n_groups != 0 || (s->W / n_groups) % 2 != 0 || s->W / n_groups < 2)
        return fail(RSCM_ERR_INVALID, "%d walkers do not split into %d groups of an even number (>= 2) of walkers", s->W, n_groups);
    if ((s->n_ranks > 1 || s->sharded) && n_groups != 1) return fail(RSCM_ERR_INVALID, "a sharded sampler runs one ensemble (n_groups = 1)");
    s->groups = n_groups;
    s->positions_set = false;  // positions are scored per group layout: set them again
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_set_positions(rscm_sampler* s, const double* positions)
{
    GUARD_BEGIN
    if (!s || !positions) return fail(RSCM_ERR_INVALID, "sampler or positions is NULL");
    rscm_ens* h = s->ev;
    if (!h->params_set)  // kinds with structural rows (ClimateUDEB) are configured by rscm_ens_set_params
        return fail(RSCM_ERR_STATE, "set the evaluator's parameters once (rscm_ens_set_params) before sampling");
    if (int rc = sampler_ready(s)) return rc;
    const size_t W = (size_t)s->W, D = (size_t)s->D;
    std::vector<double> soa(D * W);  // [W][D] row-major in, [D][W] on the device
    for (size_t w = 0; w < W; ++w)
        for (size_t d = 0; d < D; ++d) soa[d * W + w] = positions[w * D + d];
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(s->d_pos, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(s->d_nacc, 0, W * sizeof(int64_t)));
    HIPCHK(hipMemset(s->d_nprop, 0, W * sizeof(int64_t)));
    s->iteration = 0;
    s->positions_set = true;
    if (s->sharded) {  // scored half by half through rscm_sampler_half_step(identity = 1) + the exchange
        HIPCHK(rscm::launch_fill(s->d_logp, (int64_t)W, -std::numeric_limits<double>::infinity(), h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return RSCM_OK;
    }
    for (int32_t half = 0; half < 2; ++half)
        if (int rc = sampler_half_step(s, half, 1)) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_iterate(rscm_sampler* s, int32_t n_iterations)
{
    GUARD_BEGIN
    if (!s) return fail(RSCM_ERR_INVALID, "sampler is NULL");
    if (!s->positions_set) return fail(RSCM_ERR_STATE, "walker positions not set");
    if (n_iterations < 0) return fail(RSCM_ERR_INVALID, "n_iterations must be >= 0");
    if (s->sharded)
        return fail(RSCM_ERR_STATE, "a sharded sampler is driven half-step by half-step (rscm_sampler_half_step, all-gather, "
                                    "rscm_sampler_apply_exchange)");
    rscm_ens* h = s->ev;
    if (int rc = sampler_ready(s)) return rc;
    HIPCHK(hipEventRecord(s->ev0, h->stream));
    for (int32_t it = 0; it < n_iterations; ++it) {
        s->iteration += 1;
        // first half against the second, then the second against the updated first (ensemble.rs:509-515)
        if (int rc = sampler_half_step(s, 0, 0)) return rc;
        if (int rc = sampler_half_step(s, 1, 0)) return rc;
    }
    HIPCHK(hipEventRecord(s->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_begin_iteration(rscm_sampler* s)
{
    if (!s) return fail(RSCM_ERR_INVALID, "sampler is NULL");
    if (!s->positions_set) return fail(RSCM_ERR_STATE, "walker positions not set");
    s->iteration += 1;
    return RSCM_OK;
}

int rscm_sampler_half_step(rscm_sampler* s, int32_t half, int32_t identity)
{
    GUARD_BEGIN
    if (!s) return fail(RSCM_ERR_INVALID, "sampler is NULL");
    if (!s->positions_set) return fail(RSCM_ERR_STATE, "walker positions not set");
    if (half != 0 && half != 1) return fail(RSCM_ERR_INVALID, "half must be 0 or 1");
    if (int rc = sampler_ready(s)) return rc;
    return sampler_half_step(s, half, identity ? 1 : 0);
    GUARD_END
}

int rscm_sampler_exchange_buffers(rscm_sampler* s, void** send, void** recv, int64_t* doubles_per_rank)
{
    if (!s || !send || !recv || !doubles_per_rank) return fail(RSCM_ERR_INVALID, "NULL argument");
    s->sharded = true;   // from here on the caller drives the half-steps and the exchange
    *send = s->d_send;
    *recv = s->d_recv;
    *doubles_per_rank = (int64_t)(s->D + 1) * s->n_local;
    return RSCM_OK;
}

int rscm_sampler_apply_exchange(rscm_sampler* s, int32_t half)
{
    GUARD_BEGIN
    if (!s) return fail(RSCM_ERR_INVALID, "sampler is NULL");
    if (!s->sharded) return fail(RSCM_ERR_STATE, "this sampler is not sharded (rscm_sampler_exchange_buffers first)");
    if (half != 0 && half != 1) return fail(RSCM_ERR_INVALID, "half must be 0 or 1");
    rscm::SamplerArgs a = sampler_args(s, half, 0);
    a.exchange = s->d_recv;
    HIPCHK(rscm::launch_sampler_unpack(a, s->ev->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_sync(rscm_sampler* s)
{
    GUARD_BEGIN
    if (!s) return fail(RSCM_ERR_INVALID, "sampler is NULL");
    HIPCHK(hipStreamSynchronize(s->ev->stream));
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_last_ms(const rscm_sampler* s, float* out)
{
    GUARD_BEGIN
    if (!s || !out) return fail(RSCM_ERR_INVALID, "sampler or out is NULL");
    HIPCHK(hipEventElapsedTime(out, s->ev0, s->ev1));
    return RSCM_OK;
    GUARD_END
}

int rscm_sampler_get(rscm_sampler* s, double* positions, double* log_prob, int64_t* n_accepted, int64_t* n_proposed)
{
    GUARD_BEGIN
    if (!s) return fail(RSCM_ERR_INVALID, "sampler is NULL");
    if (!s->positions_set) return fail(RSCM_ERR_STATE, "walker positions not set");
    const size_t W = (size_t)s->W, D = (size_t)s->D;
    HIPCHK(hipStreamSynchronize(s->ev->stream));
    if (positions) {
        std::vector<double> soa(D * W);
        HIPCHK(hipMemcpy(soa.data(), s->d_pos, soa.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t w = 0; w < W; ++w)
            for (size_t d = 0; d < D; ++d) positions[w * D + d] = soa[d * W + w];
    }
    if (log_prob) HIPCHK(hipMemcpy(log_prob, s->d_logp, W * sizeof(double), hipMemcpyDeviceToHost));
    if (n_accepted) HIPCHK(hipMemcpy(n_accepted, s->d_nacc, W * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (n_proposed) HIPCHK(hipMemcpy(n_proposed, s->d_nprop, W * sizeof(int64_t), hipMemcpyDeviceToHost));
    return RSCM_OK;
    GUARD_END
}

}  // extern "C"
