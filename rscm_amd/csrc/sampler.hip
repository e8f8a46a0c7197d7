// Device-side affine-invariant stretch-move sampler (Goodman & Weare 2010) for the calibration
// loop: what the reference does on the host per iteration in
//   EnsembleSampler::run / update_group          crates/rscm-calibrate/src/sampler/ensemble.rs:496-547
//   StretchMove::propose / accept                crates/rscm-calibrate/src/sampler/moves.rs:40-125
//   ParameterSet::log_prior                      crates/rscm-calibrate/src/parameter_set.rs
// stays on the GPU here: proposals are written straight into the evaluating ensemble's parameter
// block, the fused run+likelihood kernel (two_layer.hip) scores them, and the accept step updates
// the walker positions -- no host round trip per half-ensemble.
//
// Random numbers: Philox4x32-10 keyed by the seed, counter = (iteration, half, walker, stream), so a
// run is reproducible and independent of launch geometry.  The reference draws from thread_rng:
// only the distribution is comparable (tests/test_gpu_sampler.py checks moments and the
// acceptance rule), as for the Latin hypercube.
#include "philox.hpp"
#include "rscm_device.hpp"

namespace rscm {

namespace {

// distribution.rs: Uniform :152-175, Normal :249-262, LogNormal :346-356, Bound :479-490 (the inner
// density, unnormalised, inside [lo, hi]; lo = -inf, hi = +inf when the prior is not truncated)
__device__ __forceinline__ double ln_prior(int32_t kind, double a, double b, double lo, double hi, double x)
{
    if (!(x >= lo && x <= hi)) return -__builtin_inf();
    if (kind == 0) {  // Uniform(low = a, high = b)
        if (!(x >= a && x <= b)) return -__builtin_inf();
        return -log(b - a);
    }
    if (kind == 2) {  // LogNormal(mu = a, sigma = b)
        if (x <= 0.0) return -__builtin_inf();
        const double ln_x = log(x);
        const double z = (ln_x - a) / b;
        return -0.5 * z * z - ln_x - log(b) - 0.5 * log(2.0 * 3.14159265358979323846);
    }
    // Normal(mean = a, std = b)
    const double z = (x - a) / b;
    return -0.5 * z * z - log(b) - 0.5 * log(2.0 * 3.14159265358979323846);
}

// Walker index of half-walker k of half `half`: independent ensembles ("groups") of Wg walkers each,
// laid out one after the other; each is split into its own two halves.
__device__ __forceinline__ int32_t walker_of(const SamplerArgs& a, int32_t k, int32_t half)
{
    const int32_t Wg = a.n_walkers / a.n_groups, Hg = Wg / 2;
    return (k / Hg) * Wg + half * Hg + (k % Hg);
}

// One thread per active walker of this rank's block of the half being updated: local index kl,
// global half-walker index k = k_offset + kl.  Everything random is keyed on k, so a chain does not
// depend on how the walkers are split over ranks.
__global__ __launch_bounds__(kBlock) void propose_kernel(SamplerArgs a)
{
    const int32_t kl = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    const int32_t H = a.n_local;
    if (kl >= H) return;
    const int32_t k = a.k_offset + kl;
    // each group draws complementary walkers from itself only
    const int32_t Wg = a.n_walkers / a.n_groups, Hg = Wg / 2;
    const int32_t g = k / Hg;
    const int32_t active = walker_of(a, k, a.half);
    double z = 1.0;
    int32_t comp = active;  // identity proposal: scores the walker where it stands
    if (!a.identity) {
        uint32_t c[4] = {(uint32_t)k, (uint32_t)a.iteration, (uint32_t)a.half, 0x57A7u};
        philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        const double u = u01_from_bits(c[0], c[1]);
        const double s = (a.stretch_a - 1.0) * u + 1.0;  // moves.rs:55-59: z = ((a-1) u + 1)^2 / a
        z = s * s / a.stretch_a;
        // a uniformly chosen walker of the complementary half (moves.rs:118-121)
        const uint32_t j = (uint32_t)((((uint64_t)c[2] << 32) | c[3]) % (uint64_t)Hg);
        comp = g * Wg + (1 - a.half) * Hg + (int32_t)j;
    }
    const int64_t W = a.n_walkers;
    double lp = 0.0;
    if (!a.param_ptr)
        for (int32_t r = 0; r < a.n_params; ++r) a.eval_params[(size_t)r * H + kl] = a.base_params[r];
    for (int32_t d = 0; d < a.n_dims; ++d) {
        const double x = a.pos[(size_t)d * W + active];
        const double cval = a.pos[(size_t)d * W + comp];
        const double y = a.identity ? x : cval + z * (x - cval);  // y = c + z (x - c)
        a.proposal[(size_t)d * H + kl] = y;
        lp += ln_prior(a.prior_kind[d], a.prior_a[d], a.prior_b[d], a.prior_lo[d], a.prior_hi[d], y);
    }
    // A proposal outside the prior's support is rejected whatever the model says
    // (ensemble.rs:143-177), so the model is not asked: the lane evaluates the walker's current,
    // valid position instead and the result is discarded.  Garbage parameters (negative heat
    // capacities ...) would push whole wavefronts onto the kernel's slow replay path.
    const bool in_support = lp > -__builtin_inf();
    for (int32_t d = 0; d < a.n_dims; ++d) {
        // one evaluating ensemble: row param_rows[d] of its block; a graph: the row of the ensemble that owns the parameter
        double* dst = a.param_ptr ? a.param_ptr[d] + kl : a.eval_params + (size_t)a.param_rows[d] * H + kl;
        *dst = (in_support || a.identity) ? a.proposal[(size_t)d * H + kl] : a.pos[(size_t)d * W + active];
    }
    a.z[kl] = z;
    a.lp[kl] = lp;
}

__global__ __launch_bounds__(kBlock) void accept_kernel(SamplerArgs a)
{
    const int32_t kl = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    const int32_t H = a.n_local;
    if (kl >= H) return;
    const int32_t k = a.k_offset + kl;
    const int32_t active = walker_of(a, k, a.half);
    const int64_t W = a.n_walkers;
    // log prior + log likelihood; anything failing is -inf (ensemble.rs:143-177)
    const double lp = a.lp[kl];
    double new_logp = lp + a.loglik[kl];
    if (!(lp > -__builtin_inf()) || new_logp != new_logp) new_logp = -__builtin_inf();
    if (a.identity) {
        a.logp[active] = new_logp;
        return;
    }
    uint32_t c[4] = {(uint32_t)k, (uint32_t)a.iteration, (uint32_t)a.half, 0xACCEu};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const double u = u01_from_bits(c[0], c[1]);
    // moves.rs:84-106: q = z^(d-1) p(y)/p(x); accept if u < min(1, q); a -inf proposal never
    const double log_ratio = (double)(a.n_dims - 1) * log(a.z[kl]) + (new_logp - a.logp[active]);
    const bool finite_new = new_logp > -__builtin_inf() && new_logp < __builtin_inf();
    const bool accept = finite_new && (u < exp(log_ratio));
    a.n_proposed[active] += 1;
    if (accept) {
        a.n_accepted[active] += 1;
        a.logp[active] = new_logp;
        for (int32_t d = 0; d < a.n_dims; ++d) a.pos[(size_t)d * W + active] = a.proposal[(size_t)d * H + kl];
    }
}

// Sharded sampler (one rank per GPU, each owning a block of both halves): after its accept step a
// rank packs the positions and log probabilities of its block of the updated half -- [D + 1][n_local]
// doubles, 2.8 MB per rank-half at 1e5 walkers x 6 dimensions on one rank, 350 KB on eight -- the ranks
// all-gather the blocks (RCCL), and every rank unpacks all of them into its replica of pos / logp: the
// complementary half the next half-step draws from is then the same everywhere.
__global__ __launch_bounds__(kBlock) void pack_kernel(SamplerArgs a)
{
    const int32_t kl = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    const int32_t H = a.n_local;
    if (kl >= H) return;
    const int32_t active = walker_of(a, a.k_offset + kl, a.half);
    const int64_t W = a.n_walkers;
    for (int32_t d = 0; d < a.n_dims; ++d) a.exchange[(size_t)d * H + kl] = a.pos[(size_t)d * W + active];
    a.exchange[(size_t)a.n_dims * H + kl] = a.logp[active];
}

__global__ __launch_bounds__(kBlock) void unpack_kernel(SamplerArgs a)
{
    const int32_t x = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    const int32_t H = a.n_local;
    if (x >= H * a.n_ranks) return;
    const int32_t r = x / H, kl = x - r * H;
    const int32_t active = walker_of(a, r * H + kl, a.half);   // rank r owns half-walkers [r H, (r + 1) H)
    const int64_t W = a.n_walkers;
    const double* blk = a.exchange + (size_t)r * (a.n_dims + 1) * H;
    for (int32_t d = 0; d < a.n_dims; ++d) a.pos[(size_t)d * W + active] = blk[(size_t)d * H + kl];
    a.logp[active] = blk[(size_t)a.n_dims * H + kl];
}

}  // namespace

hipError_t launch_sampler_propose(const SamplerArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(propose_kernel, dim3((unsigned)((a.n_local + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sampler_accept(const SamplerArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(accept_kernel, dim3((unsigned)((a.n_local + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sampler_pack(const SamplerArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((a.n_local + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sampler_unpack(const SamplerArgs& a, hipStream_t s)
{
    const int32_t n = a.n_local * a.n_ranks;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace rscm
