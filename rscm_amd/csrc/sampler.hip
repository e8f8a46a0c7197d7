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

// One thread per active walker k of the half being updated.
__global__ __launch_bounds__(kBlock) void propose_kernel(SamplerArgs a)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    const int32_t H = a.n_walkers / 2;
    if (k >= H) return;
    // independent ensembles ("groups") of Wg walkers each, laid out one after the other; each is
    // split into its own two halves and draws complementary walkers from itself only
    const int32_t Wg = a.n_walkers / a.n_groups, Hg = Wg / 2;
    const int32_t g = k / Hg, jg = k % Hg;
    const int32_t active = g * Wg + a.half * Hg + jg;
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
    for (int32_t r = 0; r < a.n_params; ++r) a.eval_params[(size_t)r * H + k] = a.base_params[r];
    for (int32_t d = 0; d < a.n_dims; ++d) {
        const double x = a.pos[(size_t)d * W + active];
        const double cval = a.pos[(size_t)d * W + comp];
        const double y = a.identity ? x : cval + z * (x - cval);  // y = c + z (x - c)
        a.proposal[(size_t)d * H + k] = y;
        lp += ln_prior(a.prior_kind[d], a.prior_a[d], a.prior_b[d], a.prior_lo[d], a.prior_hi[d], y);
    }
    // A proposal outside the prior's support is rejected whatever the model says
    // (ensemble.rs:143-177), so the model is not asked: the lane evaluates the walker's current,
    // valid position instead and the result is discarded.  Garbage parameters (negative heat
    // capacities ...) would push whole wavefronts onto the kernel's slow replay path.
    const bool in_support = lp > -__builtin_inf();
    for (int32_t d = 0; d < a.n_dims; ++d)
        a.eval_params[(size_t)a.param_rows[d] * H + k] =
            (in_support || a.identity) ? a.proposal[(size_t)d * H + k] : a.pos[(size_t)d * W + active];
    a.z[k] = z;
    a.lp[k] = lp;
}

__global__ __launch_bounds__(kBlock) void accept_kernel(SamplerArgs a)
{
    const int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    const int32_t H = a.n_walkers / 2;
    if (k >= H) return;
    const int32_t Wg = a.n_walkers / a.n_groups, Hg = Wg / 2;
    const int32_t active = (k / Hg) * Wg + a.half * Hg + (k % Hg);
    const int64_t W = a.n_walkers;
    // log prior + log likelihood; anything failing is -inf (ensemble.rs:143-177)
    const double lp = a.lp[k];
    double new_logp = lp + a.loglik[k];
    if (!(lp > -__builtin_inf()) || new_logp != new_logp) new_logp = -__builtin_inf();
    if (a.identity) {
        a.logp[active] = new_logp;
        return;
    }
    uint32_t c[4] = {(uint32_t)k, (uint32_t)a.iteration, (uint32_t)a.half, 0xACCEu};
    philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const double u = u01_from_bits(c[0], c[1]);
    // moves.rs:84-106: q = z^(d-1) p(y)/p(x); accept if u < min(1, q); a -inf proposal never
    const double log_ratio = (double)(a.n_dims - 1) * log(a.z[k]) + (new_logp - a.logp[active]);
    const bool finite_new = new_logp > -__builtin_inf() && new_logp < __builtin_inf();
    const bool accept = finite_new && (u < exp(log_ratio));
    a.n_proposed[active] += 1;
    if (accept) {
        a.n_accepted[active] += 1;
        a.logp[active] = new_logp;
        for (int32_t d = 0; d < a.n_dims; ++d) a.pos[(size_t)d * W + active] = a.proposal[(size_t)d * H + k];
    }
}

}  // namespace

hipError_t launch_sampler_propose(const SamplerArgs& a, hipStream_t s)
{
    const int32_t H = a.n_walkers / 2;
    hipLaunchKernelGGL(propose_kernel, dim3((unsigned)((H + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sampler_accept(const SamplerArgs& a, hipStream_t s)
{
    const int32_t H = a.n_walkers / 2;
    hipLaunchKernelGGL(accept_kernel, dim3((unsigned)((H + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace rscm
