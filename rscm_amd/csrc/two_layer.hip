// Two-layer energy-balance ensemble kernel for gfx950 (MI355X).
//
// One thread per ensemble member; the model's whole time loop, the RK4 sub-steps and the
// TwoLayer right-hand side are fused in that thread, so per member-year the only HBM traffic is
// the two 8-byte state stores (16 B) that the reference's stepper also produces
// (crates/rscm-core/src/model/runtime.rs:480 writes outputs at index n+1).
//
// What it replaces, per step n (reference file:line):
//   Model::step_model_component      crates/rscm-core/src/model/runtime.rs:368-497
//   TwoLayer::solve                  crates/rscm-two-layer/src/component.rs:223-251
//   IVPBuilder::to_rk4 + Rk4 (10 sub-steps of h = 0.1 for an annual step)
//                                    crates/rscm-core/src/ivp/mod.rs:245-253
//   TwoLayer::calculate_dy_dt        crates/rscm-two-layer/src/component.rs:159-189
// The third integrand (heat content, component.rs:186-187) is integrated from 0 and dropped by
// the reference (:236,245-248); it cannot influence Ts/Td and is not computed here.
//
// Layout: params [6][N], state series [T][N] (member fastest => a wavefront stores 512
// contiguous bytes per variable per year), shared forcing [S][T] staged once per workgroup in
// LDS (751 x 8 B = 6 KB per scenario) and read as a broadcast (one scenario) or a per-lane
// gather (scenario_of_member).  No inter-workgroup communication, so the blockIdx -> XCD
// round-robin needs no remap: every workgroup streams its own column block and re-reads only
// the (L2-resident) forcing.
//
// The year loop issues no vector-memory LOAD: forcing comes from LDS (or is prefetched one year
// ahead on the L2 path) and the sub-step counts through the scalar cache, so no s_waitcnt vmcnt
// ever waits behind the previous year's stores -- that wait was 31 % of wave time at
// 1.5 waves/SIMD (profiles/r1_exact_1e5_v1_pmc.txt).
//
// EXACT mode, speculative year: the RK4 sub-steps of a year run branch-free with
//   * n/C as q = n*r; rem = fma(-C,q,n); fma(rem,r,q)   (rk4_device.hpp: identical to IEEE
//     division when the numerator's biased exponent is in [512,1535] and C's in [895,1151]),
//   * k1 + k2*2 as fma(k2, 2, k1)                        (identical while 2*k2 cannot overflow),
// while one integer max3 per right-hand side accumulates whether every numerator stayed inside
// the window.  If any did not (zeros in the first year, a member overflowing towards inf, ...)
// the year is recomputed for those lanes from the saved state with the compiler's full IEEE
// division and unfused doubling.  Either way the stored bits equal the reference's arithmetic.
#include "two_layer_body.hpp"

namespace rscm {

namespace {

template <int MODE, bool LDS, bool STORE>
__global__ __launch_bounds__(kBlock) void two_layer_kernel(TwoLayerArgs a)
{
    extern __shared__ double lds_forcing[];
    if constexpr (LDS) {
        const int32_t len = a.step_end - a.step_begin;
        const int32_t total = a.n_scen * len;
        for (int32_t idx = threadIdx.x; idx < total; idx += kBlock) {
            const int32_t s = idx / len, k = idx - s * len;
            lds_forcing[idx] = a.forcing[(size_t)s * a.n_times + a.step_begin + a.src_off + k];
        }
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    tl::two_layer_body<MODE, LDS, STORE>(a, lds_forcing, i, a.step_begin, a.step_end);
}

}  // namespace

// (Round 5 built ONE persistent launch with a dependency-ordered work queue here -- tasks = (64-member block, chunk of steps) claimed by
// resident wavefronts, a block's chunk waiting only for the same block's previous chunk, the state handed over through device-scope
// accesses -- bit-identical to the plain launch, and removed it again: it never beat the two-stream cut of rscm_gpu.cpp.  The premise
// was wrong: 1e5 members are 1563 independent chains for 1024 SIMDs, a SIMD needs two to be saturated, and ONE wavefront alone runs at
// 0.93 of a saturated SIMD's rate (65 536 members: 1.505 ms; 131 072: 2.807) -- the configuration's own bound is 2.21 ms per pass, the
// cut runs take 2.31, the queue 2.45 with one wavefront per SIMD and worse with more (fewer blocks than wavefronts: every task waits for
// its predecessor).  DESIGN.md section 4.1, profiles/r5_queue_experiment.txt, commit history of this file.)

template <bool STORE>
static hipError_t launch_impl(const TwoLayerArgs& a, int mode, hipStream_t s)
{
    if (a.n_members <= 0) return hipSuccess;
    if (STORE && a.step_end <= a.step_begin) return hipSuccess;
    const size_t lds = a.lds_forcing ? (size_t)a.n_scen * (a.step_end - a.step_begin) * sizeof(double) : 0;
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    void (*kern)(TwoLayerArgs) =
        mode == 0 ? (a.lds_forcing ? two_layer_kernel<0, true, STORE> : two_layer_kernel<0, false, STORE>)
                  : (a.lds_forcing ? two_layer_kernel<1, true, STORE> : two_layer_kernel<1, false, STORE>);
    if (lds > (size_t)kMaxStaticLds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_two_layer(const TwoLayerArgs& a, int mode, hipStream_t s)
{
    return launch_impl<true>(a, mode, s);
}

hipError_t launch_two_layer_loglik(const TwoLayerArgs& a, int mode, hipStream_t s)
{
    return launch_impl<false>(a, mode, s);
}

}  // namespace rscm
