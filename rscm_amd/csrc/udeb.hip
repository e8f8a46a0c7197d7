// ClimateUDEB ensemble kernel for gfx950 (MI355X): rscm-magicc's 4-box upwelling-diffusion
// energy-balance model, one thread per ensemble member.
//
// What it replaces, per model step n (reference file:line):
//   ClimateUDEB::solve_impl            crates/rscm-magicc/src/climate/udeb/mod.rs:399-656
//   adjusted_ecs / LAMCALC re-solve    mod.rs:302-350, climate/lamcalc.rs
//   step_hemisphere (implicit 50-layer column, Thomas solve), update_upwelling, diagnostics
//                                      crates/rscm-magicc/src/climate/udeb/ocean_column.rs
//   thomas_solve, invert_4x4           crates/rscm-core/src/utils/linear_algebra.rs
// around the stepper conventions of crates/rscm-core/src/model/runtime.rs (ERF exogenous:
// at_start = F[n], at_end = F[n+1]; outputs written at n+1).
//
// A different kernel class from two_layer.hip: ~110 doubles of private state per member and a
// serial tridiagonal recurrence per hemisphere and sub-step.  Layout and placement:
//   * the two 50-layer columns stay on chip for the whole launch: the hemisphere being solved is
//     in registers (shared with the Thomas d' array), the other one is parked in the lane's LDS
//     slots (25.6 KB per wavefront) and the two are exchanged after every solve.  HBM sees the
//     columns once at the start (resume) and once at the end: ocean[hemi][layer][N], member
//     fastest.  Keeping them in HBM cost 19.2 KB of traffic per member-year and left the single
//     resident wavefront per SIMD waiting on it (profiles/r1_udeb_5e4.txt, the earlier layout);
//   * the Thomas work array c' (50 doubles) stays in registers (layer loops fully unrolled, NL is
//     a template constant); one copy of the solver serves both hemispheres;
//   * everything that depends only on the ocean geometry (area factors, 1 - relative depth, the
//     profile-advection weights built from the CMIP5 initial profiles) is a uniform table in the
//     kernel-argument segment, read through the scalar cache;
//   * the scalar state values live in registers across the whole launch and go to
//     `scal[kUdebScalars][N]` only at the end (resume); the temperature history for the
//     time-varying ECS is hist[T][N], consumed through a running window sum.
// Tolerance parity: quotients are products with refined reciprocals, sums of products are fused,
// the row coefficients are regrouped (step_hemisphere) and LAMCALC solves its 4x4 system by
// structured elimination.  Measured deviation from the CPU oracle over a 1024-member, 400-year
// ensemble: 5e-14 relative (scripts/udeb_deviation.py); tests/test_gpu_udeb.py states 1e-9.
// LAMCALC's convergence test and the |adjusted ECS - ECS| > 1e-10 switch are discontinuous in
// their inputs, so a member sitting within rounding of one of those thresholds may take the
// other branch than the reference for that year (an O(1e-3) relative change of lambda for one
// year); no such member occurs in the test ensembles.
#include <cstdlib>

#include "udeb_any_body.hpp"
#include "experiment_env.hpp"

namespace rscm {

namespace {

using namespace udeb;

// One thread per member (udeb_body.hpp: Udeb1): the launch of large ensembles.  DYN: NL is the capacity of the register-resident
// column, the layer count is a.n_layers <= NL (every count the fixed instances do not cover, up to kUdebMaxOnChipLayers).
template <int NL, bool FAST, bool DYN = false, int LOW = NL>
__global__ __launch_bounds__(kUdebBlock) void udeb_kernel(UdebArgs a)
{
    __shared__ double park[NL][kUdebBlock];
    const int64_t i = (int64_t)blockIdx.x * kUdebBlock + threadIdx.x;
    if (i >= a.n_members) return;
    Udeb1<NL, DYN, LOW> m(park);
    m.begin(a, i);
    if (m.status != 0) {  // the reference refuses to build this component: every output NaN
        for (int32_t n = a.step_begin; n < a.step_end; ++n) m.template step<true, FAST>(a, n);
        return;
    }
    for (int32_t n = a.step_begin; n < a.step_end; ++n) m.template step<false, FAST>(a, n);
    m.end(a);
}

// Two wavefronts per 64 members, one hemisphere each (udeb_body.hpp), one wavefront per SIMD.
template <int NL, bool FAST, bool DYN = false, int LOW = NL>
__global__ __launch_bounds__(kUdeb2Block) void udeb2_kernel(UdebArgs a)
{
    __shared__ Udeb2Lds lds;
    Udeb2<NL, DYN, LOW> m(lds);
    m.begin(a);
    for (int32_t n = a.step_begin; n < a.step_end; ++n) m.template step<FAST>(a, n);
    m.end(a);
}

// 65 .. kUdebMaxLdsLayers layers: a hemisphere per wavefront as above, the column in registers (it fills them) and the sweep's c' array
// in LDS (2 x 128 rows x 512 B = 128 KB per workgroup: one workgroup per CU, i.e. two of a CU's four SIMDs busy).  Measured, per
// 65 536 members x 750 years: 499 ms at 65 layers against 76.7 ms at 64 (profiles/r5_udeb_layer_counts.txt) -- 3.3x (128 layers) to
// 6.4x (65) the on-chip kernels' cost per layer, not the 2x the occupancy alone would give: col[128] fills the 256 VGPRs and the
// kernel spills 624-672 bytes per lane to scratch (`make check-udeb-scratch` reports it; profiles/r6_udeb_lds_65.txt: 11.3 GB written
// per launch against 3.2 GB at 50 layers, issue utilisation 0.25).  Still 1.6x (65 layers) to 2.5x (100) faster than the columns-in-HBM kernel.  No configuration
// of the reference or of MAGICC7 has more than 50 layers: recorded, not pursued.  The geometry table comes from device memory (6 KB: past the
// kernel-argument segment), through the scalar cache all the same.
template <bool FAST>
__global__ __launch_bounds__(kUdeb2Block) void udeb2_lds_kernel(UdebArgs a)
{
    __shared__ Udeb2Lds lds;
    extern __shared__ double ncp_slots[];   // [2][kUdebMaxLdsLayers][64]
    Udeb2<kUdebMaxLdsLayers, true, kUdebMaxOnChipLayers + 1, true> m(lds);
    m.ncp_lds = ncp_slots + (size_t)(threadIdx.x >> 6) * kUdebMaxLdsLayers * 64 + (threadIdx.x & 63);
    m.begin(a);
    for (int32_t n = a.step_begin; n < a.step_end; ++n) m.template step<FAST>(a, n);
    m.end(a);
}

// Any other layer count (>= 2): columns in HBM, plain loops (udeb_any_body.hpp).  256 threads: nothing lives in registers across rows.
template <bool FAST>
__global__ __launch_bounds__(256) void udeb_any_kernel(UdebArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_members) return;
    udeb::udeb_any_member<FAST>(a, i);
}

// The base LAMCALC solve of every member, once per parameter set (rscm_gpu.cpp, ensure_derived): rows of `out` as in udeb_body.hpp.
__global__ __launch_bounds__(256) void udeb_derive_kernel(const double* __restrict__ params, uint64_t uniform_rows, int64_t n_members,
                                                          double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_members) return;
    auto P = [&](int j) -> double { return param_at(params, uniform_rows, j, n_members, i); };
    udeb::UdebP p;
    udeb::fill_lamcalc_inputs(p, P(10), P(11), P(12), P(16), P(17), P(18), P(19), P(20), P(28), P(29), P(30), P(31));
    const udeb::LamResult r = udeb::lamcalc(p, p.ecs);
    out[i] = r.lam_o;
    out[(size_t)n_members + i] = r.lam_l;
    out[(size_t)2 * n_members + i] = r.eff;
    out[(size_t)3 * n_members + i] = r.ok ? 1.0 : 0.0;
#pragma unroll
    for (int k = 4; k < kDerivedRows; ++k) out[(size_t)k * n_members + i] = 0.0;
}

}  // namespace

uint64_t udeb_derive_sources()
{
    uint64_t m = 0;
    for (int j : {10, 11, 12, 16, 17, 18, 19, 20, 28, 29, 30, 31}) m |= 1ull << j;
    return m;
}

hipError_t launch_udeb_derive(const double* params, uint64_t uniform_rows, int64_t n_members, double* out, hipStream_t s)
{
    if (n_members <= 0) return hipSuccess;
    hipLaunchKernelGGL(udeb_derive_kernel, dim3((unsigned)((n_members + 255) / 256)), dim3(256), 0, s, params, uniform_rows, n_members, out);
    return hipGetLastError();
}

// 0: one thread per member, 2: a hemisphere per wavefront, 3: the columns-in-HBM kernel whatever the count (A/B runs and the
// bit-for-bit test of the runtime-count kernels against it); -1: by ensemble size
static thread_local int t_udeb_variant = -1;
void set_udeb_variant(int variant) { t_udeb_variant = variant; }

// LOW: the smallest layer count a runtime-count instance serves (the previous capacity + 1)
template <int NL, bool DYN, int LOW = NL>
static void launch_udeb_nl(const UdebArgs& a, int variant, hipStream_t s)
{
    const bool two_waves = variant == 2;
    if (two_waves) {
        const dim3 grid((unsigned)((a.n_members + 63) / 64));
        if (a.fast) hipLaunchKernelGGL((udeb2_kernel<NL, true, DYN, LOW>), grid, dim3(kUdeb2Block), 0, s, a);
        else hipLaunchKernelGGL((udeb2_kernel<NL, false, DYN, LOW>), grid, dim3(kUdeb2Block), 0, s, a);
    } else {
        const dim3 grid((unsigned)((a.n_members + kUdebBlock - 1) / kUdebBlock));
        if (a.fast) hipLaunchKernelGGL((udeb_kernel<NL, true, DYN, LOW>), grid, dim3(kUdebBlock), 0, s, a);
        else hipLaunchKernelGGL((udeb_kernel<NL, false, DYN, LOW>), grid, dim3(kUdebBlock), 0, s, a);
    }
}

// The layer counts whose columns stay on chip (registers + LDS) for a whole launch: 20 / 30 / 40 / 50 with the count compiled
// in, every other count up to kUdebMaxOnChipLayers in the next capacity's runtime-count instance (same statements per row).
bool udeb_layers_unrolled(int32_t n_layers)
{
    return n_layers >= 2 && n_layers <= kUdebMaxOnChipLayers;
}

// ... with the layer count a compile-time constant of the instance
bool udeb_layers_fixed(int32_t n_layers)
{
    return n_layers == 20 || n_layers == 30 || n_layers == 40 || n_layers == 50;
}

hipError_t launch_udeb(const UdebArgs& a, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    if (!a.derived) return hipErrorInvalidValue;
    // Up to 32 768 members there are fewer wavefronts than SIMDs either way: the two-wavefront kernel solves a member's
    // two hemispheres at the same time on two SIMDs (half the latency per model step); beyond that the two kernels
    // take the same time per member (profiles/r3_udeb_two_wave_experiment.txt) and the one-thread kernel is kept.
    // (A third kernel -- four wavefronts per 64 members, each hemisphere's column cut in the middle, two wavefronts per SIMD -- was built
    // in round 3 and removed in round 4: 71 ms against 54 ms at 65 536 members x 750 years, ahead only below ~4096 members, and its
    // twisted factorisation agrees with these two to rounding, not to the bit -- selecting it by ensemble size would have made a
    // member's bits depend on how many members run beside it.  profiles/r3_udeb4_65536.txt, DESIGN.md section 8, commit 03abd7b.)
    // rscm_gpu_set_udeb_variant (include/rscm_gpu_internal.h) forces one kernel for the calling thread (A/B runs, tests); the
    // experiments build also reads RSCM_UDEB_VARIANT = 0 / 2 / 3 for the whole process (experiment_env.hpp).
    static const int forced = (int)rscm::experiment_env("RSCM_UDEB_VARIANT", -1);
    int variant = t_udeb_variant >= 0 ? t_udeb_variant : forced;
    const bool in_hbm = variant == 3 || a.n_layers > kUdebMaxLdsLayers;
    if (!in_hbm && a.n_layers > kUdebMaxOnChipLayers) {   // c' in LDS, the two-wavefront shape at every ensemble size
        if (!a.tables_dev) return hipErrorInvalidValue;
        const size_t lds = (size_t)2 * kUdebMaxLdsLayers * 64 * sizeof(double);
        void (*kern)(UdebArgs) = a.fast ? udeb2_lds_kernel<true> : udeb2_lds_kernel<false>;
        {   // (every launch: the attribute belongs to the function ON THE CURRENT DEVICE, and a host may drive several)
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)((a.n_members + 63) / 64)), dim3(kUdeb2Block), lds, s, a);
        return hipGetLastError();
    }
    if (variant != 0 && variant != 2) variant = (a.n_total > 0 ? a.n_total : a.n_members) <= 32768 ? 2 : 0;
    if (in_hbm) {   // more layers than the registers hold (or asked for): columns in HBM, plain loops
        if (a.n_layers < 2 || !a.tables_dev || !a.work) return hipErrorInvalidValue;
        const dim3 grid((unsigned)((a.n_members + 255) / 256));
        if (a.fast) hipLaunchKernelGGL(udeb_any_kernel<true>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(udeb_any_kernel<false>, grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    switch (a.n_layers) {   // the column loops are unrolled, the column lives in registers
        case 20: launch_udeb_nl<20, false>(a, variant, s); break;
        case 30: launch_udeb_nl<30, false>(a, variant, s); break;
        case 40: launch_udeb_nl<40, false>(a, variant, s); break;
        case 50: launch_udeb_nl<50, false>(a, variant, s); break;
        default:   // any other count: the smallest capacity that holds it, the count at run time
            if (a.n_layers < 20) launch_udeb_nl<20, true, 2>(a, variant, s);
            else if (a.n_layers < 30) launch_udeb_nl<30, true, 21>(a, variant, s);
            else if (a.n_layers < 40) launch_udeb_nl<40, true, 31>(a, variant, s);
            else if (a.n_layers < 50) launch_udeb_nl<50, true, 41>(a, variant, s);
            else launch_udeb_nl<kUdebMaxOnChipLayers, true, 51>(a, variant, s);
    }
    return hipGetLastError();
}

}  // namespace rscm
