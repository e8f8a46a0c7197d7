// CO2Budget and TerrestrialCarbon ensemble kernels for gfx950 (MI355X), one thread per member.
//
// What they replace, per model step n (reference file:line):
//   CO2Budget::solve / solve_budget            crates/rscm-magicc/src/carbon/budget.rs:96-190
//   TerrestrialCarbon::solve / solve_pools     crates/rscm-magicc/src/carbon/terrestrial.rs:103-330
//   derived turnover times                     crates/rscm-magicc/src/parameters/terrestrial_carbon.rs:103-168
// under the stepper conventions of crates/rscm-core/src/model/runtime.rs: inputs are exogenous
// series shared per scenario (index n), the concentration / the four pools are the component's
// own state (index n, carried in registers across the launch), everything is written at index
// n+1, dt = bounds[n+1] - bounds[n].
//
// CO2Budget has no transcendental: its results carry the same bits as the CPU oracle's.
// TerrestrialCarbon evaluates one log and up to five exp per member-year from the device math
// library (tests/test_gpu_carbon.py states the tolerance).  Both stream their state rows to HBM
// (24 / 40 B per member-year); CO2Budget is bound by that write stream, TerrestrialCarbon by
// the transcendental VALU work.
#include "carbon_body.hpp"

namespace rscm {

namespace {

template <int SRC>
__global__ __launch_bounds__(kBlock) void carbon_cycle_kernel(CarbonArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    carbon::carbon_cycle_body<0, SRC>(a, i, a.step_begin, a.step_end);
}

template <int SRC>
__global__ __launch_bounds__(kBlock) void carbon_cycle_fast_kernel(CarbonArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    carbon::carbon_cycle_body<1, SRC>(a, i, a.step_begin, a.step_end);
}

template <int SRC>
__global__ __launch_bounds__(kBlock) void co2_budget_kernel(CarbonArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    carbon::co2_budget_body<SRC>(a, i, a.step_begin, a.step_end);
}

template <int SRC>
__global__ __launch_bounds__(kBlock) void terrestrial_kernel(CarbonArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    carbon::terrestrial_body<SRC>(a, i, a.step_begin, a.step_end);
}

__global__ __launch_bounds__(kBlock) void terrestrial_derive_kernel(const double* __restrict__ params, uint64_t uniform_rows, int64_t n_members,
                                                                    double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_members) return;
    auto P = [&](int j) -> double { return param_at(params, uniform_rows, j, n_members, i); };
    double d[kDerivedRows];
    carbon::terrestrial_member_constants(P(0), P(8), P(9), P(10), P(11), P(12), P(13), P(14), P(15), P(16), P(17), d);
#pragma unroll
    for (int k = 0; k < kDerivedRows; ++k) out[(size_t)k * n_members + i] = d[k];
}

}  // namespace

uint64_t terrestrial_derive_sources()
{
    uint64_t m = 1ull << 0;
    for (int j = 8; j <= 17; ++j) m |= 1ull << j;
    return m;
}

hipError_t launch_terrestrial_derive(const double* params, uint64_t uniform_rows, int64_t n_members, double* out, hipStream_t s)
{
    if (n_members <= 0) return hipSuccess;
    hipLaunchKernelGGL(terrestrial_derive_kernel, dim3((unsigned)((n_members + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, params, uniform_rows,
                       n_members, out);
    return hipGetLastError();
}

hipError_t launch_carbon(const CarbonArgs& a, int mode, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    if (a.kind == kKindCo2Budget) {
        RSCM_LAUNCH_BY_SOURCE(co2_budget_kernel, a, grid, dim3(kBlock), s, a);
    } else if (a.kind == kKindTerrestrialCarbon) {
        if (!a.derived) return hipErrorInvalidValue;
        RSCM_LAUNCH_BY_SOURCE(terrestrial_kernel, a, grid, dim3(kBlock), s, a);
    } else if (a.kind == kKindCarbonCycle) {
        if (mode != 0) RSCM_LAUNCH_BY_SOURCE(carbon_cycle_fast_kernel, a, grid, dim3(kBlock), s, a);
        else RSCM_LAUNCH_BY_SOURCE(carbon_cycle_kernel, a, grid, dim3(kBlock), s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rscm
