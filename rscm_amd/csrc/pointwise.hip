// Stateless pointwise components as ensemble kernels for gfx950 (MI355X), one thread per member:
// rscm-magicc's OzoneForcing, AerosolDirect, AerosolIndirect and rscm-components'
// FourBoxOceanHeatUptake, OceanSurfacePartialPressure.
//
// What they replace, per model step n (reference file:line):
//   OzoneForcing::solve / calculate_forcings      crates/rscm-magicc/src/forcing/ozone.rs:99-238
//   AerosolDirect::solve / calculate_forcing      crates/rscm-magicc/src/forcing/aerosol_direct.rs:86-239
//   AerosolIndirect::solve / calculate_forcing    crates/rscm-magicc/src/forcing/aerosol_indirect.rs:75-170
//   FourBoxOceanHeatUptake::solve                 crates/rscm-components/src/components/four_box_ocean_heat_uptake.rs:85-112
//   OceanSurfacePartialPressure::solve            crates/rscm-components/src/components/ocean_carbon_cycle/ocean_surface_partial_pressure.rs:57-122
// under the stepper conventions of crates/rscm-core/src/model/runtime.rs: inputs are exogenous
// series shared per scenario (index n), outputs land at index n+1; index 0 keeps what rscm_ens_set_initial put there (NaN otherwise, builder.rs:772-790).
//
// The expressions are the reference's, operation for operation (no contraction: the build uses
// -ffp-contract=off); pow and log come from the device math library, so agreement with the CPU
// oracle is to their last-place error (tests/test_gpu_pointwise.py states 1e-12), exact where no
// transcendental is involved (AerosolDirect, FourBoxOceanHeatUptake, the ozone temperature
// feedback).
// Rooflines: 8-32 B written per member-year; OzoneForcing carries one f64 pow and one log per
// member-year and is VALU-bound, the two aerosol kernels are bound by the HBM write stream.
#include "pointwise_body.hpp"

namespace rscm {

namespace {

template <int KIND, int SRC>
__global__ __launch_bounds__(kBlock) void pointwise_kernel(PointwiseArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    pw::pointwise_body<KIND, SRC>(a, i, a.step_begin, a.step_end);
}

template <int KIND>
hipError_t launch_kind(const PointwiseArgs& a, hipStream_t s)
{
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    if (a.linked) hipLaunchKernelGGL((pointwise_kernel<KIND, 2>), grid, dim3(kBlock), 0, s, a);
    else if (a.scen) hipLaunchKernelGGL((pointwise_kernel<KIND, 1>), grid, dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((pointwise_kernel<KIND, 0>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_pointwise(const PointwiseArgs& a, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    switch (a.kind) {
        case kKindOzoneForcing: return launch_kind<kKindOzoneForcing>(a, s);
        case kKindAerosolDirect: return launch_kind<kKindAerosolDirect>(a, s);
        case kKindAerosolIndirect: return launch_kind<kKindAerosolIndirect>(a, s);
        case kKindFourBoxOhu: return launch_kind<kKindFourBoxOhu>(a, s);
        case kKindOspp: return launch_kind<kKindOspp>(a, s);
        case kKindCo2Erf: return launch_kind<kKindCo2Erf>(a, s);
        case kKindAggregate: return launch_kind<kKindAggregate>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace rscm
