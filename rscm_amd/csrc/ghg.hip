// GhgForcing ensemble kernel for gfx950 (MI355X): rscm-magicc's well-mixed greenhouse gas
// forcing, one thread per ensemble member.
//
// What it replaces, per model step n (reference file:line):
//   GhgForcing::solve / calculate_forcings     crates/rscm-magicc/src/forcing/ghg.rs:272-345
//   IPCCTAR (Myhre et al. 1998) and OLBL forms  forcing/ghg.rs:119-269
// under the stepper conventions of crates/rscm-core/src/model/runtime.rs: the concentrations are
// exogenous (index n), the three ERFs land at index n+1; index 0 keeps what rscm_ens_set_initial put there (NaN otherwise, builder.rs:772-790).
//
// A stateless pointwise component: the concentrations are shared by every member of a scenario,
// only the 21 parameters are per member.  Everything that depends on the concentrations alone --
// their square roots, ln CO2, the 0.75 and 1.52 powers of the Myhre overlap term -- is a
// scenario table the host builds once per rscm_ens_set_forcing (rscm_gpu.cpp, ghg_tables), and
// everything that depends on the parameters alone is formed once per member at the top of the
// launch.  What is left per member-year is a dozen multiply-adds (OLBL) or that plus two
// logarithms (IPCCTAR) and 24 bytes of output: the kernel is bound by the HBM write stream.
// The factorisations ln(C/C0) = ln C - ln C0 and (M N)^p = M^p N^p round differently from the
// reference's expressions; agreement with the CPU oracle is to ~1e-15 absolute
// (tests/test_gpu_ghg.py states 1e-12), not bit for bit.
#include "ghg_body.hpp"

namespace rscm {

namespace {

template <int METHOD, bool HAS_SCEN, bool LINKED>
__global__ __launch_bounds__(kBlock) void ghg_kernel(GhgArgs a, const double* __restrict__ tables)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    ghg::ghg_body<METHOD, HAS_SCEN, LINKED>(a, tables, i, a.step_begin, a.step_end);
}

template <int METHOD>
__global__ __launch_bounds__(kBlock) void ghg_derive_kernel(const double* __restrict__ params, uint64_t uniform_rows, int64_t n_members, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_members) return;
    auto P = [&](int j) -> double { return param_at(params, uniform_rows, j, n_members, i); };
    double d[kDerivedRows];
    ghg::member_constants<METHOD>(P(1), P(2), P(3), P(7), P(8), P(10), d);
#pragma unroll
    for (int k = 0; k < kDerivedRows; ++k) out[(size_t)k * n_members + i] = d[k];
}

}  // namespace

uint64_t ghg_derive_sources(int32_t method)
{
    const uint64_t pre_industrial = (1ull << 1) | (1ull << 2) | (1ull << 3);
    return method == 0 ? pre_industrial : pre_industrial | (1ull << 7) | (1ull << 8) | (1ull << 10);
}

hipError_t launch_ghg_derive(const double* params, uint64_t uniform_rows, int32_t method, int64_t n_members, double* out, hipStream_t s)
{
    if (n_members <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_members + kBlock - 1) / kBlock));
    if (method == 0) hipLaunchKernelGGL(ghg_derive_kernel<0>, grid, dim3(kBlock), 0, s, params, uniform_rows, n_members, out);
    else hipLaunchKernelGGL(ghg_derive_kernel<1>, grid, dim3(kBlock), 0, s, params, uniform_rows, n_members, out);
    return hipGetLastError();
}

hipError_t launch_ghg(const GhgArgs& a, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    if (!a.derived) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    const bool scen = a.scen != nullptr;
    if (a.linked) {
        if (a.method == 0) hipLaunchKernelGGL((ghg_kernel<0, false, true>), grid, dim3(kBlock), 0, s, a, a.tables);
        else hipLaunchKernelGGL((ghg_kernel<1, false, true>), grid, dim3(kBlock), 0, s, a, a.tables);
    } else if (a.method == 0) {
        if (scen) hipLaunchKernelGGL((ghg_kernel<0, true, false>), grid, dim3(kBlock), 0, s, a, a.tables);
        else hipLaunchKernelGGL((ghg_kernel<0, false, false>), grid, dim3(kBlock), 0, s, a, a.tables);
    } else {
        if (scen) hipLaunchKernelGGL((ghg_kernel<1, true, false>), grid, dim3(kBlock), 0, s, a, a.tables);
        else hipLaunchKernelGGL((ghg_kernel<1, false, false>), grid, dim3(kBlock), 0, s, a, a.tables);
    }
    return hipGetLastError();
}

}  // namespace rscm
