// CH4Chemistry and N2OChemistry ensemble kernels for gfx950 (MI355X), one thread per member.
//
// What they replace, per model step n (reference file:line):
//   CH4Chemistry::solve / solve_concentration / prather_iteration
//                                   crates/rscm-magicc/src/chemistry/ch4.rs:121-330
//   N2OChemistry::solve / solve_concentration / iteration
//                                   crates/rscm-magicc/src/chemistry/n2o.rs:96-260
// under the stepper conventions of crates/rscm-core/src/model/runtime.rs: emissions and
// temperature are exogenous series shared per scenario (index n); the concentration is the
// component's own state -- at_start() is index n, previous() index n-1, at_offset(-k) index n-k,
// with the reference's fall-backs when the history is shorter than the lag; concentration and
// lifetime are written at index n+1.
//
// The last two concentrations travel in registers; N2O's stratospheric-delay lookups read this
// thread's own earlier rows of the stored series (coalesced across the wavefront when the delay is
// uniform).  The arithmetic is the reference's, operation for operation; pow and exp come from
// the device math library (tests/test_gpu_chem.py states the tolerance).  Both kernels are
// VALU-bound: four Prather passes per member-year, each with one f64 pow and several divisions,
// against 16 B written.
#include "chem_body.hpp"

namespace rscm {

namespace {

template <int SRC>
__global__ __launch_bounds__(kBlock) void ch4_kernel(ChemArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    chem::ch4_body<SRC>(a, i, a.step_begin, a.step_end);
}

template <int SRC>
__global__ __launch_bounds__(kBlock) void n2o_kernel(ChemArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    chem::n2o_body<SRC>(a, i, a.step_begin, a.step_end);
}

}  // namespace

hipError_t launch_chem(const ChemArgs& a, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    if (a.kind == kKindCh4Chemistry) {
        RSCM_LAUNCH_BY_SOURCE(ch4_kernel, a, grid, dim3(kBlock), s, a);
    } else if (a.kind == kKindN2oChemistry) {
        RSCM_LAUNCH_BY_SOURCE(n2o_kernel, a, grid, dim3(kBlock), s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rscm
