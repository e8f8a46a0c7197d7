// The whole-graph launch: every component of a linked graph -- the light ones AND ClimateUDEB and OceanCarbon --
// for every model step of a window chunk in ONE launch (gfx950).
//
// Model::run (crates/rscm-core/src/model/runtime.rs:504-527) steps any graph with one loop: for each step, for
// each component in graph order, solve.  As linked ensembles that was, for a graph holding a heavy component, five
// launches per model step (csrc/group.hip fuses the light components between the heavy ones), and ClimateUDEB's
// launch reloaded and stored its two ocean columns (2 x 50 doubles per member) and re-derived its per-member
// constants (parameters, the LAMCALC base solve) every step: 64 us of its 240 us per step at 125 000 members.
// Here thread i runs the reference's loop itself for member i:
//     for n in chunk:  for op in graph order:  body(op, member i, step n)
// with one thread per member and one wavefront per workgroup, as ClimateUDEB's own kernel has it.  Every graph edge
// is per member (rscm_ens_link_input), so a thread reads only what it wrote earlier in the launch (program order,
// same work-item) or what earlier launches wrote: the same computation as the per-step launches, bit for bit
// (tests/test_gpu_group.py, tests/test_gpu_window.py).
//   * ClimateUDEB (udeb_body.hpp, Udeb1): columns, scalars and the LAMCALC base solve live in registers / LDS across
//     the steps of the launch; HBM sees the columns once per launch.
//   * OceanCarbon: the O(T) recurrence of RSCM_MODE_FAST (ocean_body.hpp), one model step per call; its running
//     sums and the last 60 pulses go through HBM every step as in its one-step launches (81 doubles in, 33 out).
//   * the light components: group_body.hpp, with the thread-private LDS slots of the multi-step group kernel
//     (parameters that vary, the latest row of a series, linked values) after the first step of the launch.
// The launch runs at one wavefront per SIMD (ClimateUDEB's 512 registers): the light bodies, which live on
// occupancy when launched alone, here run between column solves that take 50x their time.
#include "group_body.hpp"
#include "ocean_body.hpp"
#include "udeb_body.hpp"

namespace rscm {

namespace {

constexpr int kKindUdeb = 2, kKindOceanCarbon = 11;   // RSCM_KIND_UDEB, RSCM_KIND_OCEAN_CARBON (rscm_gpu.cpp asserts)

// UFAST: ClimateUDEB's arithmetic mode (one kernel per mode: both inlined into one function cost the column solve 6 % more)
template <int NL, bool UFAST>
__global__ __launch_bounds__(kUdebBlock) void graph_kernel(const GraphHeavy hv, const GroupOp* __restrict__ ops, int32_t n_ops, int64_t n_members,
                                                           int32_t step_begin, int32_t step_end, unsigned long long* __restrict__ stamps)
{
    __shared__ double park[NL][kUdebBlock];
    extern __shared__ double lds_slots[];
    const int64_t i = (int64_t)blockIdx.x * kUdebBlock + threadIdx.x;
    if (i >= n_members) return;
    udeb::Udeb1<NL> ud(park);
    // diagnostic (rscm_gpu_graph_stamps, include/rscm_gpu_internal.h): shader cycles per component kind, summed over
    // the wavefronts, in stamps[kind] (slot 31: begin() / end()); null in normal runs -- one uniform branch per op
    unsigned long long t_mark = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int slot) {
        if (stamps) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) atomicAdd(&stamps[slot], t - t_mark);
            t_mark = __builtin_amdgcn_s_memtime();
        }
    };
    if (hv.has_udeb) ud.begin(hv.udeb, i);
    stamp(31);
    for (int32_t b = step_begin; b < step_end; ++b) {
        const bool last = b + 1 == step_end;
        // Nothing a light component or OceanCarbon loads per member (parameter rows, mostly) changes from step to step,
        // and the op table is read-only: left to itself the compiler hoists ~130 parameter loads of the eleven light
        // components out of the step loop and keeps them live across ClimateUDEB's column solves, which then spill to
        // scratch memory inside their unrolled loops (15 scratch accesses per solve, 1.6x the solve time).  The member
        // index those bodies see is opaque per step, so their loads stay where they are used.
        int32_t zero = 0;
        asm volatile("" : "+v"(zero));
        const int64_t ib = i + zero;
        for (int32_t k = 0; k < n_ops; ++k, stamp(ops[k - 1].kind & 31)) {
            const GroupOp& op = ops[k];
            if (op.kind == kKindUdeb) {
                // (slots 28 / 29: the step up to its sub-step loop, the loop; the rest of the step goes to the kind's slot)
                ud.template step<true, UFAST>(hv.udeb, b, [&](int where) { stamp(28 + where); });
            } else if (op.kind == kKindOceanCarbon) {
                ocean::ocean_recur_run<60, 2>(hv.ocean, hv.ocean.irf, hv.ocean.mode_table, ib, b, b + 1, hv.ocean.rebuild != 0 && b == step_begin);
            } else if (b > step_begin) {   // the slots hold what the previous step left
                run_op<true>(op, ib, b, b + 1, LdsCache<true, kUdebBlock>{lds_slots + threadIdx.x, op.cache, last});
            } else {
                run_op<true>(op, ib, b, b + 1, LdsCache<false, kUdebBlock>{lds_slots + threadIdx.x, op.cache, last});
            }
        }
    }
    if (hv.has_udeb) ud.end(hv.udeb);
    stamp(31);
}

}  // namespace

hipError_t launch_graph(const GraphHeavy& hv, const GroupOp* d_ops, int32_t n_ops, int64_t n_members, int32_t step_begin, int32_t step_end,
                        int32_t cache_slots, unsigned long long* stamps, hipStream_t s)
{
    if (n_ops <= 0 || n_members <= 0 || step_end <= step_begin) return hipSuccess;
    if (hv.has_udeb && (hv.udeb.n_layers != 50 || hv.udeb.step_begin != step_begin)) return hipErrorInvalidValue;
    if (hv.has_ocean && (!hv.ocean.recur || hv.ocean.near != 60 || hv.ocean.steps != 12)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n_members + kUdebBlock - 1) / kUdebBlock));
    const size_t lds = (size_t)cache_slots * kUdebBlock * sizeof(double);
    if (hv.has_udeb && hv.udeb.fast) hipLaunchKernelGGL((graph_kernel<50, true>), grid, dim3(kUdebBlock), lds, s, hv, d_ops, n_ops, n_members, step_begin, step_end, stamps);
    else hipLaunchKernelGGL((graph_kernel<50, false>), grid, dim3(kUdebBlock), lds, s, hv, d_ops, n_ops, n_members, step_begin, step_end, stamps);
    return hipGetLastError();
}

}  // namespace rscm
