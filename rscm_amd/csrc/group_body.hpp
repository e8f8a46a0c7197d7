// The per-member body of every light component kind, dispatched on an op of a fused launch's table (csrc/group.hip: the
// light components of one step, or a whole light graph over many steps).
#pragma once

#include "carbon_body.hpp"
#include "chem_body.hpp"
#include "ghg_body.hpp"
#include "pointwise_body.hpp"
#include "two_layer_body.hpp"

namespace rscm {

// FULL = false leaves out the register-hungry bodies (TerrestrialCarbon 134 VGPRs, CH4 110, GhgForcing 102,
// OzoneForcing 101, AerosolDirect 92, N2O 87): a segment made of box models, CO2ERF, budgets, aggregates and
// grid transforms only then runs at <= 96 registers, i.e. five wavefronts per SIMD instead of three -- these
// launches wait on four dependent memory round trips per step and need the occupancy to hide them.
template <bool FULL, class Cache>
__device__ __forceinline__ void run_op(const GroupOp& op, int64_t i, int32_t b, int32_t e, const Cache& cache)
{
    switch (op.kind) {
        case 0:  // RSCM_KIND_TWO_LAYER (forcing through L2: a linked series or the scenario table)
            if (op.variant == 0) tl::two_layer_body<0, false, true>(op.u.tl, nullptr, i, b, e, cache);
            else tl::two_layer_body<1, false, true>(op.u.tl, nullptr, i, b, e, cache);
            break;
        case 3:  // RSCM_KIND_GHG_FORCING, linked concentrations
            if constexpr (FULL) {
                if (op.variant == 0) ghg::ghg_body<0, false, true>(op.u.ghg, nullptr, i, b, e);
                else ghg::ghg_body<1, false, true>(op.u.ghg, nullptr, i, b, e);
            }
            break;
        case kKindOzoneForcing: if constexpr (FULL) pw::pointwise_body<kKindOzoneForcing, 2>(op.u.pw, i, b, e); break;
        case kKindAerosolDirect: if constexpr (FULL) pw::pointwise_body<kKindAerosolDirect, 2>(op.u.pw, i, b, e); break;
        case kKindAerosolIndirect: pw::pointwise_body<kKindAerosolIndirect, 2>(op.u.pw, i, b, e, cache); break;
        case kKindFourBoxOhu: pw::pointwise_body<kKindFourBoxOhu, 2>(op.u.pw, i, b, e, cache); break;
        case kKindOspp: pw::pointwise_body<kKindOspp, 2>(op.u.pw, i, b, e, cache); break;
        case kKindCo2Erf: pw::pointwise_body<kKindCo2Erf, 2>(op.u.pw, i, b, e, cache); break;
        case kKindAggregate: pw::pointwise_body<kKindAggregate, 2>(op.u.pw, i, b, e, cache); break;
        case kKindCh4Chemistry: if constexpr (FULL) chem::ch4_body<2>(op.u.chem, i, b, e); break;
        case kKindN2oChemistry: if constexpr (FULL) chem::n2o_body<2>(op.u.chem, i, b, e); break;
        case kKindCo2Budget: carbon::co2_budget_body<2>(op.u.carbon, i, b, e, cache); break;
        case kKindTerrestrialCarbon: if constexpr (FULL) carbon::terrestrial_body<2>(op.u.carbon, i, b, e); break;
        case kKindCarbonCycle: if (op.variant == 0) carbon::carbon_cycle_body<0, 2>(op.u.carbon, i, b, e, cache);
            else carbon::carbon_cycle_body<1, 2>(op.u.carbon, i, b, e, cache);
            break;
        default: break;
    }
}

}  // namespace rscm
