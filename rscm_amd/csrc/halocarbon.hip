// HalocarbonChemistry ensemble kernels for gfx950 (MI355X).
//
// What they replace, per model step n (reference file:line):
//   HalocarbonChemistry::solve / step_concentrations / decay_species
//                                    crates/rscm-magicc/src/chemistry/halocarbon.rs:79-98, 228-350
//   species_forcing, calculate_{total,fgas,montreal}_forcing, calculate_eesc   :100-226
//   HalocarbonParameters::emission_to_concentration_factor
//                                    crates/rscm-magicc/src/parameters/halocarbon.rs:66-77
// under the stepper conventions of crates/rscm-core/src/model/runtime.rs: the 41 emission series
// are exogenous and shared per scenario (index n), the 41 concentrations are the component's own
// states (index n), new concentrations and the four aggregates (formed from the NEW
// concentrations) land at index n+1.
//
// The 41 species are independent first-order decays, so the work is spread over
// (member, species) threads: 41 x N threads each carry one concentration through time in a
// register and stream its series to HBM (member fastest).  exp(-dt/tau) is re-evaluated only when
// the step length changes -- the same input gives the same bits.  A second kernel forms the four
// aggregates per (member, 16-year chunk): it walks the species in the reference's order, loading
// each species' parameters once per chunk, so every sum adds its terms in the reference's order.
// Both kernels are bound by the HBM stream of the 41 concentration series (written once, read
// once: ~700 B per member-year).  exp comes from the device math library: agreement with the
// CPU oracle is to its last-place error (tests/test_gpu_halocarbon.py).
#include "rscm_device.hpp"

namespace rscm {

namespace {

constexpr int kSpecies = 41, kFgases = 23, kGlobals = 6, kFields = 7, kChunk = 16;

template <bool HAS_SCEN>
__global__ __launch_bounds__(kBlock) void halo_species_kernel(HaloArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    const int s = blockIdx.y;
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    const double lifetime = P(kGlobals + s * kFields + 0), mol_weight = P(kGlobals + s * kFields + 3);
    // emission_to_concentration_factor
    const double atm_mass_g = P(4) * 1e12;
    const double conv = (P(3) / mol_weight) * (1e9 / atm_mass_g) * 1e12 / P(5);
    const double* __restrict__ e = a.emissions + ((HAS_SCEN ? (size_t)a.scen[i] : (size_t)0) * kSpecies + s) * T;
    double* __restrict__ c_series = a.series + (size_t)s * a.rows * N + i;
    if (s == 0) a.status[i] = 0;
    double c = c_series[(size_t)a.step_begin * N];
    double dt_prev = __builtin_nan(""), decay = 0.0;
    for (int32_t n = a.step_begin; n < a.step_end; ++n) {
        const double dt = a.bounds[n + 1] - a.bounds[n];
        if (dt != dt_prev) {  // uniform branch: the time axis is shared
            decay = exp(-dt / lifetime);
            dt_prev = dt;
        }
        const double emissions_ppt = e[n] * conv;
        c = c * decay + emissions_ppt * lifetime * (1.0 - decay);
        c_series[(size_t)(n + 1) * N] = c;
    }
}

__global__ __launch_bounds__(kBlock) void halo_aggregate_kernel(HaloArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    const int64_t N = a.n_members;
    const int32_t row0 = a.step_begin + 1 + (int32_t)blockIdx.y * kChunk;  // first output row of this chunk
    if (row0 > a.step_end) return;
    const int32_t rows = (a.step_end - row0 + 1) < kChunk ? (a.step_end - row0 + 1) : kChunk;
    auto P = [&](int j) -> double { return param_at(a.params, a.uniform_rows, j, N, i); };
    const double br_mult = P(0), cfc11_norm = P(1);
    const size_t vs = (size_t)a.rows * N;
    double total[kChunk], fgas[kChunk], montreal[kChunk], eesc[kChunk];
#pragma unroll
    for (int y = 0; y < kChunk; ++y) total[y] = fgas[y] = montreal[y] = eesc[y] = 0.0;
    for (int s = 0; s < kSpecies; ++s) {
        const int b = kGlobals + s * kFields;
        const double radeff = P(b + 1), conc_pi = P(b + 2), n_cl = P(b + 4), n_br = P(b + 5), release = P(b + 6);
        const bool releases = release > 0.0;
        const double halogen_loading = n_cl + br_mult * n_br;
        const double normalised_release = release / cfc11_norm;
        const double* __restrict__ c_series = a.series + (size_t)s * vs + i;
#pragma unroll
        for (int y = 0; y < kChunk; ++y) {
            if (y < rows) {
                const double c = c_series[(size_t)(row0 + y) * N];
                const double f = (c - conc_pi) * radeff / 1000.0;
                total[y] += f;
                if (s < kFgases) fgas[y] += f; else montreal[y] += f;
                if (releases) eesc[y] += c * halogen_loading * normalised_release;
            }
        }
    }
    double* out = a.series + (size_t)kSpecies * vs + i;
#pragma unroll
    for (int y = 0; y < kChunk; ++y) {
        if (y < rows) {
            const size_t r = (size_t)(row0 + y) * N;
            out[r] = total[y];
            out[vs + r] = fgas[y];
            out[2 * vs + r] = montreal[y];
            out[3 * vs + r] = eesc[y];
        }
    }
}

}  // namespace

hipError_t launch_halocarbon(const HaloArgs& a, hipStream_t s)
{
    if (a.step_end <= a.step_begin || a.n_members <= 0) return hipSuccess;
    const unsigned gx = (unsigned)((a.n_members + kBlock - 1) / kBlock);
    if (a.scen) hipLaunchKernelGGL((halo_species_kernel<true>), dim3(gx, kSpecies), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((halo_species_kernel<false>), dim3(gx, kSpecies), dim3(kBlock), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned chunks = (unsigned)((a.step_end - a.step_begin + kChunk - 1) / kChunk);
    hipLaunchKernelGGL(halo_aggregate_kernel, dim3(gx, chunks), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace rscm
