// Device code of the stateless pointwise components, shared by pointwise.hip (one launch per component)
// and group.hip (several linked components of one model step in one launch).  See pointwise.hip.
#pragma once

#include "rk4_device.hpp"
#include "rscm_device.hpp"

namespace rscm {
namespace pw {

template <int KIND>
struct Shape;
template <>
struct Shape<kKindOzoneForcing> { static constexpr int P = 13, NI = 6, NO = 3; };
template <>
struct Shape<kKindAerosolDirect> { static constexpr int P = 27, NI = 4, NO = 4; };
template <>
struct Shape<kKindAerosolIndirect> { static constexpr int P = 9, NI = 2, NO = 1; };
template <>
struct Shape<kKindFourBoxOhu> { static constexpr int P = 4, NI = 1, NO = 4; };
template <>
struct Shape<kKindOspp> { static constexpr int P = 13, NI = 2, NO = 1; };
template <>
struct Shape<kKindCo2Erf> { static constexpr int P = 2, NI = 1, NO = 1; };   // CO2ERF
template <>
struct Shape<kKindAggregate> { static constexpr int P = 9, NI = 8, NO = 1; };   // schema aggregate

constexpr double kLn2 = 0.693147180559945309417;  // 2.0_f64.ln()

// forcing/ozone.rs:99-164; in = {EESC, CH4, NOx, CO, NMVOC, temperature}
__device__ __forceinline__ void eval(const double (&p)[13], const double (&in)[6], double (&out)[3])
{
    const double delta_eesc = in[0] - p[0];
    // x^y for x > 0 as exp(y ln x): |y ln x| is O(1..10) here, so the power keeps ~1e-15 relative
    // accuracy at a third of the instructions of the general pow()
    out[0] = delta_eesc <= 0.0 ? 0.0 : p[1] * exp(p[2] * log_f64(delta_eesc / 100.0));
    const double ch4 = in[1];
    const double ch4_term = (ch4 > 0.0 && p[8] > 0.0) ? p[4] * log_f64(ch4 / p[8]) : 0.0;
    const double delta_nox = in[2] - p[9], delta_co = in[3] - p[10], delta_nmvoc = in[4] - p[11];
    const double precursor = p[5] * delta_nox + p[6] * delta_co + p[7] * delta_nmvoc;
    out[1] = p[3] * (ch4_term + precursor);
    out[2] = p[12] * in[5];
}

// forcing/aerosol_direct.rs:86-158; in = {SOx, BC, OC, NOx}; out = FourBox {NO, NL, SO, SL}
__device__ __forceinline__ void eval(const double (&p)[27], const double (&in)[4], double (&out)[4])
{
    const double sox = p[0] * (in[0] - p[20]);
    const double bc = p[1] * (in[1] - p[21]);
    const double oc = p[2] * (in[2] - p[22]);
    const double nit = p[3] * (in[3] - p[23]);
    const double total = sox + bc + oc + nit;
    const double total_abs = fabs(sox) + fabs(bc) + fabs(oc) + fabs(nit);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double weighted = (fabs(sox) * p[4 + i] + fabs(bc) * p[8 + i] + fabs(oc) * p[12 + i] + fabs(nit) * p[16 + i]) / total_abs;
        double r = total * weighted;
        if (total_abs < 1e-15) r = total / 4.0;
        if (fabs(total) < 1e-15) r = 0.0;
        out[i] = r;
    }
}

// forcing/aerosol_indirect.rs:75-115; in = {SOx, OC}
__device__ __forceinline__ void eval(const double (&p)[9], const double (&in)[2], double (&out)[1])
{
    const double burden = p[2] * in[0] + p[3] * in[1];
    const double burden_pi = p[2] * p[4] + p[3] * p[5];
    const double delta = burden - burden_pi;
    out[0] = delta <= 0.0 ? 0.0 : p[0] * log_f64(1.0 + delta / p[1]);
}

// four_box_ocean_heat_uptake.rs:85-112; in = {ERF|Aggregated}; out = FourBox {NO, NL, SO, SL}
__device__ __forceinline__ void eval(const double (&p)[4], const double (&in)[1], double (&out)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = in[0] * p[i];
}

// ocean_surface_partial_pressure.rs:57-122; in = {delta SST, delta DIC}.  As upstream: the factors
// are written 10e-3 .. 10e-10 and the fifth term carries the fourth power.
__device__ __forceinline__ void eval(const double (&p)[13], const double (&in)[2], double (&out)[1])
{
    const double d = in[1];
    const double d2 = d * d, d3 = d * d2, d4 = d2 * d2;
    const double bits[5] = {d, d2 * 10e-3, -d3 * 10e-5, d4 * 10e-7, -d4 * 10e-10};
    double delta = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) delta = delta + (p[3 + i] + p[8 + i] * p[2]) * bits[i];
    out[0] = (p[0] + delta) * exp(p[1] * in[0]);
}

// co2_erf.rs:57-60; p = {erf_2xco2, conc_pi}; in = {Atmospheric Concentration|CO2}
__device__ __forceinline__ void eval(const double (&p)[2], const double (&in)[1], double (&out)[1])
{
    out[0] = (p[0] / kLn2) * log_f64(1.0 + (in[0] - p[1]) / p[1]);   // (rk4_device.hpp: the coupled chain's logarithm, so that the linked chain carries its bits)
}

// compute_aggregate, schema.rs:760-802; params = {operation (0 Sum, 1 Mean, 2 Weighted), weights[8]}; up to
// eight contributors, read at_end() (AggregatorComponent, schema.rs:886-901).  NaN contributors are skipped
// and all-NaN gives NaN, so an unused row is simply a NaN row (the input block of this kind starts out all
// NaN); rows beyond the last contributor that was ever linked or set (n_inputs_used, wave-uniform) are NaN by
// construction and not read at all.  Operations 3-5 are the helper stages of a Mean over more than eight
// contributors (rscm_gpu.h): 3 counts the non-NaN rows, 4 the same with row 0 a count carried in, 5 divides
// row 0 (their sum) by row 1 (their number).
//
// Written apart from the other pointwise kinds: in a graph this component is pure overhead (one load, one
// add, one store) and what it costs is memory latency, so the rows in use are requested first, all together,
// then the nine parameters (scalar loads issued together when the block is uniform, as it is in practice: the
// operation is a scalar register then and branched on rather than selected on) -- two waits per step.
template <int SRC, class Cache = NoCache>
__device__ __forceinline__ void aggregate_body(const PointwiseArgs& a, int64_t i, int32_t step_begin, int32_t step_end, const Cache& cache = Cache())
{
    const int64_t N = a.n_members;
    const int32_t used = a.n_inputs_used;
    const MemberInputs<SRC, 8> inputs(a.inputs, a.scen, a.links, a.n_times, N, i);
    // The contributors of the first step are asked for BEFORE the operation is known (the operation is a scalar load the
    // branches below wait for; the rows then are on their way already), and all of a step's rows together: one wait.
    double v[8];
    auto fetch = [&](int32_t n) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = k < used ? inputs.at(k, n + 1, cache) : __builtin_nan("");
    };
    if (step_begin < step_end) fetch(step_begin);
    double prm[9];   // the operation and the eight weights, together
    cache.params(a.params, a.uniform_rows, N, i, prm);
    const int op = (int)prm[0];
    if (used == 1 && op < 2) {
        // One contributor, Sum or Mean (the total of a single forcing, say): 0.0 + v, and v / 1 is v.
        for (int32_t n = step_begin; n < step_end; ++n) {
            if (n > step_begin) fetch(n);
            const double result = v[0] == v[0] ? 0.0 + v[0] : __builtin_nan("");
            a.out[(a.rows > 1 ? (size_t)(n + 1) : (size_t)0) * N + i] = result;
            cache.put(0, result);
        }
        if (cache.last_step()) a.status[i] = 0;
        return;
    }
    double w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (op == 2 && k < used) ? prm[1 + k] : 0.0;
    if (cache.last_step()) a.status[i] = 0;
    for (int32_t n = step_begin; n < step_end; ++n) {
        if (n > step_begin) fetch(n);
        double result;
        if (__builtin_expect(op >= 3, 0)) {
            const double in0 = v[0];   // (NaN beyond the rows in use)
            if (op == 5) {
                const double in1 = v[1];
                result = in1 > 0.0 ? in0 / in1 : __builtin_nan("");
            } else {
                int cnt = (op == 3 && in0 == in0) ? 1 : 0;
#pragma unroll
                for (int k = 1; k < 8; ++k) cnt += v[k] == v[k];
                result = (op == 4 ? in0 : 0.0) + (double)cnt;
            }
        } else {
            // (rows beyond the ones in use are NaN here and skipped like any NaN contributor: the same sum)
            double s = 0.0;
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (v[k] == v[k]) {
                    s = s + (op == 2 ? v[k] * w[k] : v[k]);
                    ++cnt;
                }
            }
            if (op == 1) s = s / (double)cnt;
            result = cnt ? s : __builtin_nan("");
        }
        a.out[(a.rows > 1 ? (size_t)(n + 1) : (size_t)0) * N + i] = result;
        cache.put(0, result);
    }
}

// Member i of component KIND over the steps [step_begin, step_end).
template <int KIND, int SRC, class Cache = NoCache>
__device__ __forceinline__ void pointwise_body(const PointwiseArgs& a, int64_t i, int32_t step_begin, int32_t step_end, const Cache& cache = Cache())
{
    if constexpr (KIND == kKindAggregate) {
        aggregate_body<SRC>(a, i, step_begin, step_end, cache);
        return;
    } else {
    using S = Shape<KIND>;
    const int64_t N = a.n_members;
    const int32_t T = a.n_times;
    double p[S::P];
    cache.params(a.params, a.uniform_rows, N, i, p);
    const MemberInputs<SRC, S::NI> inputs(a.inputs, a.scen, a.links, T, N, i);
    const size_t var_stride = (size_t)a.rows * N;
    for (int32_t n = step_begin; n < step_end; ++n) {
        double in[S::NI], out[S::NO];
#pragma unroll
        for (int k = 0; k < S::NI; ++k) in[k] = inputs.at(k, n, cache);
        eval(p, in, out);
        const size_t r = (a.rows > 1 ? (size_t)(n + 1) : (size_t)0) * N + i;
#pragma unroll
        for (int o = 0; o < S::NO; ++o) {
            a.out[(size_t)o * var_stride + r] = out[o];
            cache.put(o, out[o]);
        }
    }
    if (cache.last_step()) a.status[i] = 0;   // (last: the byte's store may alias anything, no load moves across it)
    }
}

}  // namespace pw
}  // namespace rscm
