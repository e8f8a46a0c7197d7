// Internal launch interface between the C-ABI layer (rscm_gpu.cpp) and the gfx950 kernels.
// Not part of the public boundary (that is include/rscm_gpu.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rscm {

#ifndef RSCM_BLOCK
#define RSCM_BLOCK 256
#endif
// The RSCM_KIND_* values of include/rscm_gpu.h that the kernels dispatch on (the device code does not
// include the public header; rscm_gpu.cpp static_asserts that the two agree).
enum Kind : int {
    kKindOzoneForcing = 4, kKindAerosolDirect = 5, kKindAerosolIndirect = 6, kKindCh4Chemistry = 7, kKindN2oChemistry = 8,
    kKindCo2Budget = 9, kKindTerrestrialCarbon = 10, kKindFourBoxOhu = 13, kKindOspp = 14, kKindCarbonCycle = 15,
    kKindCo2Erf = 16, kKindAggregate = 17
};

constexpr int kBlock = RSCM_BLOCK;           // 256 = 4 wavefronts of 64: one per SIMD of a CU
constexpr int kMaxStaticLds = 64 * 1024;     // above this the launcher raises the dynamic limit
constexpr int kMaxLds = 160 * 1024;          // CDNA4: 160 KiB per CU

// Linked inputs (rscm_ens_link_input): input row k of a member is read from the stored series of
// another ensemble of the same shape -- row[k] is that series, [T][N], off[k] the index offset of
// the reference's VariableSource (0: Exogenous / OwnState -> index n, 1: UpstreamOutput -> n+1).
// Where a fused multi-step launch (csrc/group.hip) keeps an op's per-member values between model steps: LDS
// slots (one double per thread each) instead of the round trip through HBM.  -1: not kept.
struct OpCache {
    int32_t series_slot;   // first slot of the latest row of the op's own series (its state, and what consumers read)
    int32_t param_slot;    // first slot of its parameter rows
    int32_t link_slot[8];  // per input row: the slot of the producer's value this link reads at the current step
    uint32_t link_warm;    // bit k: link k reads what its producer kept in the PREVIOUS step (feedback): not at a launch's first step
};

// A null row[k] leaves row k with the scenario table.  Passed by value: the kernels read it from
// the kernarg segment with scalar loads.
constexpr int kMaxLinks = 8;
struct InputLinks {
    const double* row[kMaxLinks];
    int32_t off[kMaxLinks];
};

#ifdef __HIPCC__
// Parameter j of member i from the [P][N] block.  Rows the host found to be the same for every member
// (bit j of `uniform`, rows 0..63) are read from element 0 by every lane: one 64-byte request per
// wavefront from a line that stays hot in L1/L2 instead of 512 coalesced bytes from HBM -- in a graph
// of linked ensembles most rows are uniform (a calibration varies a handful), and the light
// components' launches are bound by exactly this traffic.  The value read is the same either way.
__device__ __forceinline__ double param_at(const double* __restrict__ params, uint64_t uniform, int j, int64_t N, int64_t i)
{
    const bool u = j < 64 && ((uniform >> (j & 63)) & 1ull) != 0;
    return params[(size_t)j * N + (u ? (int64_t)0 : i)];
}

// The same value, with a uniform row read through the constant address space: s_load_dwordx2 into scalar
// registers -- no vector memory instruction, no address arithmetic per lane, and code that branches on the
// value (the aggregate's operation) branches uniformly.  For the light bodies only: in the register-bound
// kernels (ClimateUDEB, OceanCarbon) dozens of parameters in scalar registers spill.  The block is not
// written while a kernel that reads it runs.
__device__ __forceinline__ double param_at_scalar(const double* __restrict__ params, uint64_t uniform, int j, int64_t N, int64_t i)
{
    const bool u = j < 64 && ((uniform >> (j & 63)) & 1ull) != 0;  // wave-uniform: a scalar branch
    if (u) {
        typedef const __attribute__((address_space(4))) double* scalar_row;
        // (the row address is wave-uniform; saying so keeps the scalar load legal where register pressure has moved the
        // pointer into vector registers -- without it hipcc 7.2 emits an s_load with a VGPR address there and stops with
        // "Illegal instruction detected: Operand has incorrect register class"; a no-op when the address is scalar already)
        const uintptr_t p = (uintptr_t)(params + (size_t)j * N);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        return *(scalar_row)(((uintptr_t)hi << 32) | lo);
    }
    return params[(size_t)j * N + i];
}

// All P parameter rows of a component at once.  Where every row is uniform (the usual case in a graph of linked ensembles:
// a calibration varies a handful of rows of a few components) these are P scalar loads issued back to back and awaited
// ONCE; asked for one by one through param_at_scalar each sits behind its own uniform-bit branch and is awaited before
// the next is issued -- 27 dependent trips through the scalar cache per model step for AerosolDirect, and the fused
// launches of the light components spend two thirds of their time in s_waitcnt (gpurun_out/r3e).
// Mixed blocks take the vector loads of param_at: no branches either, one wait.
template <int P>
__device__ __forceinline__ void params_block(const double* __restrict__ params, uint64_t uniform, int64_t N, int64_t i, double (&p)[P])
{
    constexpr uint64_t mask = P >= 64 ? ~0ull : ((1ull << P) - 1ull);
    if ((uniform & mask) == mask) {
        typedef const __attribute__((address_space(4))) double* scalar_row;
        const uintptr_t q = (uintptr_t)params;   // (wave-uniform; said so for the reason given in param_at_scalar)
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
        const scalar_row row0 = (scalar_row)(((uintptr_t)hi << 32) | lo);
#pragma unroll
        for (int j = 0; j < P; ++j) p[j] = row0[(size_t)j * N];
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) p[j] = param_at(params, uniform, j, N, i);
    }
}

// How a body reads parameters, its own previous state and linked inputs, and where it leaves its results
// besides the series in HBM.  The stand-alone kernels use NoCache (everything compiles to the plain loads);
// the fused multi-step launch passes an LdsCache.
struct NoCache {
    static constexpr bool kOn = false;
    __device__ __forceinline__ double param(const double* __restrict__ params, uint64_t uniform, int j, int64_t N, int64_t i) const
    {
        return param_at(params, uniform, j, N, i);
    }
    __device__ __forceinline__ double param_scalar(const double* __restrict__ params, uint64_t uniform, int j, int64_t N, int64_t i) const
    {
        return param_at_scalar(params, uniform, j, N, i);
    }
    template <int P>
    __device__ __forceinline__ void params(const double* __restrict__ block, uint64_t uniform, int64_t N, int64_t i, double (&p)[P]) const
    {
        params_block<P>(block, uniform, N, i, p);
    }
    __device__ __forceinline__ double state(int, const double* p) const { return *p; }
    __device__ __forceinline__ void put(int, double) const {}
    __device__ __forceinline__ bool has_link(int) const { return false; }
    __device__ __forceinline__ double link(int) const { return 0.0; }
    __device__ __forceinline__ bool last_step() const { return true; }
};

// Thread-private columns of LDS: slot s of this thread at col[s * kCacheStride].  Every value is written and
// read by the same thread, in program order: no barrier.  WARM: not the first model step of the launch, so
// the slots hold what the previous step left (the launch runs its first step with the cold variant).  What comes out of a slot is the double that went in -- the
// same bits as the row in HBM, which is still written every step.
// STRIDE: threads per workgroup (the fused group kernel: kBlock; the whole-graph kernel: one wavefront).
template <bool WARM, int STRIDE = kBlock>
struct LdsCache {
    static constexpr bool kOn = true;
    static constexpr int kCacheStride = STRIDE;
    double* col;
    OpCache c;
    bool last;
    __device__ __forceinline__ double param(const double* __restrict__ params, uint64_t uniform, int j, int64_t N, int64_t i) const
    {
        const bool u = j < 64 && ((uniform >> (j & 63)) & 1ull) != 0;
        if (u || c.param_slot < 0) return param_at_scalar(params, uniform, j, N, i);  // a uniform row is one scalar load
        double* slot = col + (size_t)(c.param_slot + j) * kCacheStride;
        if constexpr (WARM) {
            return *slot;
        } else {
            const double v = params[(size_t)j * N + i];
            *slot = v;
            return v;
        }
    }
    __device__ __forceinline__ double param_scalar(const double* __restrict__ params, uint64_t uniform, int j, int64_t N, int64_t i) const
    {
        return param(params, uniform, j, N, i);
    }
    template <int P>
    __device__ __forceinline__ void params(const double* __restrict__ block, uint64_t uniform, int64_t N, int64_t i, double (&p)[P]) const
    {
        constexpr uint64_t mask = P >= 64 ? ~0ull : ((1ull << P) - 1ull);
        if ((uniform & mask) == mask) {   // nothing of this block lives in a slot
            params_block<P>(block, uniform, N, i, p);
            return;
        }
#pragma unroll
        for (int j = 0; j < P; ++j) p[j] = param(block, uniform, j, N, i);
    }
    __device__ __forceinline__ double state(int v, const double* p) const
    {
        if constexpr (WARM) {
            if (c.series_slot >= 0) return col[(size_t)(c.series_slot + v) * kCacheStride];
        }
        return *p;
    }
    __device__ __forceinline__ void put(int v, double x) const
    {
        if (c.series_slot >= 0) col[(size_t)(c.series_slot + v) * kCacheStride] = x;
    }
    __device__ __forceinline__ bool has_link(int k) const
    {
        return c.link_slot[k] >= 0 && (WARM || ((c.link_warm >> k) & 1u) == 0);
    }
    __device__ __forceinline__ double link(int k) const { return col[(size_t)c.link_slot[k] * kCacheStride]; }
    __device__ __forceinline__ bool last_step() const { return last; }
};

// The same interface with an op's parameters and its own latest row in REGISTERS for the whole launch (csrc/group.hip,
// group_seq_kernel: a kernel compiled for a sequence of kinds keeps one of these per op across its step loop).  The bodies are
// inlined into that loop; what they form from the parameters alone -- reciprocals of heat capacities and lifetimes, folded
// coefficients, h/3 -- is loop-invariant to the compiler and leaves the loop, the state needs no trip through LDS, and only what
// OTHER ops read goes through the LDS slots (put() writes both).  prm / st point at arrays in the kernel's frame; every index the
// bodies pass is a literal after inlining, so the arrays are registers.  What comes out is the double that went in: the same bits.
template <bool WARM, int STRIDE = kBlock>
struct RegCache {
    static constexpr bool kOn = true;
    static constexpr int kCacheStride = STRIDE;
    const double* prm;   // [P] this member's parameter values, loaded once per launch
    double* st;          // [V - 1] the op's latest row (its state for the next step)
    double* col;         // this thread's column of LDS slots
    OpCache c;
    bool last;
    __device__ __forceinline__ double param(const double* __restrict__, uint64_t, int j, int64_t, int64_t) const { return prm[j]; }
    __device__ __forceinline__ double param_scalar(const double* __restrict__, uint64_t, int j, int64_t, int64_t) const { return prm[j]; }
    template <int P>
    __device__ __forceinline__ void params(const double* __restrict__, uint64_t, int64_t, int64_t, double (&p)[P]) const
    {
#pragma unroll
        for (int j = 0; j < P; ++j) p[j] = prm[j];
    }
    __device__ __forceinline__ double state(int v, const double*) const { return st[v]; }
    __device__ __forceinline__ void put(int v, double x) const
    {
        st[v] = x;
        if (c.series_slot >= 0) col[(size_t)(c.series_slot + v) * kCacheStride] = x;
    }
    __device__ __forceinline__ bool has_link(int k) const
    {
        return c.link_slot[k] >= 0 && (WARM || ((c.link_warm >> k) & 1u) == 0);
    }
    __device__ __forceinline__ double link(int k) const { return col[(size_t)c.link_slot[k] * kCacheStride]; }
    __device__ __forceinline__ bool last_step() const { return last; }
};

// The NI input rows of member i.  SRC 0: one shared table [NI][T]; 1: per-member scenario of a
// table [S][NI][T]; 2: linked rows (coalesced [T][N] reads) mixed with table rows.  SRC < 2
// compiles to the plain table indexing the kernels had before links existed.
template <int SRC, int NI>
struct MemberInputs {
    const double* base;
    int32_t T;
    int64_t N, i;
    const InputLinks* links;
    __device__ __forceinline__ MemberInputs(const double* table, const int32_t* scen, const InputLinks& links_, int32_t n_times, int64_t N_,
                                            int64_t i_)
        : T(n_times), N(N_), i(i_), links(&links_)
    {
        static_assert(SRC != 2 || NI <= kMaxLinks, "more input rows than InputLinks holds");
        const size_t s = SRC == 0 ? (size_t)0 : (scen ? (size_t)scen[i_] : (size_t)0);
        base = table + s * NI * n_times;
    }
    // SRC 2: the address is formed where the value is wanted (wave-uniform row choice, a few scalar
    // instructions) -- in a fused launch most linked values come out of LDS and need none
    __device__ __forceinline__ double at(int k, int32_t n) const
    {
        if constexpr (SRC == 2) {
            const double* row = links->row[k];
            if (row) return row[(size_t)(links->off[k] + n) * N + i];
        }
        return base[(size_t)k * T + n];
    }
    // the value of the CURRENT model step (the one the fused launch is at): from the producer's LDS slot if kept there
    template <class Cache>
    __device__ __forceinline__ double at(int k, int32_t n, const Cache& cache) const
    {
        if constexpr (Cache::kOn && SRC == 2) {
            if (cache.has_link(k)) return cache.link(k);
        }
        return at(k, n);
    }
};

// The NI rows of one model step, asked for together.  The light bodies request the rows of their first step -- and their
// state -- before they do any arithmetic on their parameters, and in a launch over several steps the rows of step n + 1
// while step n is computed: a component's step is then ONE trip to memory (parameters, state and inputs in flight
// together) where it was two or three (parameters; the quotients formed from them; then state and inputs, behind the
// status byte's store, which may alias anything).  Fused launches of light components are bound by exactly these trips.
template <int NI>
struct StepRows {
    double v[NI];
};
template <int SRC, int NI>
__device__ __forceinline__ StepRows<NI> rows_at(const MemberInputs<SRC, NI>& in, int32_t n)
{
    StepRows<NI> r;
#pragma unroll
    for (int k = 0; k < NI; ++k) r.v[k] = in.at(k, n);
    return r;
}

// The same rows with every address formed once, up front (pointer + stride per row in vector registers):
// for kernels that read their inputs inside long multi-step loops and are short of SCALAR registers
// (OceanCarbon's convolution keeps a 47-entry window of the response table there).
template <int SRC, int NI>
struct MemberInputsEager {
    const double* base;
    int32_t T;
    const double* p[SRC == 2 ? NI : 1];
    size_t stride[SRC == 2 ? NI : 1];
    __device__ __forceinline__ MemberInputsEager(const double* table, const int32_t* scen, const InputLinks& links, int32_t n_times, int64_t N,
                                                 int64_t i)
        : T(n_times)
    {
        static_assert(SRC != 2 || NI <= kMaxLinks, "more input rows than InputLinks holds");
        const size_t s = SRC == 0 ? (size_t)0 : (scen ? (size_t)scen[i] : (size_t)0);
        base = table + s * NI * n_times;
        if constexpr (SRC == 2) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                if (links.row[k]) {
                    p[k] = links.row[k] + (size_t)links.off[k] * N + i;
                    stride[k] = (size_t)N;
                } else {
                    p[k] = base + (size_t)k * n_times;
                    stride[k] = 1;
                }
            }
        }
    }
    __device__ __forceinline__ double at(int k, int32_t n) const
    {
        if constexpr (SRC == 2) return p[k][(size_t)n * stride[k]];
        else return base[(size_t)k * T + n];
    }
};

// kernel<0> / kernel<1> / kernel<2> by where the inputs come from
#define RSCM_LAUNCH_BY_SOURCE(KERNEL, args, grid, block, stream, ...)                                        \
    do {                                                                                                     \
        if ((args).linked) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, stream, __VA_ARGS__);             \
        else if ((args).scen) hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, stream, __VA_ARGS__);          \
        else hipLaunchKernelGGL((KERNEL<0>), grid, block, 0, stream, __VA_ARGS__);                           \
    } while (0)
#endif

// Stand-alone two-layer run over steps [step_begin, step_end).
struct TwoLayerArgs {
    int64_t n_members;       // members this launch covers ...
    int64_t row_stride;      // ... of an ensemble of this many: the stride of the [rows][N] series and [P][N] parameter rows.  Equal
                             // unless the host launches a BLOCK of members (pointers moved to its first member; rscm_gpu.cpp, split runs)
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t n_scen;
    int32_t src_off;         // 0: Exogenous -> F[n]; 1: UpstreamOutput -> F[n+1]
    int32_t lds_forcing;     // 1: forcing slice staged in LDS, 0: read through L2
    const double* params;    // [6][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* forcing;   // [S][T]
    const double* link;      // [T][N] linked forcing (InputLinks, one row) or nullptr
    const int32_t* scen;     // [N] or nullptr
    const int32_t* nsub;     // [T-1] RK4 sub-steps of step n = ceil((b[n+1]-b[n])/h)
    double h;                // RK4 step (reference: 0.1)
    double h_half, h_sixth;  // h / 2.0 and h / 6.0 as the host's IEEE division rounds them (= the device's): uniform
                             // values, but a per-lane division per launch -- per model step in a fused graph launch
    double* ts;              // [T][N]
    double* td;              // [T][N]
    uint8_t* status;         // [N]
    // Fused likelihood (launch_two_layer_loglik): no series rows beyond step_begin are written;
    // observations merged over both variables and sorted by time index.
    int32_t n_obs;
    int32_t normalize;
    int32_t first_is_deep;        // which variable's group comes first in the caller's order
    const int32_t* obs_tidx;      // [n_obs] ascending
    const int32_t* obs_is_deep;   // [n_obs] 0: Surface Temperature, 1: Deep Ocean Temperature
    const double* obs_value;
    const double* obs_sigma;
    double* loglik;               // [N]
};

// Coupled chain CarbonCycle -> CO2ERF -> Sum -> TwoLayer over steps [step_begin, step_end).
struct CoupledArgs {
    int64_t n_members;
    int64_t row_stride;      // as in TwoLayerArgs
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t n_scen;
    int32_t lds_forcing;
    const double* params;     // [10][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* emissions;  // [S][T]
    const int32_t* scen;
    const int32_t* nsub_tl;   // [T-1]
    const int32_t* nsub_cc;   // [T-1]
    double h_tl, h_cc;
    double* ts; double* td; double* conc; double* cum_uptake; double* cum_emis;
    double* erf_co2; double* erf_total;   // each [T][N]
    uint8_t* status;
};

// ClimateUDEB (rscm-magicc) over steps [step_begin, step_end).
constexpr int kUdebBlock = 64;  // one wavefront per workgroup: ~250 VGPRs per lane
constexpr int kUdebNParams = 37;
constexpr int kUdebScalars = 11;
constexpr int kUdebMaxOnChipLayers = 64;   // up to this many ocean layers a member's columns stay in registers + LDS (csrc/udeb.hip)
constexpr int kUdebMaxLdsLayers = 128;     // ... and up to this many with the Thomas sweep's c' array in LDS (a hemisphere per wavefront)
// The geometry table has two homes, and the unrolled sweeps read BOTH by their compiled CAPACITY, not by the layer count (rows past
// the count are zero rows: exact no-ops).  So each home must hold, zero-padded, as many rows as the largest capacity that reads it:
//   UdebArgs::tables (by value, kernarg)  <- every on-chip instance,        capacity <= kUdebArgTableRows
//   rscm_ens::d_udeb_tables (device)      <- the c'-in-LDS instance,        capacity <= kUdebDevTableRows
// udeb_body.hpp static_asserts each instantiation against its home; rscm_gpu.cpp allocates and zero-fills by these constants.
constexpr int kUdebArgTableRows = kUdebMaxOnChipLayers;
constexpr int kUdebDevTableRows = kUdebMaxLdsLayers;

struct UdebArgs {
    int64_t n_members;
    int64_t row_stride;     // as in TwoLayerArgs: the stride of every [..][N] array when a block of members is launched
    int64_t n_total;        // members of the whole ensemble (0: n_members): the kernel variant is chosen by this, so that the member
                            // blocks of one cut run take one kernel
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t n_scen;
    int32_t n_layers, steps_per_year, land_hc, efficacy_apply;  // uniform over the ensemble
    int32_t fast;           // RSCM_MODE_FAST: one refinement term of the row reciprocals instead of two
    const double* params;   // [37][N], ClimateUDEBParameters order (include/rscm_gpu.h)
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* erf;      // [S][T]
    const int32_t* scen;    // [N] or nullptr
    const double* link;     // [T][N] linked forcing or nullptr
    const double* bounds;   // [T+1] (device)
    const int32_t* win_kfull;  // [T] first history entry that enters the cumulative-T window whole
    const double* win_partw;   // [T] weight of entry win_kfull-1 (0: not in the window)
    // rows [NL][6] = {af_top, af_bot, af_diff, 1 - rel_depth, G_nh, G_sh}, NL = n_layers <= 64 (udeb_tables.hpp), zero rows after
    // them: passed BY VALUE so the kernel reads them from the kernarg segment with scalar loads (no VGPRs, no vmcnt)
    double tables[6 * kUdebArgTableRows];
    const double* derived;     // [kDerivedRows][N] member constants: the base LAMCALC solve (launch_udeb_derive; udeb_body.hpp)
    int32_t derived_uniform;   // every parameter row it is formed from is uniform: element 0 serves all members
    const double* tables_dev;  // the same rows in device memory, any NL: the columns-in-HBM kernel (udeb_any_body.hpp); else nullptr
    double* work;              // [NL][N] the Thomas sweep's c' array of that kernel; else nullptr
    double* ocean;          // [2][NL][N] layer temperatures
    double* scal;           // [10][N] upwelling, land, ground, alpha_eff, hemi exchange (x2 hemispheres)
    double* hist;           // [T][N] year-weighted global temperature history
    double* st0; double* st1; double* st2; double* st3;   // Surface Temperature boxes, [T][N] each
    double* heat_uptake; double* ohc; double* sst;        // [T][N]
    uint8_t* status;        // [N]
};

// GhgForcing: scenario table rows built by the host (rscm_gpu.cpp, ghg_tables), [S][kGhgRows][T]
enum GhgRow {
    kGhgCo2 = 0, kGhgLnCo2, kGhgSqrtCo2, kGhgSqrtCh4, kGhgSqrtN2o,
    kGhgCh4P75, kGhgCh4TimesP152,  // CH4^0.75, CH4 * CH4^1.52
    kGhgN2oP75, kGhgN2oP152,       // N2O^0.75, N2O^1.52
    kGhgRows
};

struct GhgArgs {
    int64_t n_members;
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t rows;            // stored rows per series (T)
    int32_t method;          // 0 = IPCCTAR, 1 = OLBL (uniform over the ensemble)
    const double* params;    // [21][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* derived;   // [kDerivedRows][N] member constants (launch_derive); row layout in ghg_body.hpp
    int32_t derived_uniform; // every parameter row they are formed from is uniform: element 0 serves all members (scalar loads)
    const double* tables;    // [S][kGhgRows][T]
    const int32_t* scen;     // [N] or null
    const double* conc;      // [S][3][T] the concentrations themselves (linked launches)
    InputLinks links;        // used when linked != 0: concentrations per member, no tables
    int32_t linked;
    double* erf_co2; double* erf_ch4; double* erf_n2o;  // [T][N] each
    uint8_t* status;
};

// OzoneForcing / AerosolDirect / AerosolIndirect (csrc/pointwise.hip)
struct PointwiseArgs {
    int64_t n_members;
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t rows;            // stored rows per series (T)
    int32_t kind;            // RSCM_KIND_OZONE_FORCING / _AEROSOL_DIRECT / _AEROSOL_INDIRECT / ...
    const double* params;    // [P][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* inputs;    // [S][n_inputs][T]
    const int32_t* scen;     // [N] or null
    InputLinks links;        // used when linked != 0
    int32_t linked;
    int32_t n_inputs_used;   // aggregate kind: 1 + the highest contributor row that is linked or was given a series (the rest is NaN)
    double* out;             // [n_outputs][rows][N]
    uint8_t* status;
};

// CH4Chemistry / N2OChemistry (csrc/chem.hip)
struct ChemArgs {
    int64_t n_members;
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t kind;            // RSCM_KIND_CH4_CHEMISTRY / RSCM_KIND_N2O_CHEMISTRY
    const double* params;    // [P][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* inputs;    // [S][n_inputs][T]
    const int32_t* scen;     // [N] or null
    InputLinks links;        // used when linked != 0
    int32_t linked;
    const double* bounds;    // [T+1]
    double* conc;            // [T][N] state, row 0 = initial value
    double* lifetime;        // [T][N]
    uint8_t* status;
};

// CO2Budget / TerrestrialCarbon (csrc/carbon.hip)
struct CarbonArgs {
    int64_t n_members;
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t kind;            // RSCM_KIND_CO2_BUDGET / RSCM_KIND_TERRESTRIAL_CARBON
    const double* params;    // [P][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* derived;   // TerrestrialCarbon: [kDerivedRows][N] member constants (launch_derive); row layout in carbon_body.hpp
    int32_t derived_uniform; // every parameter row they are formed from is uniform: element 0 serves all members
    const double* inputs;    // [S][n_inputs][T]
    const int32_t* scen;     // [N] or null
    InputLinks links;        // used when linked != 0
    int32_t linked;
    const double* bounds;    // [T+1]
    const int32_t* nsub;     // CarbonCycle: [T-1] RK4 sub-steps per model step
    double h;                // CarbonCycle: RK4 step
    double h_half, h_sixth;  // h / 2.0, h / 6.0 (see TwoLayerArgs)
    int32_t rows;            // stored rows per series (T, or the window length)
    double* series;          // [n_states + n_outputs][rows][N], states first
    uint8_t* status;
};

// OceanCarbon (csrc/ocean.hip): model steps per pass over the flux history
template <bool FUSED>
constexpr int kOceanTileYears = FUSED ? 4 : 3;
constexpr int kOceanSplitYears = 2;  // years of a tile that is split over one-step launches

// OceanCarbon, RSCM_MODE_FAST: the far part of the impulse response (lags >= near, all in the late
// exponential-sum regime) as a sum of decaying modes  r(lag) ~ sum_q c[q] d[q]^(lag - near)  fitted by the
// host (rscm_gpu.cpp, ocean_fit_modes).  Each mode carries one running sum per member:
//   S_q(m) = d[q] S_q(m-1) + f(m - near) - e[q] f(m - H),   e[q] = d[q]^(H - near)
// so a sub-step costs O(modes + near) instead of O(H) multiply-adds.  Modes are sorted by rate; only the
// first n_exit of them still have a non-negligible weight when a pulse leaves the H-month window.
constexpr int kOceanModes = 21;   // constant + five rates + their fifteen pairwise sums (six-term late forms)
struct OceanModes {
    double d[kOceanModes], c[kOceanModes], e[kOceanModes];
    int32_t n_modes, n_exit;
};

// OceanCarbon (csrc/ocean.hip)
struct OceanArgs {
    int64_t n_members;
    int32_t n_times;
    int32_t step_begin, step_end;
    int32_t steps;           // sub-steps per model step (12)
    int32_t fused;           // RSCM_MODE_FAST: fused multiply-add in the convolution
    int64_t max_hist;        // max_history_months
    const double* params;    // [24][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* inputs;    // [S][2][T]: CO2, SST anomaly
    const int32_t* scen;     // [N] or null
    InputLinks links;        // used when linked != 0
    int32_t linked;
    const double* bounds;    // [T+1]
    const double* irf;       // [max(max_hist, 1)] scaled impulse response at lag k/12 yr
    double* hist;            // [hist_rows][N] flux history, ppm/month: a ring, pulse j in row j mod hist_rows
    int32_t hist_rows;       // min((T-1)*steps, max_hist + ring slack): the whole run's pulses, or the convolution window
    double* partial;         // [(tile years - 1) * steps][N] running sums parked between the launches of a split tile
    int32_t part;            // -1: whole tiles; p >= 0: year p of a tile split over one-step launches
    int32_t rows;            // stored rows per series (T, or the window length)
    int32_t recur;           // RSCM_MODE_FAST with fitted modes: the O(T) recurrence kernel (near = 60 or 120 lags explicit)
    int32_t near;
    int32_t rebuild;         // the mode sums do not stand at step_begin: re-form them from the flux history first
    double* mode_state;      // [kOceanModes][N] the running sums S_q between launches
    const double* mode_table; // [3][kOceanModes] on the device: d_q, c_q, e_q (read with scalar loads)
    OceanModes modes;        // the same by value (n_exit; the constants for host-side use)
    double* series;          // [3][rows][N]: pCO2, cumulative uptake, flux
    uint8_t* status;
};

// HalocarbonChemistry (csrc/halocarbon.hip)
struct HaloArgs {
    int64_t n_members;
    int32_t n_times;
    int32_t step_begin, step_end;
    const double* params;     // [293][N]
    uint64_t uniform_rows;   // bit j: parameter row j (< 64) holds one value for all members (param_at)
    const double* emissions;  // [S][41][T]
    const int32_t* scen;      // [N] or null
    const double* bounds;     // [T+1]
    int32_t rows;             // stored rows per series (T, or the window length)
    double* series;           // [41 + 4][rows][N]: concentrations, then total / F-gas / Montreal forcing, EESC
    uint8_t* status;
};

// device stretch-move sampler (csrc/sampler.hip); pos is [D][W], the per-half buffers [.][W/2]
struct SamplerArgs {
    int32_t n_walkers, n_dims, n_params;
    int32_t n_groups;    // independent ensembles of n_walkers / n_groups walkers each
    int32_t half;        // 0 / 1: the half (of every group) being updated
    int32_t k_offset;    // this rank's block of the half: half-walker indices [k_offset, k_offset + n_local)
    int32_t n_local;     // = members of the evaluating ensemble (n_walkers / 2 when not sharded)
    int32_t iteration;
    int32_t identity;    // score the walkers where they stand (initialisation)
    uint64_t seed;
    double stretch_a;
    const int32_t* param_rows;   // [D] evaluator parameter row of each sampled dimension
    const double* base_params;   // [P] values of the rows that are not sampled
    const int32_t* prior_kind;   // [D] 0 = Uniform(a, b), 1 = Normal(mean a, std b), 2 = LogNormal(mu a, sigma b)
    const double* prior_a;
    const double* prior_b;
    const double* prior_lo;      // [D] truncation (Bound), -inf / +inf when there is none
    const double* prior_hi;
    double* pos;         // [D][W]
    double* logp;        // [W]
    double* proposal;    // [D][n_local]
    double* z;           // [n_local]
    double* lp;          // [n_local] log prior of the proposals
    const double* loglik;  // [n_local] written by the evaluator's fused run+likelihood launch
    double* eval_params;   // [P][n_local] the evaluator's parameter block
    double* const* param_ptr;  // graph evaluator: [D] parameter row of each sampled dimension in its owning ensemble's block ([n_local] each); else null
    double* exchange;      // pack: [D + 1][n_local] out; unpack: [n_ranks][D + 1][n_local] in (positions, then log prob)
    int32_t n_ranks;
    int64_t* n_accepted;   // [W]
    int64_t* n_proposed;   // [W]
};

struct LoglikArgs {
    int64_t n_members;
    int32_t n_obs;
    int32_t normalize;
    const double* const* obs_series;  // [n_obs] device pointers to the [N] row of (var, tidx)
    const int32_t* obs_group;         // [n_obs] variable id, for the per-variable partial sums
    const double* obs_value;
    const double* obs_sigma;
    double* out;                      // [N]
};

// One component of a fused lock-step launch (csrc/group.hip): the arguments its own kernel would get;
// the step range comes with the launch.  The table of ops lives in device memory and is read with
// scalar loads.
struct GroupOp {
    int32_t kind;      // RSCM_KIND_* of a kind the group kernel knows, -1: not fusable
    int32_t variant;   // two-layer: arithmetic mode; GhgForcing: method
    union Args {
        TwoLayerArgs tl;
        GhgArgs ghg;
        PointwiseArgs pw;
        ChemArgs chem;
        CarbonArgs carbon;
        Args() : pw() {}
    } u;
    OpCache cache;     // LDS slots of a multi-step launch (all -1 otherwise)
    // One-step launches (csrc/group.hip): the op runs model step `step + step_off` of the launch -- 1 for the ops of the NEXT step's
    // first segment when the scheduler merges them with this step's last segment (csrc/lockstep.cpp, MERGED schedule).
    int32_t step_off;
    GroupOp() : kind(-1), variant(0), u(), cache(), step_off(0) {}
};
constexpr int kMaxGroupOps = 16;
// up to this many ops travel by value in the kernel-argument segment (4 KiB in all) instead of a device table: the eight light
// components of the MAGICC graph's first segment, or that segment merged with the three of the previous step's last one (eleven)
constexpr int kGroupTableOps = 12;
struct GroupTable {
    GroupOp ops[kGroupTableOps];
};
static_assert(sizeof(GroupOp) == 296, "GroupOp grew: twelve of them must fit the kernel-argument segment");
static_assert(sizeof(GroupTable) <= 3840, "the by-value op table must fit the kernel-argument segment beside the other arguments");
// all_small: every op is one of the kinds group_kind_is_small accepts (the low-register variant of the kernel)
bool group_kind_is_small(int32_t kind);
// cache_slots > 0 (all_small only): the ops carry LDS slots (OpCache), cache_slots doubles per thread in all.
// Exactly one of d_ops (device table) and table (host, passed by value, n_ops <= kGroupTableOps) is given.
hipError_t launch_group(const GroupOp* d_ops, const GroupTable* table, int32_t n_ops, int64_t n_members, int32_t step_begin, int32_t step_end,
                        bool all_small, int32_t cache_slots, hipStream_t s);

// One model step of a by-value table whose first n_first ops and next n_second ops have no edge between them (two wavefronts per 64
// members run them at the same time), the rest after a workgroup barrier (csrc/group.hip, group_split_kernel).
// own_kernel: use the kernel compiled for this sequence of kinds and this cut where one exists (group_split_seq_available), else the interpreter
hipError_t launch_group_split(const GroupTable& table, int32_t n_first, int32_t n_second, int32_t n_ops, int64_t n_members, int32_t step, bool all_small,
                              hipStream_t s, bool own_kernel = true);
bool group_split_seq_available(const GroupTable& table, int32_t n_first, int32_t n_second, int32_t n_ops);

// A multi-step launch of a light graph whose sequence of kinds has a kernel of its own (csrc/group.hip: the op table
// by value, the kinds compile-time): true if one was launched (*status: its launch status), false if the sequence has none.
bool group_seq_available(const int32_t* kinds, int32_t n_ops);
bool launch_group_seq(const GroupTable& table, int32_t n_ops, int64_t n_members, int32_t step_begin, int32_t step_end, int32_t cache_slots,
                      hipStream_t s, hipError_t* status);

hipError_t launch_two_layer(const TwoLayerArgs& a, int mode, hipStream_t s);
hipError_t launch_two_layer_loglik(const TwoLayerArgs& a, int mode, hipStream_t s);
hipError_t launch_coupled(const CoupledArgs& a, int mode, hipStream_t s);
hipError_t launch_udeb(const UdebArgs& a, hipStream_t s);
bool udeb_layers_unrolled(int32_t n_layers);  // the layer counts whose columns stay on chip (2 .. kUdebMaxOnChipLayers)
bool udeb_layers_fixed(int32_t n_layers);     // ... with the count compiled into the instance (20 / 30 / 40 / 50)
void set_udeb_variant(int variant);            // which ClimateUDEB kernel the calling thread's launches take (udeb.hip; -1: by size)
hipError_t launch_ghg(const GhgArgs& a, hipStream_t s);
// Member constants: what a light component's body formed from its parameters alone at the top of EVERY one-step launch (GhgForcing:
// four powers, three logarithms, two square roots of the pre-industrial concentrations; TerrestrialCarbon: its four turnover times,
// seven divisions -- parameters/terrestrial_carbon.rs:103-168) is formed once per parameter set by these small kernels, by the same
// device functions, and stored beside the parameter block: out [kDerivedRows][N].  Which parameter rows they read: *_derive_sources.
constexpr int kDerivedRows = 8;
hipError_t launch_ghg_derive(const double* params, uint64_t uniform_rows, int32_t method, int64_t n_members, double* out, hipStream_t s);
hipError_t launch_terrestrial_derive(const double* params, uint64_t uniform_rows, int64_t n_members, double* out, hipStream_t s);
hipError_t launch_udeb_derive(const double* params, uint64_t uniform_rows, int64_t n_members, double* out, hipStream_t s);
uint64_t udeb_derive_sources();
uint64_t ghg_derive_sources(int32_t method);      // bit j: parameter row j enters the member constants
uint64_t terrestrial_derive_sources();
hipError_t launch_pointwise(const PointwiseArgs& a, hipStream_t s);
hipError_t launch_chem(const ChemArgs& a, hipStream_t s);
hipError_t launch_carbon(const CarbonArgs& a, int mode, hipStream_t s);   // mode: CarbonCycle only (the other two kinds have one arithmetic)
hipError_t launch_ocean(const OceanArgs& a, hipStream_t s);
hipError_t launch_halocarbon(const HaloArgs& a, hipStream_t s);
hipError_t launch_sampler_propose(const SamplerArgs& a, hipStream_t s);
hipError_t launch_sampler_accept(const SamplerArgs& a, hipStream_t s);
// sharded sampler: this rank's block of the updated half into a.exchange / every rank's block out of it
hipError_t launch_sampler_pack(const SamplerArgs& a, hipStream_t s);
hipError_t launch_sampler_unpack(const SamplerArgs& a, hipStream_t s);
hipError_t launch_loglik(const LoglikArgs& a, hipStream_t s);
hipError_t launch_fill(double* p, int64_t n, double v, hipStream_t s);
hipError_t launch_broadcast_row(double* row, int64_t n, const double* src, int64_t n_src,
                                hipStream_t s);
// windowed series buffers [n_vars][R][N] (RSCM_FLAG_WINDOWED): move rows [shift, shift+keep) to the
// front, fill rows [row_begin, R), copy one row of chosen variables into / out of a row store
hipError_t launch_slide_rows(double* buf, int64_t N, int32_t R, int32_t n_vars, int32_t shift, int32_t keep, hipStream_t s);
hipError_t launch_fill_rows(double* buf, int64_t N, int32_t R, int32_t n_vars, int32_t row_begin, double value, hipStream_t s);
hipError_t launch_gather_rows(const double* src, int64_t N, int32_t src_rows, int32_t src_row, const int32_t* vars, int32_t n_out,
                              double* dst, int32_t dst_rows, int32_t dst_row, hipStream_t s);
hipError_t launch_scatter_rows(double* src, int64_t N, int32_t src_rows, int32_t src_row, const int32_t* vars, int32_t n_out,
                               const double* dst, int32_t dst_rows, int32_t dst_row, hipStream_t s);
// The window upkeep of a lock-step graph (slides, NaN fills, output rows) batched: one launch over a table of up to
// kMaxWindowOps descriptors instead of one small launch per ensemble and event.  A descriptor is a row move (rows [shift, shift +
// keep) of every variable to the front; shift >= keep: source and destination rows are disjoint), a row copy into the output store,
// or a fill of rows [fill_from, R).  Moves and copies only read what neither of them writes and go in one launch; fills (which
// overwrite the rows the moves read) in a second one.
constexpr int kMaxWindowOps = 40;
struct WindowOp {
    double* buf;          // [n_vars][R][N]
    const int32_t* vars;  // copy: variable ids kept in the output store (null: every variable, in order)
    double* dst;          // copy: [n_out][dst_rows][N]
    int64_t N;
    int32_t R, n_vars;
    int32_t kind;         // 0: move, 1: copy, 2: fill
    int32_t shift, keep;  // move
    int32_t src_row, n_out, dst_rows, dst_row;  // copy
    int32_t fill_from;    // fill (with NaN)
};
struct WindowBatch {
    WindowOp ops[kMaxWindowOps];
};
hipError_t launch_window_batch(const WindowBatch& batch, int32_t n_ops, hipStream_t s);
// partial[4*n_blocks] then reduced into out[4] = {count_finite, sum, min, max}
hipError_t launch_summary(const double* row, int64_t n, double* partial, int32_t n_blocks,
                          double* out, hipStream_t s);
int32_t summary_blocks(int64_t n);
hipError_t launch_summary_rows(const double* rows, int64_t n, int32_t n_rows, double* partial, int32_t n_blocks,
                               double* out, hipStream_t s);
// per row of rows[n_rows][N]: out[r] = {count of non-NaN, quantile q[0], ..., q[n_q-1]} (numpy nanquantile, linear)
hipError_t launch_quantile_rows(const double* rows, int64_t N, int32_t n_rows, const double* d_q, int32_t n_q, double* d_out,
                                hipStream_t s);
hipError_t launch_lhs(double* params, int32_t n_params, int64_t n_local, uint64_t seed,
                      const double* low, const double* high, int64_t member_offset,
                      int64_t n_total, hipStream_t s);
// out[i] = a[i] / b[i] through (1) the compiler's IEEE division and (2) the hoisted-reciprocal
// path used by the kernels, for the parity test of the latter.
hipError_t launch_divtest(const double* num, const double* den, double* out_ref,
                          double* out_fast, uint8_t* used_fast, int64_t n, hipStream_t s);

}  // namespace rscm
