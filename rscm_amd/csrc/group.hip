// Fused lock-step launches for graphs of linked ensembles (gfx950).
//
// Model::step walks the component graph (crates/rscm-core/src/model/runtime.rs:368-527); as linked
// ensembles that is one launch per component and model step.  Most components of a MAGICC-style graph
// are a few dozen operations per member and step -- chemistry, the forcing formulas, aggregates, grid
// transforms -- and their launches are bound by the dependent-kernel boundary (~2 us each) and by the
// tail of a grid that lives for 5-10 us, not by arithmetic or bandwidth.  This kernel runs a whole run
// of consecutive such components of one model step in ONE launch: thread i executes, in graph order,
// the per-member body of every op in the table for member i.
//
// Why that is the same computation: every edge of the graph is per member -- a linked input of member
// i is row n or n+1 of the SAME member i of the producer (rscm_ens_link_input) -- so thread i reads
// only what thread i wrote earlier in this launch (program order, same address, same work-item) or
// what earlier launches wrote.  The bodies are the very functions the stand-alone kernels call
// (*_body.hpp), instantiated with the same template arguments, so a fused step carries the bits of the
// unfused one (tests/test_gpu_links.py, tests/test_gpu_group.py).
//
// The op table (kind + the argument struct each kernel would have received) sits in device memory; its
// address is wave-uniform and the table is read-only for the launch, so the fields arrive through the
// scalar cache.  Heavy components -- ClimateUDEB (two 50-layer columns in registers and LDS),
// OceanCarbon (history convolution), HalocarbonChemistry (species-parallel grid) -- keep their own
// launches; rscm_gpu.cpp cuts the step's component list into segments accordingly.
#include <utility>

#include "group_body.hpp"

namespace rscm {

namespace {

// CACHED (a graph of light components only, stepped many model steps in one launch): between the steps every op
// keeps its varying parameter rows, the latest row of its series and thereby what its consumers read in
// thread-private LDS slots (OpCache, assigned by rscm_gpu.cpp) -- in steady state the launch reads nothing back
// from HBM and only streams the series out, like a kernel written for the graph would.
template <bool FULL, bool CACHED, class Ops>
__device__ __forceinline__ void run_graph(const Ops& ops, int32_t n_ops, int64_t n_members, int32_t step_begin, int32_t step_end, double* lds_slots)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_members) return;
    // Several model steps in one launch when the table is the WHOLE graph (no heavy component in between):
    // step after step, component after component, as Model::run does -- the waves never drain between steps.
    if constexpr (CACHED) {
        // the first step fills the slots (cold: parameters and states come from HBM), the others live on them
        for (int32_t k = 0; k < n_ops; ++k) {
            const GroupOp& op = ops[k];
            run_op<FULL>(op, i, step_begin, step_begin + 1, LdsCache<false>{lds_slots + threadIdx.x, op.cache, step_begin + 1 == step_end});
        }
        for (int32_t b = step_begin + 1; b < step_end; ++b) {
            for (int32_t k = 0; k < n_ops; ++k) {
                const GroupOp& op = ops[k];
                run_op<FULL>(op, i, b, b + 1, LdsCache<true>{lds_slots + threadIdx.x, op.cache, b + 1 == step_end});
            }
        }
    } else {
        for (int32_t b = step_begin; b < step_end; ++b) {
            for (int32_t k = 0; k < n_ops; ++k) {
                const int32_t at = b + ops[k].step_off;   // (a merged one-step launch: the next step's first segment rides along)
                run_op<FULL>(ops[k], i, at, at + 1, NoCache());
            }
        }
    }
}

// The op table in device memory (any number of ops; rscm_gpu.cpp uploads what changed since the last launch) ...
template <bool FULL, bool CACHED>
__global__ __launch_bounds__(kBlock) void group_kernel(const GroupOp* __restrict__ ops, int32_t n_ops, int64_t n_members, int32_t step_begin,
                                                       int32_t step_end)
{
    extern __shared__ double lds_slots[];
    run_graph<FULL, CACHED>(ops, n_ops, n_members, step_begin, step_end, lds_slots);
}

// ... or, up to kGroupTableOps ops, by value in the kernel-argument segment: nothing to upload when a window slide
// or a new link changes an op's pointers (the windowed MAGICC graph re-sent 2.7 ops per model step), and the
// fields still arrive through scalar loads.
template <bool FULL, bool CACHED>
__global__ __launch_bounds__(kBlock) void group_kernel_args(const GroupTable table, int32_t n_ops, int64_t n_members, int32_t step_begin,
                                                            int32_t step_end)
{
    extern __shared__ double lds_slots[];
    // Every op starts by reading its fields out of this table through the scalar cache, and a step's ops run one after the other:
    // n_ops first-touch trips to L2 in a row, each exposed (all wavefronts of a one-step launch start together).  One dword of every
    // 64-byte line the ops in use cover, requested here together and awaited once, leaves the lines in the scalar cache for them.
    {
        const uint32_t* words = reinterpret_cast<const uint32_t*>(&table);
        const int32_t n_lines = (int32_t)(((size_t)n_ops * sizeof(GroupOp) + 63) / 64);
        uint32_t touched = 0;
        for (int32_t l = 0; l < n_lines; ++l) touched |= words[(size_t)l * 16];
        asm volatile("" ::"s"(touched));
    }
    run_graph<FULL, CACHED>(table.ops, n_ops, n_members, step_begin, step_end, lds_slots);
}


// ---- two independent chains of a step on two wavefronts ---------------------------------------------------------
// A one-step launch of light components is a chain of dependent memory trips, one op after the other: at
// 125 000 members the eight ops of the MAGICC graph's first segment take 33 us of which ~7 us are arithmetic.  Many of those ops do not
// depend on each other -- the aerosol forcings and the chemistry -> greenhouse-gas forcing branch only meet in the Sum of the forcings.
// The host (csrc/lockstep.cpp, plan_split) cuts the segment into two sets of ops with no edge between them plus a tail that may read
// both; here a workgroup of TWO wavefronts serves 64 members: wavefront 0 runs the first set for them, wavefront 1 the second, at the
// same time; after a workgroup barrier (release / acquire at workgroup scope: both wavefronts sit on one CU and share its L1) wavefront 0
// runs the tail.  Every op still executes the same body on the same operands, in an order the graph's edges allow: the same bits
// (tests/test_gpu_group.py).  Twice the wavefronts of the plain launch must be resident (3908 at 125 000 members: they are, at <= 128
// registers); no lane leaves before the barrier.
template <bool FULL, class Ops>
__device__ __forceinline__ void split_range(const Ops& ops, int32_t begin, int32_t end, int64_t i, int32_t step, bool live)
{
    if (!live) return;
    for (int32_t k = begin; k < end; ++k) {
        const int32_t at = step + ops[k].step_off;
        run_op<FULL>(ops[k], i, at, at + 1, NoCache());
    }
}

template <bool FULL>
__global__ __launch_bounds__(128) void group_split_kernel(const GroupTable table, int32_t n_first, int32_t n_second, int32_t n_ops, int64_t n_members,
                                                          int32_t step)
{
    {   // the table's lines up front, as in group_kernel_args
        const uint32_t* words = reinterpret_cast<const uint32_t*>(&table);
        const int32_t n_lines = (int32_t)(((size_t)n_ops * sizeof(GroupOp) + 63) / 64);
        uint32_t touched = 0;
        for (int32_t l = 0; l < n_lines; ++l) touched |= words[(size_t)l * 16];
        asm volatile("" ::"s"(touched));
    }
    const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(threadIdx.x >> 6));   // wave-uniform, and said so: the op index stays scalar
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const bool live = i < n_members;
    // ONE call site of the bodies (a second one takes the uses of the by-value table past the point where the compiler stops treating its
    // stack copy as read-only and keeps the copy: 2.4 KB of scratch per lane): phase 0 = this wavefront's set, phase 1 = the tail.
    for (int32_t phase = 0; phase < 2; ++phase) {
        int32_t begin, end;
        if (phase == 0) {
            begin = wave == 0 ? 0 : n_first;
            end = wave == 0 ? n_first : n_first + n_second;
        } else {
            __syncthreads();
            begin = n_first + n_second;
            end = wave == 0 ? n_ops : begin;
        }
        split_range<FULL>(table.ops, begin, end, i, step, live);
    }
}

// ---- the cut launch with the op KINDS fixed at compile time ------------------------------------------------------------
// The interpreter above reaches an op's fields through a dynamic index into the by-value table and a switch on its kind: every field is
// a scalar load of its own in some basic block of the body, awaited where it is used -- 122 scalar-memory instructions per wavefront in
// the MAGICC graph's first segment, two thirds of the wave cycles in s_waitcnt at 0.2 issue utilisation
// (profiles/r6_configs3_share_pmc.txt, profiles/r6_group_latency_counters.txt).  For the op sequences the front end emits most often --
// the MAGICC graph's merged launch -- the kinds and the cut are template arguments: the bodies are called directly, every field of
// every op sits at a constant offset of the kernel-argument segment (invariant loads the compiler batches and hoists), there is no
// switch.  Same bodies, same template arguments, same operands as group_split_kernel: the same bits (tests/test_gpu_group.py).
template <int KIND>
__device__ __forceinline__ void run_kind_once(const GroupOp& op, int64_t i, int32_t at)
{
    if constexpr (KIND == 0) {
        if (op.variant == 0) tl::two_layer_body<0, false, true>(op.u.tl, nullptr, i, at, at + 1, NoCache());
        else tl::two_layer_body<1, false, true>(op.u.tl, nullptr, i, at, at + 1, NoCache());
    } else if constexpr (KIND == 3) {
        if (op.variant == 0) ghg::ghg_body<0, false, true>(op.u.ghg, nullptr, i, at, at + 1);
        else ghg::ghg_body<1, false, true>(op.u.ghg, nullptr, i, at, at + 1);
    }
    else if constexpr (KIND == kKindCh4Chemistry) chem::ch4_body<2>(op.u.chem, i, at, at + 1);
    else if constexpr (KIND == kKindN2oChemistry) chem::n2o_body<2>(op.u.chem, i, at, at + 1);
    else if constexpr (KIND == kKindTerrestrialCarbon) carbon::terrestrial_body<2>(op.u.carbon, i, at, at + 1);
    else if constexpr (KIND == kKindCo2Budget) carbon::co2_budget_body<2>(op.u.carbon, i, at, at + 1, NoCache());
    else if constexpr (KIND == kKindCarbonCycle) {
        if (op.variant == 0) carbon::carbon_cycle_body<0, 2>(op.u.carbon, i, at, at + 1, NoCache());
        else carbon::carbon_cycle_body<1, 2>(op.u.carbon, i, at, at + 1, NoCache());
    }
    else if constexpr (KIND == kKindOzoneForcing || KIND == kKindAerosolDirect) pw::pointwise_body<KIND, 2>(op.u.pw, i, at, at + 1);
    else pw::pointwise_body<KIND, 2>(op.u.pw, i, at, at + 1, NoCache());   // AerosolIndirect, FourBoxOHU, OSPP, CO2ERF, the aggregate
}

template <int... KINDS>
struct OpKinds {
    static constexpr int n = sizeof...(KINDS);
    static constexpr int kinds[sizeof...(KINDS) > 0 ? sizeof...(KINDS) : 1] = {KINDS...};
};

template <class Seq, int BASE, int... IDX>
__device__ __forceinline__ void run_kinds(const GroupTable& table, int64_t i, int32_t step, std::integer_sequence<int, IDX...>)
{
    (run_kind_once<Seq::kinds[IDX]>(table.ops[BASE + IDX], i, step + table.ops[BASE + IDX].step_off), ...);
}

// wavefront 0: the ops of A, wavefront 1: the ops of B, at the same time; a workgroup barrier; wavefront 0: the tail T.
template <class A, class B, class T>
__global__ __launch_bounds__(128) void group_split_seq_kernel(const GroupTable table, int64_t n_members, int32_t step)
{
    const int32_t wave = __builtin_amdgcn_readfirstlane((int32_t)(threadIdx.x >> 6));
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const bool live = i < n_members;
    if (wave == 0) {
        if (live) run_kinds<A, 0>(table, i, step, std::make_integer_sequence<int, A::n>());
    } else {
        if (live) run_kinds<B, A::n>(table, i, step, std::make_integer_sequence<int, B::n>());
    }
    __syncthreads();
    if (wave == 0 && live) run_kinds<T, A::n + B::n>(table, i, step, std::make_integer_sequence<int, T::n>());
}

// The MAGICC graph in topological order, a step's last segment merged with the next step's first (csrc/lockstep.cpp; the cut as
// plan_split makes it): wavefront 0 the temperature's grid transform, TerrestrialCarbon, CO2Budget and CH4; wavefront 1 the aerosol
// forcings, their grid transform and N2O; the tail GhgForcing, OzoneForcing and the Sum of the forcings.
using MagiccA = OpKinds<kKindAggregate, kKindTerrestrialCarbon, kKindCo2Budget, kKindCh4Chemistry>;
using MagiccB = OpKinds<kKindAerosolIndirect, kKindAerosolDirect, kKindAggregate, kKindN2oChemistry>;
using MagiccT = OpKinds<3, kKindOzoneForcing, kKindAggregate>;
// ... and in the reference's breadth-first order (what ModelBuilder.build() steps a graph in; the temperature's grid transform is a
// launch of its own there, in front of OceanCarbon): wavefront 0 the aerosol forcings, their transform and CH4; wavefront 1
// TerrestrialCarbon, CO2Budget and N2O; the tail the Sum of the forcings (which that order places BEFORE GhgForcing: it reads the
// forcing of the step before), GhgForcing and OzoneForcing.
using MagiccRefA = OpKinds<kKindAerosolIndirect, kKindAerosolDirect, kKindAggregate, kKindCh4Chemistry>;
using MagiccRefB = OpKinds<kKindTerrestrialCarbon, kKindCo2Budget, kKindN2oChemistry>;
using MagiccRefT = OpKinds<kKindAggregate, 3, kKindOzoneForcing>;

template <class A, class B, class T>
static bool split_seq_matches(const GroupTable& table, int32_t n_first, int32_t n_second, int32_t n_ops)
{
    if (n_first != A::n || n_second != B::n || n_ops != A::n + B::n + T::n) return false;
    for (int k = 0; k < A::n; ++k)
        if (table.ops[k].kind != A::kinds[k]) return false;
    for (int k = 0; k < B::n; ++k)
        if (table.ops[A::n + k].kind != B::kinds[k]) return false;
    for (int k = 0; k < T::n; ++k)
        if (table.ops[A::n + B::n + k].kind != T::kinds[k]) return false;
    return true;
}

// ---- a kernel per graph: the op KINDS fixed at compile time -------------------------------------------------
// The interpreter above pays, per op and model step, for what it cannot know: the switch on the kind, the op's
// fields re-read through the scalar cache (a dynamic index into the table), the slot records of its LDS cache --
// 444 scalar and ~150 extra vector instructions per wavefront-step on the coupled chain against the fused
// coupled kernel's 52 scalar ones (profiles/r2_group_ops_counters.txt).  For the graphs the front end emits
// most often the sequence of kinds is a template argument: the bodies are called directly, the table travels by
// value in the kernel arguments and every field, pointer and slot number is a kernel-argument load the compiler
// hoists out of the step loop.  Same bodies, same template arguments, same LDS slots: the same bits.
template <int KIND, class Cache>
__device__ __forceinline__ void run_kind(const GroupOp& op, int64_t i, int32_t b, const Cache& cache)
{
    if constexpr (KIND == 0) {
        if (op.variant == 0) tl::two_layer_body<0, false, true>(op.u.tl, nullptr, i, b, b + 1, cache);
        else tl::two_layer_body<1, false, true>(op.u.tl, nullptr, i, b, b + 1, cache);
    } else if constexpr (KIND == kKindCarbonCycle) {
        if (op.variant == 0) carbon::carbon_cycle_body<0, 2>(op.u.carbon, i, b, b + 1, cache);
        else carbon::carbon_cycle_body<1, 2>(op.u.carbon, i, b, b + 1, cache);
    }
    else if constexpr (KIND == kKindCo2Budget) carbon::co2_budget_body<2>(op.u.carbon, i, b, b + 1, cache);
    else pw::pointwise_body<KIND, 2>(op.u.pw, i, b, b + 1, cache);   // CO2ERF, aggregate, AerosolIndirect, FourBoxOHU, OSPP
}

template <int... KINDS>
struct KindSeq {
    static constexpr int n = sizeof...(KINDS);
    static constexpr int kinds[sizeof...(KINDS)] = {KINDS...};
};

// Parameters (P) and stored series (S = V - 1) of the kinds a sequence may hold, and where their blocks live in the op's arguments.
template <int KIND> struct SeqShape { static constexpr int P = pw::Shape<KIND>::P, S = pw::Shape<KIND>::NO; };
template <> struct SeqShape<0> { static constexpr int P = 6, S = 2; };
template <> struct SeqShape<kKindCarbonCycle> { static constexpr int P = 3, S = 3; };

// An op's parameters and latest row, in registers for the whole launch (RegCache, rscm_device.hpp)
template <int KIND>
struct OpRegs {
    double prm[SeqShape<KIND>::P];
    double st[SeqShape<KIND>::S];
    __device__ __forceinline__ void load(const GroupOp& op, int64_t i, int32_t step_begin)
    {
        const double* params;
        uint64_t uniform;
        int64_t N;
        if constexpr (KIND == 0) { params = op.u.tl.params; uniform = op.u.tl.uniform_rows; N = op.u.tl.row_stride; }
        else if constexpr (KIND == kKindCarbonCycle || KIND == kKindCo2Budget) { params = op.u.carbon.params; uniform = op.u.carbon.uniform_rows; N = op.u.carbon.n_members; }
        else { params = op.u.pw.params; uniform = op.u.pw.uniform_rows; N = op.u.pw.n_members; }
#pragma unroll
        for (int j = 0; j < SeqShape<KIND>::P; ++j) prm[j] = param_at(params, uniform, j, N, i);
        // the state rows the bodies ask their cache for (the pointwise kinds have outputs only: nothing is read back)
        const size_t r0 = (size_t)step_begin * N + i;
        if constexpr (KIND == 0) {
            st[0] = op.u.tl.ts[r0];
            st[1] = op.u.tl.td[r0];
        } else if constexpr (KIND == kKindCarbonCycle || KIND == kKindCo2Budget) {
            const size_t vs = (size_t)op.u.carbon.rows * N;
#pragma unroll
            for (int v = 0; v < SeqShape<KIND>::S; ++v) st[v] = op.u.carbon.series[v * vs + r0];
        } else {
#pragma unroll
            for (int v = 0; v < SeqShape<KIND>::S; ++v) st[v] = 0.0;
        }
    }
};

template <int... K> struct RegPack;
template <> struct RegPack<> {};
template <int K0, int... K>
struct RegPack<K0, K...> {
    OpRegs<K0> head;
    RegPack<K...> tail;
};
template <int I, int K0, int... K>
__device__ __forceinline__ auto& pack_get(RegPack<K0, K...>& p)
{
    if constexpr (I == 0) return p.head;
    else return pack_get<I - 1>(p.tail);
}
template <class Seq> struct PackOf;
template <int... K> struct PackOf<KindSeq<K...>> { using type = RegPack<K...>; };

template <class Seq, bool WARM, class Pack, int... IDX>
__device__ __forceinline__ void run_seq_step(const GroupTable& table, Pack& regs, int64_t i, int32_t b, bool last, double* slots,
                                             std::integer_sequence<int, IDX...>)
{
    (run_kind<Seq::kinds[IDX]>(table.ops[IDX], i, b,
                               RegCache<WARM>{pack_get<IDX>(regs).prm, pack_get<IDX>(regs).st, slots, table.ops[IDX].cache, last}), ...);
}

template <class Seq, class Pack, int... IDX>
__device__ __forceinline__ void load_seq_regs(const GroupTable& table, Pack& regs, int64_t i, int32_t step_begin, std::integer_sequence<int, IDX...>)
{
    (pack_get<IDX>(regs).load(table.ops[IDX], i, step_begin), ...);
}

// Parameters and states live in registers from the first step to the last (RegCache): the bodies are inlined into the step loop, so what
// they form from the parameters alone -- reciprocals, folded coefficients, the refined reciprocals of the speculative divisions -- is
// loop-invariant and formed once per launch, like in a kernel written for the graph; the LDS slots carry only what ops read of each other.
template <class Seq>
__global__ __launch_bounds__(kBlock) void group_seq_kernel(const GroupTable table, int64_t n_members, int32_t step_begin, int32_t step_end)
{
    extern __shared__ double lds_slots[];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_members) return;
    double* slots = lds_slots + threadIdx.x;
    const auto idx = std::make_integer_sequence<int, Seq::n>();
    typename PackOf<Seq>::type regs;
    load_seq_regs<Seq>(table, regs, i, step_begin, idx);
    run_seq_step<Seq, false>(table, regs, i, step_begin, step_begin + 1 == step_end, slots, idx);   // cold: feedback links come from memory
    for (int32_t b = step_begin + 1; b < step_end; ++b) {
        // the fields of four ops do not fit the scalar registers at once: an opaque zero offset per step keeps their
        // loads inside the step (a few s_load_dwordx16 from the kernel-argument segment) instead of hoisted and spilled
        int32_t opaque = 0;
        asm volatile("" : "+s"(opaque));
        run_seq_step<Seq, true>((&table)[opaque], regs, i, b, b + 1 == step_end, slots, idx);
    }
}

// the graphs with a kernel of their own: the coupled chain of BASELINE configs[2] / the reference's notebook as four
// linked components (CarbonCycle -> CO2ERF -> Sum -> TwoLayer), and the same with the two-layer model alone behind a Sum
using SeqCoupled = KindSeq<kKindCarbonCycle, kKindCo2Erf, kKindAggregate, 0>;
using SeqForced = KindSeq<kKindAggregate, 0>;

template <class Seq>
static bool seq_matches(const int32_t* kinds, int32_t n)
{
    if (n != Seq::n) return false;
    for (int k = 0; k < n; ++k)
        if (kinds[k] != Seq::kinds[k]) return false;
    return true;
}

}  // namespace

bool group_kind_is_small(int32_t kind)
{
    switch (kind) {
        case 0: case kKindAerosolIndirect: case kKindFourBoxOhu: case kKindOspp: case kKindCo2Erf: case kKindAggregate:
        case kKindCo2Budget: case kKindCarbonCycle:
            return true;
        default: return false;
    }
}

template <class... Args>
static void launch_variant(bool by_value, bool all_small, int32_t cache_slots, dim3 grid, size_t lds, hipStream_t s, const GroupOp* d_ops,
                           const GroupTable* table, Args... rest)
{
    if (by_value) {  // (never with LDS slots: indexing the by-value table for the slot records sends it to scratch)
        if (all_small) hipLaunchKernelGGL((group_kernel_args<false, false>), grid, dim3(kBlock), 0, s, *table, rest...);
        else hipLaunchKernelGGL((group_kernel_args<true, false>), grid, dim3(kBlock), 0, s, *table, rest...);
    } else {
        if (cache_slots > 0) hipLaunchKernelGGL((group_kernel<false, true>), grid, dim3(kBlock), lds, s, d_ops, rest...);
        else if (all_small) hipLaunchKernelGGL((group_kernel<false, false>), grid, dim3(kBlock), 0, s, d_ops, rest...);
        else hipLaunchKernelGGL((group_kernel<true, false>), grid, dim3(kBlock), 0, s, d_ops, rest...);
    }
}

bool group_seq_available(const int32_t* kinds, int32_t n_ops)
{
    return seq_matches<SeqCoupled>(kinds, n_ops) || seq_matches<SeqForced>(kinds, n_ops);
}

bool launch_group_seq(const GroupTable& table, int32_t n_ops, int64_t n_members, int32_t step_begin, int32_t step_end, int32_t cache_slots,
                      hipStream_t s, hipError_t* status)
{
    if (n_ops <= 0 || n_ops > kGroupTableOps || cache_slots <= 0 || n_members <= 0 || step_end <= step_begin) return false;
    int32_t kinds[kGroupTableOps];
    for (int32_t k = 0; k < n_ops; ++k) kinds[k] = table.ops[k].kind;
    const dim3 grid((unsigned)((n_members + kBlock - 1) / kBlock));
    const size_t lds = (size_t)cache_slots * kBlock * sizeof(double);
    if (seq_matches<SeqCoupled>(kinds, n_ops)) hipLaunchKernelGGL(group_seq_kernel<SeqCoupled>, grid, dim3(kBlock), lds, s, table, n_members, step_begin, step_end);
    else if (seq_matches<SeqForced>(kinds, n_ops)) hipLaunchKernelGGL(group_seq_kernel<SeqForced>, grid, dim3(kBlock), lds, s, table, n_members, step_begin, step_end);
    else return false;
    *status = hipGetLastError();
    return true;
}

template <class A, class B, class T>
static bool launch_split_seq(const GroupTable& table, int32_t n_first, int32_t n_second, int32_t n_ops, int64_t n_members, int32_t step, dim3 grid,
                             hipStream_t s)
{
    if (!split_seq_matches<A, B, T>(table, n_first, n_second, n_ops)) return false;
    hipLaunchKernelGGL((group_split_seq_kernel<A, B, T>), grid, dim3(128), 0, s, table, n_members, step);
    return true;
}

bool group_split_seq_available(const GroupTable& table, int32_t n_first, int32_t n_second, int32_t n_ops)
{
    return split_seq_matches<MagiccA, MagiccB, MagiccT>(table, n_first, n_second, n_ops) ||
           split_seq_matches<MagiccRefA, MagiccRefB, MagiccRefT>(table, n_first, n_second, n_ops);
}

hipError_t launch_group_split(const GroupTable& table, int32_t n_first, int32_t n_second, int32_t n_ops, int64_t n_members, int32_t step, bool all_small,
                              hipStream_t s, bool own_kernel)
{
    if (n_first < 1 || n_second < 1 || n_first + n_second > n_ops || n_ops > kGroupTableOps || n_members <= 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n_members + 63) / 64));
    if (own_kernel && (launch_split_seq<MagiccA, MagiccB, MagiccT>(table, n_first, n_second, n_ops, n_members, step, grid, s) ||
                       launch_split_seq<MagiccRefA, MagiccRefB, MagiccRefT>(table, n_first, n_second, n_ops, n_members, step, grid, s)))
        return hipGetLastError();
    if (all_small) hipLaunchKernelGGL(group_split_kernel<false>, grid, dim3(128), 0, s, table, n_first, n_second, n_ops, n_members, step);
    else hipLaunchKernelGGL(group_split_kernel<true>, grid, dim3(128), 0, s, table, n_first, n_second, n_ops, n_members, step);
    return hipGetLastError();
}

hipError_t launch_group(const GroupOp* d_ops, const GroupTable* table, int32_t n_ops, int64_t n_members, int32_t step_begin, int32_t step_end,
                        bool all_small, int32_t cache_slots, hipStream_t s)
{
    if (n_ops <= 0 || n_members <= 0 || step_end <= step_begin) return hipSuccess;
    if (cache_slots > 0 && !all_small) return hipErrorInvalidValue;
    if ((table != nullptr) == (d_ops != nullptr) || (table && (n_ops > kGroupTableOps || cache_slots > 0))) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n_members + kBlock - 1) / kBlock));
    const size_t lds = (size_t)cache_slots * kBlock * sizeof(double);
    launch_variant(table != nullptr, all_small, cache_slots, grid, lds, s, d_ops, table, n_ops, n_members, step_begin, step_end);
    return hipGetLastError();
}

}  // namespace rscm
