// Two-layer energy-balance ensemble kernel for gfx950 (MI355X).
//
// One thread per ensemble member; the model's whole time loop, the RK4 sub-steps and the
// TwoLayer right-hand side are fused in that thread, so per member-year the only HBM traffic is
// the two 8-byte state stores (16 B) that the reference's stepper also produces
// (crates/rscm-core/src/model/runtime.rs:480 writes outputs at index n+1).
//
// What it replaces, per step n (reference file:line):
//   Model::step_model_component      crates/rscm-core/src/model/runtime.rs:368-497
//   TwoLayer::solve                  crates/rscm-two-layer/src/component.rs:223-251
//   IVPBuilder::to_rk4 + Rk4 (10 sub-steps of h = 0.1 for an annual step)
//                                    crates/rscm-core/src/ivp/mod.rs:245-253
//   TwoLayer::calculate_dy_dt        crates/rscm-two-layer/src/component.rs:159-189
// The third integrand (heat content, component.rs:186-187) is integrated from 0 and dropped by
// the reference (:236,245-248); it cannot influence Ts/Td and is not computed here.
//
// Layout: params [6][N], state series [T][N] (member fastest => a wavefront stores 512
// contiguous bytes per variable per year), shared forcing [S][T] staged once per workgroup in
// LDS (751 x 8 B = 6 KB per scenario) and read as a broadcast (one scenario) or a per-lane
// gather (scenario_of_member).  No inter-workgroup communication, so the blockIdx -> XCD
// round-robin needs no remap: every workgroup streams its own column block and re-reads only
// the (L2-resident) forcing.
//
// The year loop issues no vector-memory LOAD: forcing comes from LDS (or is prefetched one year
// ahead on the L2 path) and the sub-step counts through the scalar cache, so no s_waitcnt vmcnt
// ever waits behind the previous year's stores -- that wait was 31 % of wave time at
// 1.5 waves/SIMD (profiles/r1_exact_1e5_v1_pmc.txt).
//
// EXACT mode, speculative year: the RK4 sub-steps of a year run branch-free with
//   * n/C as q = n*r; rem = fma(-C,q,n); fma(rem,r,q)   (rk4_device.hpp: identical to IEEE
//     division when the numerator's biased exponent is in [512,1535] and C's in [895,1151]),
//   * k1 + k2*2 as fma(k2, 2, k1)                        (identical while 2*k2 cannot overflow),
// while one integer max3 per right-hand side accumulates whether every numerator stayed inside
// the window.  If any did not (zeros in the first year, a member overflowing towards inf, ...)
// the year is recomputed for those lanes from the saved state with the compiler's full IEEE
// division and unfused doubling.  Either way the stored bits equal the reference's arithmetic.
#include "two_layer_body.hpp"

namespace rscm {

namespace {

template <int MODE, bool LDS, bool STORE>
__global__ __launch_bounds__(kBlock) void two_layer_kernel(TwoLayerArgs a)
{
    extern __shared__ double lds_forcing[];
    if constexpr (LDS) {
        const int32_t len = a.step_end - a.step_begin;
        const int32_t total = a.n_scen * len;
        for (int32_t idx = threadIdx.x; idx < total; idx += kBlock) {
            const int32_t s = idx / len, k = idx - s * len;
            lds_forcing[idx] = a.forcing[(size_t)s * a.n_times + a.step_begin + a.src_off + k];
        }
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n_members) return;
    tl::two_layer_body<MODE, LDS, STORE>(a, lds_forcing, i, a.step_begin, a.step_end);
}

// ---------------------------------------------------------------------------------------------------------------------------
// One persistent launch with a dependency-ordered work queue.
//
// 1e5 members are 1563 wavefront-sized blocks on 1024 SIMDs: as one launch the SIMDs that got two blocks take twice as long as
// those that got one (2.72 ms per 750 years; the chip's full rate would be 2.11).  Cutting the run into member blocks x chunks of
// steps on two streams lets the hardware dispatcher even the load out at chunk granularity (2.30 ms), but a chunk is a kernel:
// a block's next chunk cannot start before EVERY block of its half has finished the previous one.  Here the unit of work is a
// task = (64-member block, chunk of model steps) and the dependency is per block.  A fixed number of wavefronts (a few per
// SIMD) stay resident; each claims the next task with one atomic add, in chunk-major order (all blocks' chunk 0, then all
// blocks' chunk 1, ...), waits -- almost never: its predecessor was claimed a whole round of the chip earlier -- until the same
// block's previous chunk has been published, resumes from the two state values that chunk handed over, steps the chunk with the
// SAME body on the same operands (the same bits), stores the rows as always, hands its final state over and publishes.  The
// forcing of the whole launch is staged in LDS once per workgroup.
//
// Progress: tasks are claimed in increasing order and a task only ever waits for a LOWER-numbered one, which has been claimed by a
// wavefront that is resident and running (it was running when it claimed) -- the lowest unfinished task never waits, so the queue
// drains whatever the placement of the workgroups; a workgroup that starts late finds fewer tasks or none.  Every wait is
// bounded all the same (spin_limit polls, then the error flag and out), and every wavefront leaves when the counter passes the
// last task.
//
// Visibility across the 8 XCDs (their L2s are not coherent with each other): the two hand-over values and the flag are all
// device-scope (agent) relaxed atomic accesses -- stores that write through the L2, loads that do not trust its lines -- and the
// flag is stored after the wavefront has waited for its hand-over stores (s_waitcnt vmcnt(0)), read before the hand-over loads
// are issued (the branch on its value).  Nothing else is shared: the series rows are not read back inside the launch.  (The
// by-the-book form -- plain hand-over accesses around a RELEASE store / ACQUIRE load of the flag -- costs an L2 write-back and an
// L2 invalidate per task: 28 us per task measured, 3.3 ms per pass at 1e5 members against 2.3 for the cut runs.)
// a double through an agent-scope (device-coherent, L2 write-through / bypass) relaxed atomic access
__device__ __forceinline__ double load_agent(const double* p)
{
    const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)bits);
}
__device__ __forceinline__ void store_agent(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct QueueCache : NoCache {
    const double* hand;   // this member's Ts (Td at hand + stride) from the previous chunk, or nullptr: read the stored row
    size_t stride;
    double* out;          // out[0], out[1]: the state after the last step of this chunk
    bool last;
    __device__ __forceinline__ double state(int k, const double* row) const { return hand ? load_agent(hand + (size_t)k * stride) : *row; }
    __device__ __forceinline__ void put(int k, double v) const { out[k] = v; }
    __device__ __forceinline__ bool last_step() const { return last; }
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void two_layer_queue_kernel(TwoLayerArgs a, TlQueue q)
{
    extern __shared__ double lds_forcing[];
    const int32_t len = a.step_end - a.step_begin;
    for (int32_t idx = threadIdx.x; idx < a.n_scen * len; idx += kBlock) {
        const int32_t s = idx / len, k = idx - s * len;
        lds_forcing[idx] = a.forcing[(size_t)s * a.n_times + a.step_begin + a.src_off + k];
    }
    __syncthreads();   // the only workgroup barrier: from here on every wavefront is on its own
    const int lane = threadIdx.x & 63;
    const int32_t n_tasks = q.n_blocks * q.n_chunks;
    for (;;) {
        int32_t task = 0;
        if (lane == 0) task = __hip_atomic_fetch_add(q.next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= n_tasks) break;
        const int32_t chunk = task / q.n_blocks, block = task - chunk * q.n_blocks;
        if (chunk > 0) {   // the same block's previous chunk must have been published
            int32_t polls = 0;
            bool ok = true;
            while (__hip_atomic_load(q.done + block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunk) {
                __builtin_amdgcn_s_sleep(16);
                ++polls;   // (the error flag lives in host memory: looked at once in a while)
                if (polls > q.spin_limit || ((polls & 255) == 0 && __hip_atomic_load(q.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                    ok = false;
                    break;
                }
            }
            if (!ok) {   // never seen; the way out every wait must have
                if (lane == 0) __hip_atomic_store(q.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        const int64_t i = (int64_t)block * 64 + lane;
        const int32_t b = a.step_begin + chunk * q.chunk;
        const int32_t e = b + q.chunk < a.step_end ? b + q.chunk : a.step_end;
        double fin[2] = {0.0, 0.0};
        if (i < a.n_members) {
            QueueCache cache;
            cache.hand = chunk > 0 ? q.hand + i : nullptr;
            cache.stride = (size_t)a.row_stride;
            cache.out = fin;
            cache.last = e == a.step_end;
            tl::two_layer_body<MODE, true, true, QueueCache>(a, lds_forcing, i, b, e, cache, len, a.step_begin);
            if (e < a.step_end) {
                store_agent(q.hand + i, fin[0]);
                store_agent(q.hand + (size_t)a.row_stride + i, fin[1]);
            }
        }
        if (e < a.step_end) {   // every lane's hand-over stores have been performed at device scope before lane 0 publishes
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(q.done + block, chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

hipError_t launch_two_layer_queue(const TwoLayerArgs& a, const TlQueue& q, int mode, int waves_per_simd, int n_cus, hipStream_t s)
{
    if (a.n_members <= 0 || a.step_end <= a.step_begin) return hipSuccess;
    const size_t lds = (size_t)a.n_scen * (a.step_end - a.step_begin) * sizeof(double);
    void (*kern)(TwoLayerArgs, TlQueue) = mode == 0 ? two_layer_queue_kernel<0> : two_layer_queue_kernel<1>;
    if (lds > (size_t)kMaxStaticLds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // a workgroup is one wavefront per SIMD of a CU; no more workgroups than there are blocks to start with
    int64_t groups = (int64_t)n_cus * waves_per_simd;
    const int64_t useful = (q.n_blocks + 3) / 4;
    if (groups > useful) groups = useful;
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(kBlock), lds, s, a, q);
    return hipGetLastError();
}

template <bool STORE>
static hipError_t launch_impl(const TwoLayerArgs& a, int mode, hipStream_t s)
{
    if (a.n_members <= 0) return hipSuccess;
    if (STORE && a.step_end <= a.step_begin) return hipSuccess;
    const size_t lds = a.lds_forcing ? (size_t)a.n_scen * (a.step_end - a.step_begin) * sizeof(double) : 0;
    const dim3 grid((unsigned)((a.n_members + kBlock - 1) / kBlock));
    void (*kern)(TwoLayerArgs) =
        mode == 0 ? (a.lds_forcing ? two_layer_kernel<0, true, STORE> : two_layer_kernel<0, false, STORE>)
                  : (a.lds_forcing ? two_layer_kernel<1, true, STORE> : two_layer_kernel<1, false, STORE>);
    if (lds > (size_t)kMaxStaticLds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_two_layer(const TwoLayerArgs& a, int mode, hipStream_t s)
{
    return launch_impl<true>(a, mode, s);
}

hipError_t launch_two_layer_loglik(const TwoLayerArgs& a, int mode, hipStream_t s)
{
    return launch_impl<false>(a, mode, s);
}

}  // namespace rscm
