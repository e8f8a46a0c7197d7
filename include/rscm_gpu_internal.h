/*
 * rscm_gpu_internal.h -- test and A/B hooks of librscm_gpu.so.  NOT part of the drop-in boundary
 * (include/rscm_gpu.h): nothing here replaces a reference interface, a host integration binds none of it.
 * The library exports these symbols for tests/, scripts/ and bench.py only.
 *
 * Threading: the lock-step switches and counters are per calling thread (thread_local in
 * csrc/lockstep.cpp) -- the boundary's model is one handle per device per thread, so a thread that
 * flips an A/B switch changes its own rscm_ens_run_lockstep calls and nobody else's.
 */
#ifndef RSCM_GPU_INTERNAL_H
#define RSCM_GPU_INTERNAL_H

#include "rscm_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How the calling thread's rscm_ens_run_lockstep calls are cut into launches:
 *   0  one launch per component and step (no fusion);
 *   1  default: consecutive light components fused per step; a graph of light components only in one launch for
 *      all steps, with thread-private LDS slots between the steps;
 *   2  as 1 without the LDS slots;
 *   3  as 1 with every op table sent through device memory instead of the kernel arguments;
 *   4  as 1 without cutting a one-step segment's independent ops over two wavefronts (csrc/group.hip, group_split_kernel): A/B.
 *   5  as 1 without MERGED launches: a step's last fused segment and the next step's first one stay two launches (round 5's launch
 *      plan; A/B, and the yardstick of tests/test_gpu_group.py: merged == unmerged bit for bit);
 *   6  as 1 with every cut launch through the op interpreter (group_split_kernel), also where its sequence of kinds has a kernel of its
 *      own (group_split_seq_kernel: the MAGICC graph's merged launch): A/B.
 *      (Round 3's mode 4 -- ClimateUDEB and OceanCarbon inside the fused launch too, one launch per window chunk with the ocean columns
 *      resident on chip -- existed in round 3: bit-identical to mode 1 and 17 % slower on an MI355X, removed in round 4;
 *      DESIGN.md section 8g, profiles/r3_graph_stamps.json, commit f22e743.)
 * The setting is per calling THREAD: lock-step runs issued from another thread do not see it. */
RSCM_API int rscm_gpu_set_lockstep_fusion(int32_t enabled);
/* Step launches issued by the calling thread's rscm_ens_run_lockstep calls (component kernels + fused
 * groups; HalocarbonChemistry counts as one) and the component steps they carried, since the thread's
 * last call of this function; resets both counters. */
RSCM_API int rscm_gpu_lockstep_stats(int64_t* launches, int64_t* component_steps);
/* How many of the calling thread's fused launches since its last call of this function ran two independent sets of ops on two
 * wavefronts per 64 members (group_split_kernel); resets the counter. */
RSCM_API int rscm_gpu_lockstep_split_launches(int64_t* out);

/* How many of the calling thread's fused launches since its last call of this function carried the segments of TWO model steps (the
 * last segment of step n with the first segment of step n + 1; csrc/lockstep.cpp, MERGED schedule); resets the counter. */
RSCM_API int rscm_gpu_lockstep_merged_launches(int64_t* out);

/* The layout of the calling thread's last one-step fused launch that was cut over two wavefronts: out[0] = ops, out[1] / out[2] = ops
 * of the first / second set (the rest is the tail), then per op in launch order its kind and (step offset | variant << 8).
 * out must hold 3 + 2 * 12 int32. */
RSCM_API int rscm_gpu_lockstep_last_layout(int32_t* out);
/* How many of the calling thread's cut launches went through a kernel compiled for their sequence of kinds (csrc/group.hip,
 * group_split_seq_kernel) since its last call of this function; resets the counter. */
RSCM_API int rscm_gpu_lockstep_own_cut_launches(int64_t* out);

/* Which ClimateUDEB kernel the calling thread's launches take (csrc/udeb.hip): 0 one thread per member, 2 a hemisphere per
 * wavefront; -1 (default): chosen by ensemble size.  The two carry the same bits, at every layer count up to 64 (20 / 30 / 40 / 50
 * with the count compiled in, the others with the count at run time in the next capacity's instance).  3: the columns-in-HBM
 * kernel that serves more than 64 layers, for a count it would not otherwise serve -- the yardstick the runtime-count kernels
 * are held to, bit for bit (not available at 20 / 30 / 40 / 50: its work arrays are not allocated there, the launch fails).
 * The setting is per calling THREAD: launches issued from another thread do not see it. */
RSCM_API int rscm_gpu_set_udeb_variant(int32_t variant);

/* How the calling THREAD's whole-axis runs over more members than the chip holds at one wavefront per SIMD go out: -1 (default) by
 * the environment (RSCM_SPLIT_RUNS) and the sizes -- the two-stream cut where it applies; 0 always one plain launch (the yardstick of
 * the parity tests); 1 the two-stream cut where it applies whatever the environment says.  The same bits either way. */
RSCM_API int rscm_gpu_set_run_plan(int32_t mode);

/* 1 if this library was built with -DRSCM_EXPERIMENTS (`make -C rscm_amd/csrc EXPERIMENTS=1`): the environment-variable experiment
 * knobs of csrc/experiment_env.hpp (RSCM_SPLIT_CHUNK / _CHUNK2 / _FIRST, RSCM_LOCKSTEP_SPLIT, RSCM_UDEB_VARIANT) are compiled in.
 * 0 for the shipped library, whose launch plans read only the two variables documented in rscm_gpu.h ("Environment").
 * __graft_entry__.build() and tests/test_abi_symbols.py refuse an experiments build left in place. */
RSCM_API int rscm_gpu_experiments_build(void);

/* Member-constant ("derive") kernels launched by the calling THREAD since its last call of this function (GhgForcing, TerrestrialCarbon,
 * ClimateUDEB: what their bodies need of the parameters alone, formed once per parameter set); resets the counter.  A handle whose
 * parameter block the caller holds a device pointer to (rscm_ens_params_devptr) is re-derived before every RUN -- once per
 * rscm_ens_run* / rscm_ens_run_lockstep call, not once per model step of it (tests/test_gpu_links.py). */
RSCM_API int rscm_gpu_derive_launches(int64_t* out);

/* Fault injection for the cut runs (rscm_ens_last_run_plan: member blocks x step chunks on two streams): the k-th chunk launch
 * (1-based, counted over both blocks in issue order) of the calling THREAD's next cut run is not issued and reports
 * hipErrorLaunchFailure instead.  The next cut run consumes the hook whether or not it issues as many as k launches (a k beyond the
 * run's launches fails nothing and is gone afterwards); a run that is NOT cut (one plain launch) does not consume it -- it stays
 * armed for the thread's next cut run.  0 turns it off.  What must hold afterwards
 * (tests/test_gpu_parity.py): rscm_ens_run returns RSCM_ERR_DEVICE, the caller's stream has been joined with the helper stream
 * (nothing issued there is still running once the caller's stream is synchronised), the time index has not moved, and a fresh
 * whole run gives the uncut path's bits. */
RSCM_API int rscm_gpu_fail_chunk_launch(int32_t k);

/* OceanCarbon in RSCM_MODE_FAST replaces the O(T^2) history convolution of carbon/ocean.rs:151-190 by an
 * O(T) recurrence: lags below `near_lags` months explicitly, the rest through decaying modes fitted to the
 * scaled impulse response (parameters/ocean_carbon.rs:85-216) by the host.  This runs that fit alone (no
 * GPU): the largest deviation of the fitted response from the tabulated one over the window (negative: the
 * parameters do not allow the recurrence and FAST keeps the tiled convolution), the number of modes, of
 * modes that still weigh when a pulse leaves the window, and the largest amplitude. */
RSCM_API int rscm_gpu_ocean_fit_selftest(int32_t model, double irf_scale, double irf_switch_time,
                                         int64_t max_history_months, double* max_error, int32_t* n_modes,
                                         int32_t* near_lags, int32_t* n_exit, double* max_abs_coefficient);

/* Element-wise num[i]/den[i] on the device through (a) the compiler's IEEE f64 division and
 * (b) the three-instruction hoisted-reciprocal quotient of rk4_device.hpp with no fallback;
 * used_fast[i] = 1 where both operands are inside the windows in which the kernels trust (b).
 * The parity tests require out_ref == out_fast bit for bit wherever used_fast is 1. */
RSCM_API int rscm_gpu_selftest_div(int32_t device_id, int64_t n, const double* num, const double* den,
                                   double* out_ref, double* out_fast, uint8_t* used_fast);

#ifdef __cplusplus
}
#endif
#endif /* RSCM_GPU_INTERNAL_H */
