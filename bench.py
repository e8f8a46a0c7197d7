#!/usr/bin/env python3
"""Headline benchmark: ensemble-member-years / second, two-layer model, 1750-2500 annual axis
(751 points, 750 steps), f64, synthetic forcing + Latin-hypercube parameter ensemble.

A "step" is one pass of the hot path over one batch: all members stepped through all 750 model
years (one launch of the fused RK4 kernel), outputs Ts/Td written for every year to HBM.
Inputs (parameters, forcing, initial state) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          (no launcher: the N ranks are started here)

One process per GPU; members are sharded in contiguous blocks (weak scaling: --members per GPU);
the only collectives are the contract's barrier and the max-over-ranks of the wall time (plus, outside
the timed region, an all-reduce of ones and an all-gather of each rank's own times for the `collective`
and `per_rank` objects of the line).  Rank 0 prints ONE JSON line.

Started with --gpus N > 1 and no launcher environment (no RANK / WORLD_SIZE), this process becomes a
launcher: it never imports torch.cuda nor loads librscm_gpu.so, starts N children of itself with RANK,
LOCAL_RANK, WORLD_SIZE and MASTER_* set, relays rank 0's line and exits non-zero if any rank failed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T0, T1 = 1750, 2500
TL_LOW = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0])
TL_HIGH = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])
SEED = 20260327
WATCHDOG_EXIT = 3              # the watchdog ended the process: the line on stdout is valid, an extra or the teardown did not return
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
HBM_STORE_STREAM_GBS = 5460.0  # what a store-only streaming kernel reaches on the card (tools/hbm_stream.hip, profiles/r2_hbm_stream.txt)
FP64_VALU_PEAK_TINSTR = 39.3   # 256 CU x 4 SIMD x 16 f64 lanes x 2.4 GHz (FMA would count 2 flops)
ALG_BYTES_PER_MEMBER_YEAR = 16.0   # store Ts, Td
ALG_OPS_PER_MEMBER_YEAR = 700.0    # 620 add/mul + 80 div, exact mode


def profile_entry(kind, members, mode):
    """The committed rocprofv3 PMC summary of (kind, members, mode) as profiles/traffic.json records it: HBM bytes per launch, and --
    where scripts/update_traffic.py found them in the summary -- the shader clock the chip held under that kernel, the measured VALU
    issue utilisation and the executed vector instructions per wavefront-year.  {} where no profile of that configuration exists."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return {}
    return table.get(f"{kind}|{members}|{mode}") or {}


def measured_traffic(kind, members, mode):
    """HBM bytes per launch of (kind, members, mode) from the committed rocprofv3 PMC summaries
    (profiles/traffic.json); (None, None) where no profile of that configuration exists."""
    e = profile_entry(kind, members, mode)
    return (e["bytes"], e["source"]) if e else (None, None)


def two_layer_rooflines(members, years, kernel_ms, mode, kind="two_layer", bytes_per_member_year=ALG_BYTES_PER_MEMBER_YEAR):
    """The `roofline` object (HBM, as the contract asks) and the FP64-VALU one (the binding roof)
    for one launch of `members` x `years` taking `kernel_ms`."""
    my = members * years
    gbs = bytes_per_member_year * my / (kernel_ms * 1e-3) / 1e9
    traffic, source = measured_traffic(kind, members, mode)
    prof = profile_entry(kind, members, mode)
    hbm = {"bound": "hbm", "binding": "fp64_valu", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
           # the clock the chip held under this kernel when the PMC passes were taken (GRBM_GUI_ACTIVE / 8 XCDs / dispatch time):
           # fractions of the FP64 issue peak below are of the NOMINAL 2.4 GHz; the measured issue utilisation is at this clock
           "effective_clock_ghz": prof.get("clock_ghz"), "nominal_clock_ghz": 2.4,
           "peak_measured_store_stream": HBM_STORE_STREAM_GBS, "frac_of_measured": gbs / HBM_STORE_STREAM_GBS,
           "traffic_from": "profiles/traffic.json" if source else None,   # a committed PMC figure of this launch, not re-measured in this run
           "traffic_source": (f"{source}: separate rocprofv3 --pmc passes of this launch (FETCH_SIZE x2 + WRITE_SIZE, KiB), "
                              "read from profiles/traffic.json; not re-measured inside bench.py") if source else None,
           "kernel": ("coupled_fast_kernel" if mode == "fast" else "coupled_kernel") if kind == "coupled" else "two_layer_kernel",
           "kernel_ms": kernel_ms,
           "algorithmic_bytes": bytes_per_member_year * my,
           "note": f"algorithmic {bytes_per_member_year:g} B/member-year x members x {years} / launch duration; `bound` is the "
                   "roof the contract prices against, `binding` the one that limits the kernel: arithmetic intensity "
                   + ("45-110 f64 op/B against a ridge of ~5 (roofline_fp64_valu)" if kind != "coupled" else
                      "8-28 f64 instructions per byte stored against a ridge of ~5 (roofline_fp64_valu)")}
    # EXACT: the reference's 620 separately rounded add/mul + 80 divisions per member-year; FAST folds the heat
    # capacities into the coefficients, fuses, and steps (Ts, Ts - Td): 30 instructions per RK4 step (csrc/two_layer_body.hpp)
    if kind == "coupled":
        # executed vector instructions per wavefront-year, from the ISA / the SQ_INSTS_VALU counter (profiles/r4_coupled_1e6.txt,
        # profiles/r4_coupled_fast_1e6.txt): the chain has no algorithmic count in SURVEY.md, so these are as-executed figures
        ops = 1520.0 if mode == "exact" else 470.0
        note = ("1520 vector instructions per wavefront-year as executed (EXACT: the reference's expression order, IEEE divisions)"
                if mode == "exact" else
                "470 vector instructions per wavefront-year as executed (FAST: 10 x 30 two-layer, 10 x 4 carbon box, exp, log, one division)")
    else:
        ops = ALG_OPS_PER_MEMBER_YEAR if mode == "exact" else 300.0
        note = ("algorithmic 620 add/mul + 80 div per member-year (unfused, div counted as 1)" if mode == "exact" else
                "300 fused f64 instructions per member-year (10 RK4 steps x 30)")
    tins = ops * my / (kernel_ms * 1e-3) / 1e12
    valu = {"achieved": tins, "peak": FP64_VALU_PEAK_TINSTR, "unit": "T f64-instr/s", "frac": tins / FP64_VALU_PEAK_TINSTR, "note": note,
            "parallelism_bound": two_layer_parallelism_bound(members, years, kernel_ms) if (kind == "two_layer" and mode == "exact") else None,
            "effective_clock_ghz": prof.get("clock_ghz"), "nominal_clock_ghz": 2.4,
            "measured_valu_issue_utilisation_at_effective_clock": prof.get("valu_issue_utilisation"),
            "valu_instructions_per_wavefront_year_executed": prof.get("valu_per_wavefront_year"),
            "counters_source": prof.get("source")}
    return hbm, valu


CONFIGS3_SERIES = 36                     # stored variables of the MAGICC graph (scripts/bench_magicc_chain.py)


def configs3_roofline(members, steps, run_s, ranks=1, mode="fast"):
    """`roofline` of a configs[3] share: algorithmic bytes = every stored variable's new row, 36 x 8 B per member and monthly step
    (what the reference's stepper writes into its collection, runtime.rs:480), over the run's wall time.  `traffic` = the HBM bytes
    the step's four kernels move per member-step (separate FETCH_SIZE / WRITE_SIZE passes of scripts/run_configs3_share.py at 125 000
    members, profiles/traffic.json -> profiles/r6_configs3_share_pmc*.txt) x members x steps: ClimateUDEB's two 50-layer columns in and
    out every step and OceanCarbon's recurrence state are what it moves beyond the 288 algorithmic bytes.  The run is 3 launches per
    step in a dependency chain, two thirds of it ClimateUDEB's 12 column solves: FP64 issue and the chain's latency bind, not HBM."""
    alg = CONFIGS3_SERIES * 8.0 * members * steps * ranks
    gbs = alg / run_s / 1e9
    prof = profile_entry("configs3_share", 125000, mode)
    per = prof.get("bytes_per_member_step")
    out = {"bound": "hbm", "binding": "fp64_valu (ClimateUDEB, ~2/3 of a step) + the dependency chain of 3 launches per step",
           "achieved": gbs, "peak": HBM_PEAK_GBS * ranks, "unit": "GB/s", "frac": gbs / (HBM_PEAK_GBS * ranks),
           "traffic": per * members * steps * ranks if per else None,
           "traffic_from": "profiles/traffic.json" if per else None,
           "traffic_source": (f"{prof.get('source')}: {per:.0f} B per member-step measured at {prof.get('members')} members "
                              f"(read {prof.get('read_per_member_step'):.0f} + written {prof.get('written_per_member_step'):.0f}), x members x steps") if per else None,
           "traffic_over_algorithmic": per / (CONFIGS3_SERIES * 8.0) if per else None,
           "hbm_frac_of_measured_traffic": (per * members * steps * ranks / run_s / 1e9) / (HBM_PEAK_GBS * ranks) if per else None,
           "algorithmic_bytes": alg, "algorithmic_bytes_per_member_step": CONFIGS3_SERIES * 8.0,
           "kernel": "udeb_kernel + ocean kernel + group_split_kernel (3 launches per step; round 5: 4)",
           "kernel_us_per_step_profiled": prof.get("kernel_us_per_step"),
           "note": "36 series x 8 B x members x monthly steps / run wall time"}
    valu = None
    if prof.get("valu_issue_utilisation") is not None:
        valu = {"frac": prof["valu_issue_utilisation"], "unit": "of the 4-cycle f64 issue limit, whole step, measured",
                "counters_source": prof.get("source"),
                "note": "4 x sum(SQ_INSTS_VALU) / 1024 SIMDs / sum(kernel cycles) over the step's four kernels at 125 000 members"}
    return out, valu


UDEB_ALG_BYTES_PER_MEMBER_YEAR = 72.0   # 7 output rows + the history row written, ~1 history entry read back (the columns stay on chip)


def udeb_rooflines(members, years, kernel_ms, plan=(1, 1)):
    """`roofline` (HBM, algorithmic 72 B per member-year) and the binding FP64-issue figure of a ClimateUDEB launch.  The executed
    instruction count, the issue utilisation, the clock and the traffic come from the PMC summary of THE KERNEL THAT RUNS at this
    size (profiles/traffic.json): up to 32 768 members the two-wavefront udeb2_kernel (profiles/r6_udeb2_32768.txt), beyond that
    the one-thread udeb_kernel (profiles/r5_udeb_65536.txt; its traffic is linear in the members: parameters in, rows out)."""
    two_wave = members <= 32768
    at = 32768 if two_wave else 65536
    prof = profile_entry("udeb2" if two_wave else "udeb", at, "exact")
    my = members * years
    gbs = UDEB_ALG_BYTES_PER_MEMBER_YEAR * my / (kernel_ms * 1e-3) / 1e9
    traffic = prof.get("bytes") * members / float(at) if prof.get("bytes") else None
    hbm = {"bound": "hbm", "binding": "fp64_valu", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_from": "profiles/traffic.json" if traffic else None,
           "traffic_source": (f"{prof.get('source')}: this kernel measured at {at} members"
                              + ("" if members == at else ", x members / that (same kernel)")) if traffic else None,
           "kernel": "udeb2_kernel" if two_wave else "udeb_kernel", "kernel_ms": kernel_ms,
           "algorithmic_bytes": UDEB_ALG_BYTES_PER_MEMBER_YEAR * my, "launches_per_pass": plan[0] * plan[1],
           "effective_clock_ghz": prof.get("clock_ghz"), "nominal_clock_ghz": 2.4,
           "note": "7 output rows + the history row per member-year; the two 50-layer columns stay in registers + LDS for the launch"}
    out = {"hbm_frac": gbs / HBM_PEAK_GBS, "roofline": hbm}
    per_wy = prof.get("valu_per_wavefront_year")
    if per_wy:   # vector instructions per wavefront-year as EXECUTED (SQ_INSTS_VALU / waves / years; ~12 % are not f64 arithmetic:
        # moves between register files, compares, selects) x 64 lanes against 39.3 T f64 lane-ops/s at the nominal clock; the
        # two-wavefront kernel has two wavefronts per 64 members, each executing about half a member's instructions
        waves_per_64 = 2.0 if two_wave else 1.0
        tins = per_wy * waves_per_64 * my / (kernel_ms * 1e-3) / 1e12
        out["roofline_fp64_valu"] = {"achieved": tins, "peak": FP64_VALU_PEAK_TINSTR, "unit": "T f64-instr/s", "frac": tins / FP64_VALU_PEAK_TINSTR,
                                     "valu_instructions_per_wavefront_year_executed": per_wy, "wavefronts_per_64_members": waves_per_64,
                                     "measured_valu_issue_utilisation": prof.get("valu_issue_utilisation"), "measured_at_members": at,
                                     "effective_clock_ghz": prof.get("clock_ghz"), "nominal_clock_ghz": 2.4,
                                     "counters_source": prof.get("source"),
                                     "note": "one wavefront per SIMD (256 VGPR + 209 AGPR): nothing hides a dependent instruction's latency but the "
                                             "other hemisphere's independent rows; utilisation is of the 4-cycle issue limit at the measured clock"}
        out["fp64_valu_frac"] = tins / FP64_VALU_PEAK_TINSTR
    return out


# What ONE wavefront alone on a SIMD does, and two sharing one (plain launches at sizes that put exactly one / two on every SIMD,
# EXACT, 750 years; profiles/r5_queue_experiment.txt section 3): the alone one runs at 0.93 of a saturated SIMD's rate.
TL_EXACT_MS_ONE_WAVE_PER_SIMD = 1.505
TL_EXACT_MS_TWO_WAVES_PER_SIMD = 2.807


def two_layer_parallelism_bound(members, years, kernel_ms):
    """The EXACT two-layer configuration's own bound when it has fewer than two wavefront-sized member blocks per SIMD: the blocks are
    independent chains that cannot be cut in members or overlapped in time, a SIMD needs two of them to be saturated, so with
    1024 < blocks < 2048 the best any schedule can do is keep (blocks - 1024) SIMDs at the two-chain rate and the others at the
    one-chain rate to the end.  None where the configuration fills the chip (>= 2 blocks per SIMD) or fits in one round."""
    simds = 1024
    blocks = -(-members // 64)
    if not (simds < blocks < 2 * simds):
        return None
    paired = blocks - simds
    rate = paired * 2.0 / TL_EXACT_MS_TWO_WAVES_PER_SIMD + (simds - paired) * 1.0 / TL_EXACT_MS_ONE_WAVE_PER_SIMD   # blocks x 750 years per ms
    bound_ms = blocks / rate * years / 750.0
    return {"ms": bound_ms, "achieved_frac": bound_ms / kernel_ms, "blocks": blocks, "simds": simds,
            "ms_one_wavefront_per_simd": TL_EXACT_MS_ONE_WAVE_PER_SIMD, "ms_two_wavefronts_per_simd": TL_EXACT_MS_TWO_WAVES_PER_SIMD,
            "note": f"{blocks} independent 64-member chains on {simds} SIMDs: {paired} SIMDs can hold two (saturated), {simds - paired} hold one "
                    "(0.93 of the saturated rate); measured rates from profiles/r5_queue_experiment.txt"}


def describe_run_plan(roofline, plan):
    """How one pass was issued (rscm_ens_last_run_plan): `kernel_ms` is the HIP-event time of a whole pass on the launch stream, which
    forks into and joins the library's second stream -- with a cut it covers blocks x chunks overlapping launches of the kernel."""
    blocks, chunks = plan[0], plan[1]
    roofline["launches_per_pass"] = blocks * chunks
    if blocks * chunks > 1:
        roofline["run_plan"] = (f"{blocks} member blocks on two streams x {chunks} chunks of model steps, issued in turn: the same kernel on the same "
                                "operands, the wavefronts evened out over the SIMDs (include/rscm_gpu.h, rscm_ens_last_run_plan); kernel_ms = "
                                "HIP events around the whole pass / passes, the launches overlap")


# ---- the line the driver parses ------------------------------------------------------------------------------------------
# bench.py measures into a FULL record (every roofline object with its notes and sources, per-rank facts, run plans, ensemble
# statistics).  That record goes to a side file (--details, default ./bench_details.json) and to stderr; stdout carries ONE compact
# line built from it by `compact_line`: the contract's keys, numbers rounded to 6 significant digits, `extra` numbers only.
# The driver keeps 8 KB of stdout: LINE_LIMIT_BYTES is asserted in tests/test_bench_line.py at N = 1 and on an 8-rank record.
LINE_LIMIT_BYTES = 6000
LINE_LIMIT_BYTES_8_RANKS = 8000
_EXTRA_RATE_KEYS = ("member_years_per_s", "model_evaluations_per_s")
_EXTRA_MS_KEYS = ("kernel_ms", "ms", "device_ms_per_iteration")


def _num(x, sig=6):
    """A JSON-safe number: bools and ints as they are, floats rounded to `sig` significant digits, NaN / inf -> None."""
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    return None


def _pick(obj, keys):
    out = {}
    for k in keys:
        v = (obj or {}).get(k)
        if isinstance(v, str):
            out[k] = v if len(v) <= 120 else v[:117] + "..."
        elif isinstance(v, (list, tuple)):
            out[k] = [_num(e) for e in v]
        elif isinstance(v, dict):
            continue
        else:
            out[k] = _num(v)
    return out


def _short_binding(text):
    """`binding` names a roof, not a paragraph: the part before the first parenthesis."""
    return text.split(" (")[0].strip() if isinstance(text, str) else text


def compact_extra(e):
    """One extra as numbers only: rate (member-years/s or model evaluations/s), ms (its launch / pass / iteration time; seconds-long
    runs as s), frac (of the HBM roof, algorithmic bytes), fp64_frac (of the FP64 issue roof, where the record has it), traffic (HBM
    bytes per launch from the committed PMC passes, where the record has it), weak_efficiency, failed (members flagged); a failed
    extra as {"error": "..."} (the full text is in the details)."""
    if not isinstance(e, dict):
        return None
    if "error" in e:
        return {"error": str(e["error"])[:80]}
    out = {}
    for k in _EXTRA_RATE_KEYS:
        if e.get(k) is not None:
            out["rate"] = _num(e[k])
            break
    rank_ms = (e.get("per_rank") or {}).get("kernel_ms")
    for k in _EXTRA_MS_KEYS:
        if e.get(k) is not None:
            out["ms"] = _num(e[k])
            break
    else:
        if rank_ms:   # an all-ranks extra of resident launches: the slowest rank's launch duration
            out["ms"] = _num(max(rank_ms))
    if "ms" not in out:
        for k in ("run_s", "wall_s"):
            if e.get(k) is not None:
                out["s"] = _num(e[k])
                break
    roof = e.get("roofline") or {}
    frac = roof.get("frac", e.get("hbm_frac"))
    if frac is not None:
        out["frac"] = _num(frac)
    valu = e.get("roofline_fp64_valu") or {}
    if valu.get("frac") is not None:
        out["fp64_frac"] = _num(valu["frac"])
    if roof.get("traffic") is not None:
        out["traffic"] = _num(roof["traffic"])
    if e.get("weak_efficiency") is not None and e.get("ranks", 1) > 1:
        out["weak_efficiency"] = _num(e["weak_efficiency"])
    for k in ("speedup", "exchange_share_of_iteration"):
        if e.get(k) is not None:
            out[k] = _num(e[k])
    if e.get("failed_members"):
        out["failed"] = _num(e["failed_members"])
    return out


def compact_line(full, details_path=None):
    """The ONE stdout line from the full record: exactly the contract's keys (see the module docstring of tests/test_bench_line.py),
    strict JSON (no NaN / Infinity), no prose beyond `config.workload`, `cpu_baseline.sample` and the kernel / backend names."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    if full.get("rendezvous_only"):
        line["rendezvous_only"] = True
    line["config"] = _pick(full.get("config"), ("workload", "members_per_gpu", "years", "arithmetic_mode", "failed_members"))
    roof = full.get("roofline")
    if roof is not None:
        r = _pick(roof, ("bound", "binding", "achieved", "peak", "unit", "frac", "traffic", "traffic_from", "algorithmic_bytes", "kernel",
                         "kernel_ms", "launches_per_pass", "effective_clock_ghz"))
        r["binding"] = _short_binding(r.get("binding"))
        line["roofline"] = r
    valu = full.get("roofline_fp64_valu")
    if valu is not None:
        v = _pick(valu, ("frac", "achieved", "peak", "unit"))
        v["issue_utilisation"] = _num(valu.get("measured_valu_issue_utilisation_at_effective_clock"))
        pb = valu.get("parallelism_bound")
        v["parallelism_bound"] = _pick(pb, ("ms", "achieved_frac")) if pb else None
        line["roofline_fp64_valu"] = v
    cpu = full.get("cpu_baseline")
    if cpu is None or "error" in cpu:
        line["cpu_baseline"] = cpu if cpu is None else {"error": str(cpu["error"])[:120]}
    else:
        line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "threads_used", "single_thread_value", "kind", "sample"))
    line["collective"] = _pick(full.get("collective"), ("backend", "world", "rccl_ranks_seen", "ranks_seen"))
    gather = (full.get("collective") or {}).get("loss_gather")
    if gather:
        line["collective"]["loss_gather_ms"] = _num(gather.get("ms")) if "error" not in gather else None
    line["per_rank"] = _pick(full.get("per_rank"), ("kernel_ms", "weak_efficiency"))
    line["extra"] = {k: compact_extra(v) for k, v in (full.get("extra") or {}).items()}
    if full.get("watchdog"):
        line["watchdog"] = str(full["watchdog"])[:160]
    if details_path:
        line["details"] = details_path
    return line


def dumps_line(line):
    """Strict JSON on one line, no spaces after the separators."""
    return json.dumps(line, allow_nan=False, separators=(",", ":"))


def _jsonable(x):
    """The full record with NaN / inf as None and numpy scalars as Python ones (the details file is strict JSON too)."""
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        return None if (x != x or x in (float("inf"), float("-inf"))) else x
    return x


def write_details(full, path):
    """The full record -> `path` (best effort: a read-only cwd must not cost the line) and, always, stderr."""
    text = json.dumps(_jsonable(full), allow_nan=False)
    written = None
    if path:
        try:
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                f.write(text + "\n")
            os.replace(tmp, path)
            written = path
        except OSError as exc:
            sys.stderr.write(f"bench.py: could not write {path}: {exc}\n")
    sys.stderr.write("bench.py details: " + text + "\n")
    sys.stderr.flush()
    return written


def f_syn(t):
    return 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2.0 * np.pi * (t - 1750.0) / 11.0)


def make_ensemble(members, device, rank, world, mode, stream=None, coupled=False):
    import rscm_amd
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    bounds = np.append(t, t[-1] + (t[-1] - t[-2]))
    kind = rscm_amd.KIND_COUPLED if coupled else rscm_amd.KIND_TWO_LAYER
    ens = rscm_amd.Ensemble(kind, members, bounds, device=device)
    ens.set_mode(mode)
    if stream is not None:
        ens.set_stream(stream)
    # global Latin hypercube over world*members members; each rank generates its own block
    if coupled:  # BASELINE configs[2]: + tau in [15,40], alpha_T in [0,0.1]; conc_pi 278, erf_2xco2 3.7
        lo = np.concatenate([TL_LOW, [15.0, 278.0, 0.0, 3.7]])
        hi = np.concatenate([TL_HIGH, [40.0, 278.0, 0.1, 3.7]])
        ens.sample_lhs(SEED, lo, hi, rank * members, world * members)
        yrs = np.array([1750.0, 1850.0, 1950.0, 2000.0, 2020.0, 2050.0, 2100.0])
        ens.set_forcing(np.interp(t, yrs, [0.0, 0.5, 3.0, 7.0, 10.0, 5.0, 1.0]))
        for var, v in (("Atmospheric Concentration|CO2", 278.0), ("Cumulative Land Uptake", 0.0),
                       ("Cumulative Emissions|CO2", 0.0)):
            ens.set_initial(var, v)
    else:
        ens.sample_lhs(SEED, TL_LOW, TL_HIGH, rank * members, world * members)
        ens.set_forcing(f_syn(t))
    ens.set_initial("Surface Temperature", 0.0)
    ens.set_initial("Deep Ocean Temperature", 0.0)
    return ens


def make_udeb_ensemble(members, device, stream=None):
    """ClimateUDEB defaults with ECS, kappa, RLO and k_lo drawn from a Latin hypercube."""
    import rscm_amd
    from rscm_amd import _lib
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    ens = rscm_amd.Ensemble(rscm_amd.KIND_UDEB, members, np.append(t, t[-1] + 1.0), device=device)
    if stream is not None:
        ens.set_stream(stream)
    lo = np.array(_lib.UD_DEFAULTS, dtype=np.float64)
    hi = lo.copy()
    for name, (a, b) in dict(ecs=(2.0, 5.0), kappa=(0.5, 1.5), rlo=(1.2, 1.45), k_lo=(1.0, 2.0)).items():
        j = _lib.UD_PARAM_NAMES.index(name)
        lo[j], hi[j] = a, b
    ens.sample_lhs(SEED, lo, hi)
    ens.set_forcing(f_syn(t))
    for v in (1, 2, 3, 4):
        ens.set_initial(v, 0.0)
    return ens


def make_ghg_ensemble(members, device, method, stream=None):
    """GhgForcing (rscm-magicc): pre-industrial concentrations, CO2 sensitivity and the rapid
    adjustments drawn from a Latin hypercube; a smooth synthetic concentration scenario."""
    import rscm_amd
    from rscm_amd import _lib
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    ens = rscm_amd.Ensemble(rscm_amd.KIND_GHG_FORCING, members, np.append(t, t[-1] + 1.0), device=device)
    if stream is not None:
        ens.set_stream(stream)
    lo = np.array(_lib.GH_DEFAULTS, dtype=np.float64)
    lo[0] = _lib.GH_METHODS[method]
    hi = lo.copy()
    for name, (a, b) in dict(co2_pi=(275.0, 281.0), ch4_pi=(700.0, 740.0), n2o_pi=(265.0, 275.0), delq2xco2=(3.5, 4.0),
                             adjust_co2=(0.95, 1.1), adjust_ch4=(0.8, 0.95), adjust_n2o=(0.9, 1.05)).items():
        j = _lib.GH_PARAM_NAMES.index(name)
        lo[j], hi[j] = a, b
    ens.sample_lhs(SEED, lo, hi)
    yr = t - T0
    ens.set_forcing(np.stack([278.0 * 1.0015 ** yr, 722.0 + 2.0 * yr, 270.0 + 0.1 * yr]))
    return ens


def calibration_sampler(device):
    """two-layer, 6 parameters, Surface Temperature observations 1850..2020 step 10
    (sigma 0.1 K) from the default-parameter run; (runner, device sampler)."""
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.two_layer import TwoLayerBuilder
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    axis = core.TimeAxis.from_values(t)
    fixed = dict(lambda0=1.1, a=0.05, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (core.ModelBuilder().with_device(device).with_time_axis(axis)
         .with_rust_component(TwoLayerBuilder.from_parameters(fixed).build())
         .with_exogenous_variable("Effective Radiative Forcing",
                                  core.Timeseries(f_syn(t), axis, "W/m^2", core.InterpolationStrategy.Linear))
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    names = list(fixed)
    runner = cal.ModelRunner(b, names, ["Surface Temperature"])
    truth = runner.run([fixed[k] for k in names])["Surface Temperature"]
    target = cal.Target()
    for yr in range(1850, 2021, 10):
        target.add_observation("Surface Temperature", float(yr), truth[float(yr)], 0.1)
    params = cal.ParameterSet()
    for k, lo, hi in zip(names, TL_LOW, TL_HIGH):
        params.add(k, cal.Uniform(float(lo), float(hi)))
    return runner, cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)


def calibration_extra(device, walkers=100_000, iterations=20):
    """One iteration = two half-ensemble evaluations (stretch move, fused run + likelihood, accept step on the device)."""
    from rscm_amd import calibrate as cal
    runner, sampler = calibration_sampler(device)
    rng = np.random.default_rng(SEED)
    sampler.run(2, cal.WalkerInit.from_prior(), n_walkers=walkers, rng=rng, seed=1)  # warm-up
    t0 = time.perf_counter()
    sampler.run(iterations, cal.WalkerInit.from_prior(), thin=iterations, n_walkers=walkers, rng=rng, seed=2)
    dt = time.perf_counter() - t0
    runner.close()
    return {"model_evaluations_per_s": walkers * iterations / (sampler.device_ms * 1e-3),
            "device_ms_per_iteration": sampler.device_ms / iterations, "wall_s_per_iteration": dt / iterations,
            "walkers": walkers, "acceptance_rate": sampler.acceptance_rate()}


def graph_calibration_extra(device, walkers=100_000, iterations=10, mode=0):
    """The calibration loop with a GRAPH as the evaluator (rscm_sampler_create_graph): CarbonCycle -> CO2ERF -> Sum -> TwoLayer as
    four linked ensembles, TwoLayer.lambda0 and CarbonCycle.tau sampled (two owners), Ts and CO2 observed 1800..1940 (two owners);
    per half-step the graph runs 190 steps in one launch and is scored on the device."""
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.components import CarbonCycleBuilder, CO2ERFBuilder
    from rscm_amd.two_layer import TwoLayerBuilder
    t = np.arange(1750.0, 1951.0)
    axis = core.TimeAxis.from_values(t)
    schema = core.VariableSchema()
    for n in ["Emissions|CO2|Anthropogenic", "Surface Temperature", "Deep Ocean Temperature", "Atmospheric Concentration|CO2",
              "Cumulative Land Uptake", "Cumulative Emissions|CO2", "Effective Radiative Forcing|CO2", "Effective Radiative Forcing|Other"]:
        schema.add_variable(n, "")
    schema.add_aggregate("Effective Radiative Forcing", "W/m^2", "Sum", ["Effective Radiative Forcing|CO2", "Effective Radiative Forcing|Other"])
    tl = dict(lambda0=1.1, a=0.0, efficacy=1.2, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (core.ModelBuilder().with_device(device).with_time_axis(axis).with_schema(schema)
         .with_rust_component(CarbonCycleBuilder.from_parameters(dict(tau=25.0, conc_pi=278.0, alpha_temperature=0.05)).build())
         .with_rust_component(CO2ERFBuilder.from_parameters(dict(erf_2xco2=3.7, conc_pi=278.0)).build())
         .with_rust_component(TwoLayerBuilder.from_parameters(tl).build())
         .with_exogenous_variable("Emissions|CO2|Anthropogenic",
                                  core.Timeseries(np.interp(t, [1750.0, 1850.0, 1950.0], [1.0, 1.5, 4.0]), axis, "", core.InterpolationStrategy.Linear))
         .with_exogenous_variable("Effective Radiative Forcing|Other", core.Timeseries(0.2 * np.sin(t / 9.0), axis, "", core.InterpolationStrategy.Linear))
         .with_initial_values({"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0, "Atmospheric Concentration|CO2": 278.0,
                               "Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    runner = cal.ModelRunner(b, ["TwoLayer.lambda0", "tau"], ["Surface Temperature", "Atmospheric Concentration|CO2"], mode=mode)
    truth = runner.run([1.25, 30.0])
    target = cal.Target()
    for yr in range(1800, 1941, 10):
        target.add_observation("Surface Temperature", float(yr), truth["Surface Temperature"][float(yr)], 0.005)
        target.add_observation("Atmospheric Concentration|CO2", float(yr), truth["Atmospheric Concentration|CO2"][float(yr)], 0.1)
    params = cal.ParameterSet().add("TwoLayer.lambda0", cal.Uniform(0.8, 1.6)).add("tau", cal.Uniform(15.0, 45.0))
    sampler = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    sampler.run(2, cal.WalkerInit.from_prior(), n_walkers=walkers, seed=1)  # warm-up
    t0 = time.perf_counter()
    sampler.run(iterations, cal.WalkerInit.from_prior(), thin=iterations, n_walkers=walkers, seed=2)
    dt = time.perf_counter() - t0
    runner.close()
    return {"model_evaluations_per_s": walkers * iterations / (sampler.device_ms * 1e-3),
            "device_ms_per_iteration": sampler.device_ms / iterations, "wall_s_per_iteration": dt / iterations,
            "walkers": walkers, "steps_per_evaluation": 190, "acceptance_rate": sampler.acceptance_rate(),
            "arithmetic_mode": "fast" if mode else "exact",
            "note": "four linked ensembles as the sampler's evaluator; proposals, lock-step run, likelihood and accept step on the device"}


def _in_group(dist):
    return dist.is_available() and dist.is_initialized()


def _gather_obj(dist, obj):
    """Every rank's `obj`, in rank order (a list of one without a process group)."""
    if not _in_group(dist):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def scale_measure(rank, world, torch, dist, prepare, body, sync, cleanup, reps=None, min_seconds=0.4, solo=True, after=None):
    """One side measurement taken on ALL ranks at once -- the N > 1 counterpart of `side` -- with the headline's procedure: untimed
    warm-up, then `reps` units of work between a barrier + synchronize on both sides, the wall time = max over ranks, every rank's
    own time beside it.  With more than one rank, rank 0 first runs the same `reps` units ALONE (the other ranks wait at a
    barrier): `weak_efficiency` = that time / the all-ranks wall time, an in-job estimate of the N = 1 figure of the same extra on
    the same card minutes apart (the driver computes its own from its N = 1 run).

    prepare() -> state; body(state, alone, k) issues unit k of a timed region; sync(state) waits for it; cleanup(state); after(state) -> dict
    of per-rank facts gathered into `per_rank`.  No collective is entered inside prepare / body / after; a rank whose prepare fails
    tells the others so before any barrier, and a body that raises still reaches the closing barrier -- a failing extra is
    reported in its place and never hangs the job."""
    group = _in_group(dist)
    state, err = None, None
    try:
        state = prepare()
    except Exception as exc:  # noqa: BLE001 -- reported, not hidden
        err = f"{type(exc).__name__}: {exc}"[:300]
    errs = _gather_obj(dist, err)
    if any(errs):
        if state is not None:
            cleanup(state)
        return {"error": "prepare failed on rank(s) " + ", ".join(f"{r}: {e}" for r, e in enumerate(errs) if e)}

    def barrier():
        if group:
            dist.barrier()

    def device_sync():
        if torch.cuda.is_available():   # (--rendezvous-only exercises this procedure on the CPU)
            torch.cuda.synchronize()

    def run_units(n, alone):
        device_sync()
        t0 = time.perf_counter()
        for k in range(n):
            body(state, alone, k)
        sync(state)
        device_sync()
        return time.perf_counter() - t0

    result = {}
    try:
        try:
            warm = run_units(1, False)          # untimed in the result: first-touch of the series, code objects, communicators
        except Exception as exc:  # noqa: BLE001
            warm, err = None, f"{type(exc).__name__}: {exc}"[:300]
        warms = _gather_obj(dist, (warm, err))
        if any(e for _, e in warms):
            return {"error": "warm-up failed on rank(s) " + ", ".join(f"{r}: {e}" for r, (_, e) in enumerate(warms) if e)}
        if reps is None:
            reps = int(min(200, max(1, -(-min_seconds // max(1e-4, min(w for w, _ in warms))))))
        alone_s = None
        if group and solo and world > 1:
            barrier()
            if rank == 0:
                try:
                    alone_s = run_units(reps, True)
                except Exception as exc:  # noqa: BLE001
                    err = f"{type(exc).__name__}: {exc}"[:300]
            barrier()
        barrier()
        t0 = time.perf_counter()
        mine = None
        try:
            mine = run_units(reps, False)
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"[:300]
        finally:
            barrier()
        wall = time.perf_counter() - t0
        facts = None
        if err is None and after is not None:
            try:
                facts = after(state)
            except Exception as exc:  # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"[:300]
        rows = _gather_obj(dist, {"wall_s": wall, "own_s": mine, "alone_s": alone_s, "error": err, "facts": facts})
        if any(r["error"] for r in rows):
            return {"error": "failed on rank(s) " + ", ".join(f"{k}: {r['error']}" for k, r in enumerate(rows) if r["error"])}
        wall = max(r["wall_s"] for r in rows)
        result = {"ranks": len(rows), "units_per_rank": reps, "warmup_units": 1, "wall_s": wall,
                  "wall_s_per_unit": wall / reps,
                  "per_rank": {"own_s": [r["own_s"] for r in rows], "facts": [r["facts"] for r in rows]},
                  "timing": "barrier + synchronize | units | synchronize + barrier; wall_s = max over ranks of that; own_s = each rank's "
                            "time from its opening barrier to its own synchronize"}
        alone = rows[0]["alone_s"]
        if alone is not None:
            result["rank0_alone_s"] = alone
            result["weak_efficiency"] = alone / wall
            result["weak_efficiency_note"] = ("rank 0 running the same units alone in this job (the other ranks waiting at a barrier) / "
                                              "the all-ranks wall time")
        else:
            result["weak_efficiency"] = 1.0 if len(rows) == 1 else None
        return result
    finally:
        cleanup(state)


def scale_extras(args, rank, local_rank, world, torch, dist, tstream, stream, years, extra):
    """The configs that are DEFINED on more than one GPU, measured on every rank at once (and, under the same keys, on the one rank
    of an N = 1 run, so that the per-N lines compare): the north-star's weak-scaling point (1e6 members per GPU, two-layer EXACT
    and the coupled chain FAST), each rank's configs[3] share, and configs[4]'s sampler sharded over the process group."""

    def both(label, fn):
        try:
            out = fn()
        except Exception as exc:  # noqa: BLE001 -- a rank that raises outside scale_measure's own guards
            out = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if rank == 0:
            extra[label] = out
            if "error" in out:
                print(f"bench.py: extra {label} failed: {out['error']}", file=sys.stderr)

    def resident(members, mode, coupled):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        box = {}

        def prepare():
            return make_ensemble(members, local_rank, rank, world, mode, stream, coupled=coupled)

        def body(ens, alone, k):
            if k == 0:   # first unit of a timed region: restart the HIP-event bracket (the last region is the all-ranks one)
                box["n"] = 0
                ev[0].record(tstream)
            one_pass(ens)
            box["n"] += 1
            ev[1].record(tstream)

        def after(ens):
            return {"kernel_ms": ev[0].elapsed_time(ev[1]) / max(1, box["n"]), "run_plan": list(ens.last_run_plan()),
                    "failed_members": int(ens.status().sum())}

        out = scale_measure(rank, world, torch, dist, prepare, body, lambda e: e.sync(), lambda e: e.close(),
                            min_seconds=0.4, after=after)
        if "error" in out:
            return out
        n = out["units_per_rank"]
        out["member_years_per_s"] = float(out["ranks"]) * members * years * n / out["wall_s"]
        out["members_per_gpu"] = members
        kms = [f["kernel_ms"] for f in out["per_rank"]["facts"]]
        out["per_rank"]["kernel_ms"] = kms
        bpy = 56.0 if coupled else ALG_BYTES_PER_MEMBER_YEAR
        hbm, valu = two_layer_rooflines(members, years, max(kms), "fast" if mode else "exact", "coupled" if coupled else "two_layer", bpy)
        describe_run_plan(hbm, out["per_rank"]["facts"][0]["run_plan"])
        hbm["note"] = "slowest rank's launch duration; " + hbm["note"]
        out["roofline"], out["roofline_fp64_valu"] = hbm, valu
        return out

    m6 = args.scale_members
    both("scale_exact_1e6", lambda: resident(m6, 0, False))
    both("scale_coupled_fast_1e6", lambda: resident(m6, 1, True))

    def share():
        from scripts import run_configs3_share as prog
        members, yrs = args.share_members, args.share_years

        def prepare():
            from rscm_amd import _lib as L
            free0 = L.mem_info(local_rank)[0]
            t0 = time.perf_counter()
            model = prog.build(members, yrs, False, 96, device=local_rank, member_offset=rank * members, members_total=world * members)
            return {"model": model, "build_s": time.perf_counter() - t0, "hbm_gib": (free0 - L.mem_info(local_rank)[0]) / 2**30}

        def body(st, alone, k):
            st["model"].rewind()
            st["model"].run()

        def after(st):
            model = st["model"]
            rows = {n: model.get_series(n, t_stride=12)[:, :64] for n in prog.NAMES}
            small = prog.first_64(members, yrs, False, 96, device=local_rank, member_offset=rank * members, members_total=world * members)
            same = all(bool(np.array_equal(rows[n].view(np.uint64), small[n].view(np.uint64)) or np.array_equal(rows[n], small[n], equal_nan=True))
                       for n in prog.NAMES)
            warm = model.ensembles["Transform:Surface Temperature"].summary(1, yrs * 12)
            return {"build_s": st["build_s"], "hbm_allocated_gib": st["hbm_gib"], "first_64_members_equal_a_64_member_run": same,
                    "failed_members": int(model.ensembles["ClimateUDEB"].status().sum()), "warming_end_K_mean": warm["mean"]}

        out = scale_measure(rank, world, torch, dist, prepare, body, lambda st: None, lambda st: st["model"].close(), reps=1, after=after)
        if "error" in out:
            return out
        facts = out["per_rank"]["facts"]
        out["workload"] = (f"BASELINE configs[3], every rank its own share: MAGICC graph, {members} members x {yrs * 12} monthly steps, "
                           f"window 96 rows + annual outputs, mode FAST; the members of rank r are the block r of one draw of {world * members}")
        out["run_s"] = out["wall_s"]
        out["per_rank"]["run_s"] = out["per_rank"]["own_s"]
        out["member_years_per_s"] = float(out["ranks"]) * members * yrs / out["wall_s"]
        out["roofline"], out["roofline_fp64_valu"] = configs3_roofline(members, yrs * 12, out["wall_s"], out["ranks"])
        out["parity_anchor_all_ranks"] = all(f["first_64_members_equal_a_64_member_run"] for f in facts)
        out["failed_members"] = sum(f["failed_members"] for f in facts)
        if not out["parity_anchor_all_ranks"] or out["failed_members"]:
            out["error"] = "parity anchor or member status failed on a rank (per_rank.facts)"
        return out

    both("scale_configs3_share", share)

    def sampler(per_gpu):
        """configs[4]: `walkers` walkers per iteration in all (per_gpu False: the config as BASELINE.json names it, the evaluations
        of an iteration split over the ranks -- strong scaling), or per GPU (weak scaling: walkers x ranks in all)."""
        from rscm_amd import calibrate as cal
        total = args.scale_walkers * (world if per_gpu else 1)
        iters = args.scale_sweeps

        def prepare():
            runner, smp = calibration_sampler(local_rank)
            return {"runner": runner, "sampler": smp, "ms": {}, "x": {}}

        def body(st, alone, k):
            smp = st["sampler"]
            w = args.scale_walkers if alone else total     # alone: one GPU's worth of this extra, unsharded
            smp.run(iters, cal.WalkerInit.from_prior(), thin=iters, n_walkers=w, rng=np.random.default_rng(SEED), seed=2,
                    shard=False if alone else None)
            st["ms"][alone], st["x"][alone] = smp.device_ms, smp.exchange_ms

        def after(st):
            smp = st["sampler"]
            return {"device_ms_per_iteration": st["ms"][False] / iters, "exchange_ms_per_iteration": st["x"][False] / iters,
                    "exchange_bytes_per_half_step": smp.exchange_bytes_per_half_step, "acceptance_rate": smp.acceptance_rate(),
                    "alone_device_ms_per_iteration": (st["ms"][True] / iters) if True in st["ms"] else None}

        out = scale_measure(rank, world, torch, dist, prepare, body, lambda st: None, lambda st: st["runner"].close(), reps=1, after=after)
        if "error" in out:
            return out
        facts = out["per_rank"]["facts"]
        dm = max(f["device_ms_per_iteration"] for f in facts)
        xm = max(f["exchange_ms_per_iteration"] for f in facts)
        out.update({"walkers": total, "walkers_per_gpu": total // out["ranks"], "iterations": iters, "scaling": "weak" if per_gpu else "strong",
                    "device_ms_per_iteration": dm, "exchange_ms_per_iteration": xm, "exchange_share_of_iteration": xm / dm if dm else None,
                    "model_evaluations_per_s": total / (dm * 1e-3),
                    "note": "rscm_sampler_create_sharded: every rank holds a replica of the walkers and updates its block of each half; per half-step "
                            "the blocks are all-gathered (RCCL with the nccl backend, device to device); device_ms_per_iteration is the slowest "
                            "rank's host-clock time of the sweeps incl. the exchanges; exchange_ms: from this rank's block packed to every "
                            "rank's block landed (the all-gather and the wait for the slowest rank)"})
        if out["ranks"] > 1 and facts[0]["alone_device_ms_per_iteration"]:
            a = facts[0]["alone_device_ms_per_iteration"]   # device-event time of the unsharded loop on rank 0: one GPU's worth
            out["rank0_alone_device_ms_per_iteration"] = a
            if per_gpu:
                out["weak_efficiency"] = a / dm
            else:
                out["speedup"] = a / dm
                out["strong_efficiency"] = a / dm / out["ranks"]
                out["weak_efficiency"] = None
        return out

    both("scale_calibrate_sharded_1e5", lambda: sampler(False))
    both("scale_calibrate_sharded_1e5_per_gpu", lambda: sampler(True))


def one_pass(ens):
    ens.rewind()
    ens.run(sync=False)


def _coll_device(torch, dist):
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def timed_passes(ens, steps, warmup, torch, dist, world, tstream, collective=None):
    # world > 1, or a one-rank process group created for rehearsal (RSCM_BENCH_FORCE_DIST=1): the
    # barrier / max-over-ranks path of the contract runs either way.  collective=False: this rank times alone
    # (the side measurements of rank 0 at N = 1) whatever process group exists.
    if collective is None:
        collective = world > 1 or (dist.is_available() and dist.is_initialized())
    world = 2 if collective else 1
    for _ in range(warmup):
        one_pass(ens)
    ens.sync()
    if world > 1:  # communicator creation and the first collective stay outside the timed region
        dist.barrier()
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=_coll_device(torch, dist)),
                        op=dist.ReduceOp.MAX)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(tstream)
    for _ in range(steps):
        one_pass(ens)
    ev1.record(tstream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / steps  # HIP events on the launch stream
    if world > 1:
        w = torch.tensor([wall], dtype=torch.float64, device=_coll_device(torch, dist))
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    return wall, kernel_ms


def loss_gather_report(ens, members, world, dist):
    """Outside the timed region, only with a process group: BASELINE configs[4]'s exchange -- every rank scores its members on the
    device (Gaussian log-likelihood against 18 Surface Temperature observations 1850..2020) and the ranks all-gather the per-member
    losses from device memory (rscm_amd.distributed.gather_members: RCCL over xGMI with the nccl backend, 8 B per member).  Timed with
    the host clock, barrier to return; a failure is reported in the line, it never costs the headline."""
    forced = os.environ.get("RSCM_BENCH_FORCE_DIST") == "1"   # rehearsal: a one-rank RCCL group on a one-GPU box
    if not (dist.is_available() and dist.is_initialized()) or (world < 2 and not forced):
        return None
    try:
        from rscm_amd.distributed import gather_members
        if forced:
            os.environ["RSCM_FORCE_DISTRIBUTED"] = "1"
        tidx = np.arange(1850, 2021, 10) - T0
        obs = 1.0 + 0.004 * tidx
        dist.barrier()
        t0 = time.perf_counter()
        local = ens.loglik(["Surface Temperature"] * len(tidx), tidx, obs, np.full(len(tidx), 0.1), on_device=True)
        full = gather_members(local, world * members)
        dt = time.perf_counter() - t0
        return {"what": "per-member log-likelihood scored on the device, all-gathered over the ranks from device memory",
                "ms": dt * 1e3, "bytes_per_rank": 8 * members, "members_gathered": int(full.size),
                "finite": int(np.isfinite(full).sum())}
    except Exception as exc:  # noqa: BLE001 -- reported, not hidden
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


def rank_report(kernel_ms, wall, steps, torch, dist):
    """Outside the timed region: who took part in the collectives (an all-reduce of ones) and every rank's own
    kernel time, so that a straggler shows in a timed region of a few tens of milliseconds."""
    if not (dist.is_available() and dist.is_initialized()):
        return ({"backend": None, "world": 1, "rccl_ranks_seen": 0, "note": "single process, no process group"},
                {"kernel_ms": [kernel_ms], "weak_efficiency": [kernel_ms / (wall / steps * 1e3)]})
    dev = _coll_device(torch, dist)
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    mine = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    kms = [float(x.item()) for x in every]
    backend = dist.get_backend()
    coll = {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "world": dist.get_world_size(),
            "rccl_ranks_seen": int(round(float(ones.item()))) if backend == "nccl" else 0,
            "ranks_seen": int(round(float(ones.item()))), "tensors_on": dev}
    return coll, {"kernel_ms": kms, "weak_efficiency": [k / (wall / steps * 1e3) for k in kms],
                  "note": "kernel_ms: HIP events around each rank's own K launches / K; weak_efficiency: that over the "
                          "max-over-ranks wall time per step (barrier to barrier), i.e. 1 - the share of the step a rank "
                          "spent waiting for the slowest one and for the barriers"}


def host_description():
    """What the CPU leg ran on: logical cores of the host, the cores this process may use, the cgroup's CPU quota if one is
    set (a leased box may expose all of the host's cores and throttle to a share of them), the CPU model string."""
    info = {"host_cores": os.cpu_count() or 1}
    try:
        info["affinity_cores"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity_cores"] = info["host_cores"]
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        info["cgroup_cpu_quota_cores"] = None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        info["cgroup_cpu_quota_cores"] = None
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info["cpu_model"] = model
    return info


def baseline_threads(host):
    """One thread per core this process is actually GRANTED: the cores it may run on, capped by the cgroup's CPU quota when one
    is set (a leased box shows all of the host's cores and throttles to its share: more threads than the quota only take turns)."""
    import math
    quota = host.get("cgroup_cpu_quota_cores")
    n = host["affinity_cores"]
    if quota:
        n = min(n, max(1, math.ceil(quota)))
    return max(1, n)


def cpu_baseline(threads=None, target_seconds=12.0):
    """The CPU oracle (a port of the reference algorithm) on a bounded sample of the same
    workload, on the cores this process is granted (baseline_threads: affinity capped by the cgroup quota)."""
    from oracle import cbind
    host = host_description()
    threads = threads or baseline_threads(host)
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    b = cbind.bounds_from_values(t)
    F = f_syn(t)
    rng = np.random.default_rng(SEED)

    def params(n):
        return TL_LOW[:, None] + rng.random((6, n)) * (TL_HIGH - TL_LOW)[:, None]

    probe = 256 * threads
    t0 = time.perf_counter()
    cbind.two_layer_run(b, params(probe), F, 0.0, 0.0, threads=threads)
    dt = time.perf_counter() - t0
    # one pass holds 12 KB of output per member on the host: at most 5e5 members (6 GB), repeated to fill the budget
    n = int(min(max(probe, probe * target_seconds / max(dt, 1e-3)), 500_000))
    n -= n % threads
    P = params(n)
    cbind.two_layer_run(b, P[:, :probe], F, 0.0, 0.0, threads=threads)   # threads and pages warm
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while reps == 0 or (dt < 0.6 * target_seconds and reps < 64):
        cbind.two_layer_run(b, P, F, 0.0, 0.0, threads=threads)
        reps += 1
        dt = time.perf_counter() - t0
    n1 = max(256, min(n // (4 * threads) if threads < 64 else 4096, 8192))
    t0 = time.perf_counter()
    cbind.two_layer_run(b, params(n1), F, 0.0, 0.0, threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n * reps * (T1 - T0) / dt, "unit": "member-years/s", "cores": threads,
            "host_cores": host["host_cores"], "threads_used": threads, "affinity_cores": host["affinity_cores"],
            "cgroup_cpu_quota_cores": host["cgroup_cpu_quota_cores"], "cpu_model": host["cpu_model"],
            "kind": "port", "single_thread_value": n1 * (T1 - T0) / dt1,
            "sample": f"{reps} pass(es) x {n} members x {T1 - T0} years, oracle/rscm_oracle.c, {dt:.1f} s on {threads} threads",
            "sample_note": f"{reps} pass(es) over {n} members x {T1 - T0} years, oracle/rscm_oracle.c two_layer_run "
                      f"(-O2 -ffp-contract=off), {dt:.1f} s on {threads} threads (one per core this process is granted: "
                      f"min(affinity {host['affinity_cores']}, ceil(cgroup quota {host['cgroup_cpu_quota_cores']})); "
                      f"the host has {host['host_cores']})"}


def linked_graph_extra(members, device, stream, years, mode=0):
    import time
    import rscm_amd
    from rscm_amd.ensemble import run_lockstep
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    bounds = np.append(t, t[-1] + 1.0)
    fused = make_ensemble(members, device, 0, 1, mode, stream, coupled=True)
    P = fused.get_params()
    fused_best = float("inf")   # the fused coupled kernel on the same card, minutes apart at most: the yardstick
    for _ in range(3):
        fused.rewind()
        fused.sync()
        t0 = time.perf_counter()
        fused.run()
        fused_best = min(fused_best, time.perf_counter() - t0)
    fused.close()
    from rscm_amd import _lib as L0
    L0.check(L0.load().rscm_gpu_lockstep_stats(None, None))  # reset the launch counters
    kinds = (rscm_amd.KIND_CARBON_CYCLE, rscm_amd.KIND_CO2_ERF, rscm_amd.KIND_AGGREGATE, rscm_amd.KIND_TWO_LAYER)
    cc, ce, ag, tl = (rscm_amd.Ensemble(k, members, bounds, device=device) for k in kinds)
    for e in (cc, ce, ag, tl):
        e.set_stream(stream)
        e.set_mode(mode)
    cc.set_params(P[[6, 7, 8]])
    ce.set_params(P[[9, 7]])
    ag.set_params(np.zeros((9, members)))
    tl.set_params(P[:6])
    yrs = np.array([1750.0, 1850.0, 1950.0, 2000.0, 2020.0, 2050.0, 2100.0])
    cc.set_forcing(np.stack([np.interp(t, yrs, [0.0, 0.5, 3.0, 7.0, 10.0, 5.0, 1.0]), np.full(len(t), np.nan)]))
    for var, v in (("Atmospheric Concentration|CO2", 278.0), ("Cumulative Land Uptake", 0.0), ("Cumulative Emissions|CO2", 0.0)):
        cc.set_initial(var, v)
    tl.set_initial(1, 0.0)
    tl.set_initial(2, 0.0)
    cc.link_input(1, tl, 1, rscm_amd.SRC_EXOGENOUS)   # lagged temperature feedback
    ce.link_input(0, cc, 1, rscm_amd.SRC_UPSTREAM)
    ag.link_input(0, ce, 1, rscm_amd.SRC_UPSTREAM)
    tl.link_input(0, ag, 1, rscm_amd.SRC_UPSTREAM)
    best = float("inf")
    for _ in range(3):
        for e in (cc, ce, ag, tl):
            e.rewind()
        tl.sync()
        t0 = time.perf_counter()
        run_lockstep((cc, ce, ag, tl))
        best = min(best, time.perf_counter() - t0)
    cc.unlink_input(1)
    for e in (tl, ag, ce, cc):
        e.close()
    import ctypes as C
    from rscm_amd import _lib as L
    nl, ns = C.c_int64(), C.c_int64()
    L.check(L.load().rscm_gpu_lockstep_stats(C.byref(nl), C.byref(ns)))
    return {"member_years_per_s": members * years / best, "ms": best * 1e3, "launches": int(nl.value) // 3,
            "component_steps": int(ns.value) // 3,
            "fused_coupled_kernel_ms": fused_best * 1e3, "ratio_to_fused_coupled_kernel": best / fused_best,
            # the same 7 series written per member-year as the fused kernel (56 B); the LDS slots keep the reads out of HBM
            "hbm_frac": 56.0 * members * years / best / 1e9 / HBM_PEAK_GBS, "arithmetic_mode": "fast" if mode else "exact",
            "note": "CarbonCycle, CO2ERF, Sum, TwoLayer as four linked ensembles in lock-step; all four are light "
                    "components, so the run is one fused group launch (csrc/group.hip) that keeps parameters, states "
                    "and linked values in LDS between the model steps"}


def magicc_chain_extra(members, years, fast=False):
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location(
        "bench_magicc_chain", os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "bench_magicc_chain.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    model = mod.build_chain(members, years, "topological")
    if fast:  # RSCM_MODE_FAST: OceanCarbon's O(T) recurrence, fused two-layer style arithmetic where a kind has it
        from rscm_amd import _lib as L
        model.set_mode(L.MODE_FAST)
    import ctypes as C
    from rscm_amd import _lib as L2
    L2.check(L2.load().rscm_gpu_lockstep_stats(None, None))  # reset the launch counters
    t0 = time.perf_counter()
    model.run()
    dt = time.perf_counter() - t0
    warm = model.ensembles["Transform:Surface Temperature"].summary(1, years)
    n = len(model._order)
    model.close()
    nl, ns = C.c_int64(), C.c_int64()
    L2.check(L2.load().rscm_gpu_lockstep_stats(C.byref(nl), C.byref(ns)))
    return {"member_years_per_s": members * years / dt, "ms": dt * 1e3, "launches": int(nl.value), "ensembles": n,
            "finite_members": warm["count"], "mean_warming_K": warm["mean"],
            "note": "10 components + aggregate + 2 grid transforms, lock-step in topological order; runs of light "
                    "components share a launch (3 launches per model step: a step's last light launch rides with the next step's first)"}


def end_to_end_extra(members, device, mode, stream, years):
    import time
    from rscm_amd.ensemble import pinned_empty
    ens = make_ensemble(members, device, 0, 1, mode, stream)
    host_params = ens.get_params()
    out = [pinned_empty((T1 - T0 + 1, members)) for _ in range(2)]
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        ens.set_params(host_params)
        ens.rewind()
        ens.run()
        ens.get_series("Surface Temperature", out=out[0])
        ens.get_series("Deep Ocean Temperature", out=out[1])
        best = min(best, time.perf_counter() - t0)
    ens.close()
    moved = host_params.nbytes + sum(o.nbytes for o in out)
    return {"member_years_per_s": members * years / best, "ms": best * 1e3, "host_bytes": moved,
            "note": "H2D params + run + D2H of both series into pinned buffers, best of 3"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks as children of this process (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1 and a free MASTER_PORT in their environment), relay rank 0's JSON line, pass the other
    ranks' output on to stderr, and return non-zero if any rank failed (the others are then ended by their exact PIDs)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's cross-process buffers need it on this host driver
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE, stderr=None if r == 0 else subprocess.PIPE, text=True))
    import threading
    outs = [None] * n

    def drain(r):
        outs[r] = procs[r].communicate()

    threads = [threading.Thread(target=drain, args=(r,), daemon=True) for r in range(n)]
    for th in threads:
        th.start()
    failed = None
    hung_since = None      # a rank's watchdog ended it (WATCHDOG_EXIT): the other ranks' watchdogs end them within ~30 s of their own line
    while any(th.is_alive() for th in threads):
        for r, pr in enumerate(procs):
            code = pr.poll()
            if code == WATCHDOG_EXIT:
                hung_since = hung_since or time.monotonic()
            elif code not in (None, 0) and failed is None:
                failed = (r, code)
        if failed is not None or (hung_since is not None and time.monotonic() - hung_since > 60.0):
            for other in procs:   # one rank down: the others would wait in a collective until its timeout
                if other.poll() is None:
                    other.terminate()
        for th in threads:
            th.join(timeout=0.2)
    for r, pr in enumerate(procs):
        if pr.returncode not in (0, WATCHDOG_EXIT) and failed is None and hung_since is None:
            failed = (r, pr.returncode)
    hung = [r for r, pr in enumerate(procs) if pr.returncode == WATCHDOG_EXIT]
    for r in range(1, n):
        so, se = outs[r] or ("", "")
        if (so or se) and (failed is not None or hung):
            sys.stderr.write(f"---- rank {r} (exit {procs[r].returncode})\n{(so or '')[-2000:]}{(se or '')[-4000:]}\n")
    line = (outs[0] or ("", None))[0] or ""
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with {failed[1]}\n{line[-2000:]}\n")
        return 1
    # rank 0's stdout may carry library chatter ("[Gloo] Rank 0 is connected ..."): the contract is ONE JSON line on stdout
    rows = [x for x in line.splitlines() if x.strip()]
    result = next((x for x in reversed(rows) if x.lstrip().startswith("{")), None)
    for x in rows:
        if x is not result:
            sys.stderr.write(x + "\n")
    if result is None:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 1
    sys.stdout.write(result + "\n")
    sys.stdout.flush()
    if hung:   # line valid, extras (or the teardown) hung on these ranks: the line is relayed, the exit code says so
        sys.stderr.write(f"bench.py: watchdog ended rank(s) {hung}: the headline line above is valid, a side measurement or the "
                         f"teardown did not return (see the ranks' stderr)\n")
        return WATCHDOG_EXIT
    return 0


def start_watchdog(budget_s, emit, rank, present, grace_s=30.0):
    """After `budget_s` seconds: print the line as it stands (emit(note) -> True if this call printed it) and end the process with
    WATCHDOG_EXIT -- non-zero and distinct: the headline line is valid, a side measurement (or, when the line was already out, the
    teardown) did not return.  launch_ranks relays the line and passes the code on.  No exec, no collective: the other ranks'
    watchdogs end them the same way."""
    import threading

    def watchdog():
        time.sleep(max(1.0, budget_s))
        fired = emit(f"side measurements unfinished after {budget_s:g} s; present: {len(present)}")
        if not fired:
            time.sleep(grace_s)   # the line is out already: only a teardown that never returns is left to end
        sys.stderr.write(f"bench.py: rank {rank}: watchdog ends the process {'(line printed by it)' if fired else '(after the line)'}"
                         f" with exit code {WATCHDOG_EXIT}; extras present: {sorted(present)}\n")
        sys.stderr.flush()
        os._exit(WATCHDOG_EXIT)

    threading.Thread(target=watchdog, daemon=True).start()


def rendezvous_only(args, rank, world, torch, dist):
    """--rendezvous-only: everything of the N > 1 procedure that is not GPU work, on the CPU over gloo -- process group, barriers,
    max-over-ranks of a wall time, the rank report -- so that the launcher and the collectives' plumbing are covered where there is
    no GPU.  Prints a line marked as carrying no measurement."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))   # the slowest rank sets the wall time
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        w = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    collective, per_rank = rank_report(10.0 * (rank + 1), wall, 1, torch, dist)
    # the all-ranks side-measurement procedure (scale_measure) with sleeps for work: a healthy one, one whose prepare fails on
    # one rank, one whose timed body raises on one rank -- each must come back on every rank, the failures as reports
    bad = int(os.environ.get("RSCM_BENCH_SELFTEST_BAD_RANK", "1"))

    def selftest(fail_prepare=False, fail_body=False):
        def prepare():
            if fail_prepare and rank == bad:
                raise RuntimeError("prepare refused (self-test)")
            return {"units": 0}

        def body(st, alone, k):
            if fail_body and rank == bad and st["units"] >= 1:
                raise RuntimeError("body raised (self-test)")
            st["units"] += 1
            time.sleep(0.005 * (rank + 1))

        return scale_measure(rank, world, torch, dist, prepare, body, lambda st: None, lambda st: None, reps=3,
                             after=lambda st: {"units": st["units"]})

    tests = {"healthy": selftest(), "prepare_fails": selftest(fail_prepare=True), "body_fails": selftest(fail_body=True)}
    record = {"metric": "ensemble-member-years/sec, two-layer 1750-2500 f64", "value": None, "unit": "member-years/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "rendezvous_only": True,
              "collective": collective, "per_rank": per_rank, "wall_s": wall, "scale_selftest": tests}
    if os.environ.get("RSCM_BENCH_SELFTEST_HANG") == "1":
        # self-test of the watchdog's exit path: a "side measurement" that never returns on any rank.  Rank 0's watchdog prints the
        # line, every rank ends with WATCHDOG_EXIT, the launcher relays the line and returns that code.
        import threading
        once = threading.Lock()

        def emit(note=None):
            if not once.acquire(blocking=False):
                return False
            if rank == 0:
                sys.stdout.write(json.dumps(dict(record, watchdog=note)) + "\n")
                sys.stdout.flush()
            return True

        start_watchdog(args.extras_budget, emit, rank, {})
        time.sleep(3600.0)
    if rank == 0:
        print(json.dumps(record))
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--members", type=int, default=100_000, help="members per GPU (configs[1]: 1e5)")
    ap.add_argument("--mode", choices=["exact", "fast"], default="exact")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary 1e6-member / fast-mode lines")
    ap.add_argument("--scale-members", type=int, default=1_000_000,
                    help="members per GPU of scale_exact_1e6 / scale_coupled_fast_1e6 (the north-star's weak-scaling point; tests pass less)")
    ap.add_argument("--share-members", type=int, default=125_000, help="members per GPU of scale_configs3_share (configs[3]: 1e6 / 8)")
    ap.add_argument("--share-years", type=int, default=750, help="years (x 12 monthly steps) of scale_configs3_share")
    ap.add_argument("--scale-walkers", type=int, default=100_000, help="walkers per iteration of scale_calibrate_sharded_1e5 (configs[4])")
    ap.add_argument("--scale-sweeps", type=int, default=200, help="timed sweeps of scale_calibrate_sharded_1e5")
    ap.add_argument("--scale-only", action="store_true", help="of the extras, only the scale_* ones (tests)")
    ap.add_argument("--extras-budget", type=float, default=540.0,
                    help="seconds after the headline at which a watchdog prints the line as it stands and ends the process, should a side "
                         "measurement hang (default 540; the whole default run takes about a minute)")
    ap.add_argument("--details", default="bench_details.json",
                    help="where rank 0 writes the FULL record (every roofline object with its notes, per-rank facts, run plans); stdout "
                         "carries only the compact line built from it.  Empty string: stderr only")
    ap.add_argument("--no-scale", action="store_true", help="skip the scale_* extras (the configs defined on more than one GPU)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="no GPU work: the ranks meet over gloo, run the contract's barrier / max-over-ranks / rank report and rank 0 "
                         "prints a line with value null (tests/test_distributed_cpu.py checks the launcher and the N > 1 plumbing with it)")
    args = ap.parse_args()

    # No launcher environment and more than one GPU asked for: this process starts the ranks itself and never touches the GPU
    # (decided from argv and the environment alone, before torch.cuda or librscm_gpu.so are loaded).
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree")

    import torch
    import torch.distributed as dist
    # a leased box shows all of the host's cores and grants a share of them: torch's (and gloo's) CPU-side work with one thread per
    # VISIBLE core only takes turns (the two-rank gloo rehearsal of the sharded sampler: 226 ms per exchange with 256 threads)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), baseline_threads(host_description()))))
    if args.rendezvous_only:
        raise SystemExit(rendezvous_only(args, rank, world, torch, dist))
    from rscm_amd import _lib
    _lib.load()  # no CPU fallback: fail here if the HIP extension is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    # Rehearsal knobs (not used by the driver): RSCM_BENCH_BACKEND=gloo runs the collectives on
    # the CPU and RSCM_BENCH_DEVICE pins every rank to one GPU, so the N>1 code path can be
    # exercised on a single-GPU box.
    backend = os.environ.get("RSCM_BENCH_BACKEND", "nccl")
    if "RSCM_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["RSCM_BENCH_DEVICE"])
    if local_rank >= torch.cuda.device_count():  # launcher exposes one GPU per rank as device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RSCM_BENCH_FORCE_DIST=1 (rehearsal): create the process group even for one rank, so that the RCCL
    # communicator, the barrier and the device-tensor all-reduce of the N > 1 path run on a one-GPU box
    force_dist = os.environ.get("RSCM_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # librccl prints its version banner on stdout when the communicator is created: the contract is ONE JSON line there, so
        # stdout points at stderr until the first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    # a real (non-null) HIP stream shared by torch's events and the library's launches
    tstream = torch.cuda.Stream()
    stream = tstream.cuda_stream
    mode = 0 if args.mode == "exact" else 1
    years = T1 - T0

    ens = make_ensemble(args.members, local_rank, rank, world, mode, stream)
    wall, kernel_ms = timed_passes(ens, args.steps, args.warmup, torch, dist, world, tstream)
    n_fail = int(ens.status().sum())
    fails = _gather_obj(dist, n_fail)
    run_plan = ens.last_run_plan()
    s_mid = ens.summary("Surface Temperature", 270)  # year 2020
    gather = loss_gather_report(ens, args.members, world, dist)
    ens.close()
    collective, per_rank = rank_report(kernel_ms, wall, args.steps, torch, dist)
    if gather is not None:
        collective["loss_gather"] = gather

    total_member_years = float(world) * args.members * years * args.steps
    value = total_member_years / wall
    roofline_hbm, roofline_valu = two_layer_rooflines(args.members, years, kernel_ms, args.mode)
    describe_run_plan(roofline_hbm, run_plan)

    extra = {}
    # The record is complete from here on: the headline stands, the CPU baseline is taken next, the side measurements fill `extra`
    # in place.  Should one of them hang (a collective that never returns on some rank, a kernel that never ends), the watchdog
    # prints the line as it is after --extras-budget seconds and ends the process with WATCHDOG_EXIT: a side measurement never
    # costs the headline, not even by not coming back, and a hang is never reported as a clean run.
    out = {
        "metric": "ensemble-member-years/sec, two-layer 1750-2500 f64",
        "value": value,
        "unit": "member-years/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"two-layer energy balance, {args.members} members/GPU x 750 annual steps 1750-2500, RK4 h=0.1, BASELINE configs[1]",
            "members_per_gpu": args.members,
            "years": years,
            "arithmetic_mode": args.mode,
            "forcing": "F_syn (SURVEY 8d), exogenous, 1 scenario in LDS",
            "parameters": "device Latin hypercube over typical ranges, seed 20260327",
            "outputs": "Ts,Td every year to HBM (16 B/member-year)",
            "sharding": f"contiguous member blocks, {world} rank(s), no data-path collective",
            # part of the draw, not a fault: over these ranges a few per cent of the Latin hypercube's members have
            # lambda0 - a*Ts turn negative and run away to inf; they are stepped and stored like the others and flagged
            "failed_members": int(sum(fails)), "failed_members_per_rank": fails,
            "failed_members_fraction": sum(fails) / float(world * args.members),
        },
        "roofline": roofline_hbm,
        "roofline_fp64_valu": roofline_valu,
        "collective": collective,
        "per_rank": per_rank,
        "cpu_baseline": None,
        "check": {"failed_members_rank0": n_fail, "Ts_2020_mean_rank0": s_mid["mean"],
                  "finite_members_2020_rank0": s_mid["count"]},
        "extra": extra,
    }

    # The CPU leg comes right after the headline and before any extra: whatever happens later (a watchdog print included), the line
    # carries `cpu_baseline`.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as exc:  # noqa: BLE001 -- the GPU figure must not be lost to the CPU leg
            out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            print(f"bench.py: cpu_baseline failed: {exc}", file=sys.stderr)

    import threading
    printed = threading.Lock()

    def emit(note=None):
        if not printed.acquire(blocking=False):
            return False
        if rank == 0:
            if note:
                out["watchdog"] = note
            full = None
            for _ in range(5):   # (the watchdog serialises while the main thread may be adding an extra)
                try:
                    full = dict(out, extra=dict(extra))
                    details = write_details(full, args.details)
                    text = dumps_line(compact_line(full, details))
                    break
                except RuntimeError:
                    full = None
                    time.sleep(0.01)
            if full is None:
                text = dumps_line(compact_line(dict(out, extra={})))
            sys.stdout.write(text + "\n")
            sys.stdout.flush()
        return True

    start_watchdog(args.extras_budget, emit, rank, extra)

    def side(label, fn):
        """A side measurement never costs the headline line: a failure is reported in its place."""
        try:
            extra[label] = fn()
        except Exception as exc:  # noqa: BLE001 -- reported, not hidden
            extra[label] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            print(f"bench.py: extra {label} failed: {exc}", file=sys.stderr)

    # Every config that is defined on more than one GPU, on ALL ranks (and under the same keys at N = 1): scale_exact_1e6,
    # scale_coupled_fast_1e6, scale_configs3_share, scale_calibrate_sharded_1e5[_per_gpu].  After the headline, which is unchanged.
    if not args.no_extra and not args.no_scale:
        try:
            scale_extras(args, rank, local_rank, world, torch, dist, tstream, stream, years, extra)
        except Exception as exc:  # noqa: BLE001 -- e.g. a collective that fails because a peer's watchdog has ended it: the headline stands
            print(f"bench.py: rank {rank}: the scale extras stopped: {type(exc).__name__}: {exc}", file=sys.stderr)

    if rank == 0 and world == 1 and not args.no_extra and not args.scale_only:
        def two_layer_case(members, m, cp):
            e2 = make_ensemble(members, local_rank, 0, 1, m, stream, coupled=cp)
            k = max(3, args.steps // 4)
            w2, k2 = timed_passes(e2, k, 1, torch, dist, 1, tstream)
            plan2 = e2.last_run_plan()
            e2.close()
            bpy = 56.0 if cp else ALG_BYTES_PER_MEMBER_YEAR
            out = {"member_years_per_s": members * years * k / w2, "kernel_ms": k2,
                   "hbm_frac": bpy * members * years / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the north-star's target size (1e6 members) and the coupled chain carry their own roofline objects
            hbm, valu = two_layer_rooflines(members, years, k2, "fast" if m else "exact", "coupled" if cp else "two_layer", bpy)
            describe_run_plan(hbm, plan2)
            out["roofline"] = hbm
            out["roofline_fp64_valu"] = valu
            return out

        for label, members, m, cp in (("fast_1e5", args.members, 1, False),
                                      ("exact_1e6", 1_000_000, 0, False),
                                      ("fast_1e6", 1_000_000, 1, False),
                                      ("coupled_1e6", 1_000_000, 0, True),
                                      ("coupled_1e6_fast", 1_000_000, 1, True)):
            side(label, lambda members=members, m=m, cp=cp: two_layer_case(members, m, cp))

        def udeb_case(members):
            # next row: rscm-magicc ClimateUDEB, 12 sub-steps per year (65 536 members: one wavefront per
            # SIMD, exactly one round; 1e5: 1563 wavefronts on 1024 SIMDs, two rounds; 32 768: the two-wavefront kernel)
            e3 = make_udeb_ensemble(members, local_rank, stream)
            w3, k3 = timed_passes(e3, 2, 1, torch, dist, 1, tstream)
            plan3 = e3.last_run_plan()
            e3.close()
            return dict({"member_years_per_s": members * years * 2 / w3, "kernel_ms": k3}, **udeb_rooflines(members, years, k3, plan3))

        side("udeb_1e5", lambda: udeb_case(100_000))
        side("udeb_65536", lambda: udeb_case(65_536))
        side("udeb_32768", lambda: udeb_case(32_768))

        def ghg_case(method):
            # rscm-magicc GhgForcing: a pointwise component, 24 B of ERF written per member-year
            e4 = make_ghg_ensemble(1_000_000, local_rank, method, stream)
            w4, k4 = timed_passes(e4, 5, 2, torch, dist, 1, tstream)
            e4.close()
            return {"member_years_per_s": 1_000_000 * years * 5 / w4, "kernel_ms": k4,
                    "hbm_frac": 24.0 * 1_000_000 * years / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS}

        for label, method in (("ghg_olbl_1e6", "Olbl"), ("ghg_ipcctar_1e6", "Ipcctar")):
            side(label, lambda method=method: ghg_case(method))

        # the same coupled chain assembled from four linked ensembles and stepped in graph order
        # (rscm_ens_link_input / rscm_ens_run_lockstep): what an arbitrary component graph costs
        side("coupled_linked_1e6", lambda: linked_graph_extra(1_000_000, local_rank, stream, years))
        side("coupled_linked_1e6_fast", lambda: linked_graph_extra(1_000_000, local_rank, stream, years, mode=1))

        # BASELINE.json configs[3]: the emissions-driven MAGICC graph (ten rscm-magicc components, Sum of
        # eight forcings, FourBox transforms) as linked ensembles, ClimateUDEB / OceanCarbon at 12 sub-steps
        side("magicc_chain_1e5", lambda: magicc_chain_extra(100_000, years))
        side("magicc_chain_1e5_fast", lambda: magicc_chain_extra(100_000, years, fast=True))

        # BASELINE.json configs[3], one GPU's share at full size: 125 000 members x 9000 MONTHLY steps of the MAGICC
        # graph, windowed series + annual outputs (scripts/run_configs3_share.py)
        def configs3_share(*flags):
            import contextlib
            import io
            from scripts import run_configs3_share as prog
            argv, sys.argv = sys.argv, ["run_configs3_share.py", *flags]
            buf = io.StringIO()
            code = 0
            try:
                with contextlib.redirect_stdout(buf):
                    try:
                        prog.main()
                    except SystemExit as done:
                        code = done.code
            finally:
                sys.argv = argv
            out = json.loads(buf.getvalue().strip().splitlines()[-1])
            out["exit_code"] = code
            out["roofline"], out["roofline_fp64_valu"] = configs3_roofline(125_000, 9000, out["run_s"], mode="exact" if "--exact" in flags else "fast")
            if code not in (0, None):  # the parity anchor (first 64 members == a 64-member run) or a member failed
                raise RuntimeError(f"run_configs3_share exited with {code}: {json.dumps(out)[:400]}")
            return out

        side("configs3_share_125000x9000_fast", configs3_share)
        # the same in RSCM_MODE_EXACT: OceanCarbon's literal O(T^2) history convolution in the reference's summation order
        side("configs3_share_125000x9000_exact", lambda: configs3_share("--exact"))

        # SURVEY 8d asks for the end-to-end figure beside the resident one: host parameters in,
        # run, full Ts and Td series out into page-locked buffers (never reported as `value`)
        side("end_to_end_1e5", lambda: end_to_end_extra(args.members, local_rank, mode, stream, years))

        # BASELINE.json configs[4]: the calibration loop, 1e5 walkers per iteration, stretch move
        # and likelihood on the device (rscm_sampler_*)
        side("calibrate_device_1e5", lambda: calibration_extra(local_rank))
        # ... and with a graph of linked ensembles as the evaluator (rscm_sampler_create_graph)
        side("calibrate_graph_device_1e5", lambda: graph_calibration_extra(local_rank))
        side("calibrate_graph_device_1e5_fast", lambda: graph_calibration_extra(local_rank, mode=1))

    emit()
    if dist.is_available() and dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception as exc:  # noqa: BLE001 -- the line is out; a peer that has already gone must not turn this rank's exit code
            print(f"bench.py: rank {rank}: destroy_process_group: {type(exc).__name__}: {exc}", file=sys.stderr)


if __name__ == "__main__":
    main()
