"""CPU tier: the ONE line bench.py prints is something the driver can parse.

Round 5's line was 31 365 bytes (profiles/r5_bench_line.json: prose notes inside every roofline object, per-rank facts); the driver
keeps 8 KB of stdout and recorded `parsed: null`.  Since then bench.py measures into a FULL record, writes that record to
--details (bench_details.json) and to stderr, and prints `compact_line(record)`:

    metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data,
    config{workload, members_per_gpu, years, arithmetic_mode, failed_members},
    roofline{bound, binding, achieved, peak, unit, frac, traffic, traffic_from, algorithmic_bytes, kernel, kernel_ms,
             launches_per_pass, effective_clock_ghz},
    roofline_fp64_valu{frac, achieved, peak, unit, issue_utilisation, parallelism_bound{ms, achieved_frac}},
    cpu_baseline{value, unit, cores, threads_used, single_thread_value, kind, sample},
    collective{backend, world, rccl_ranks_seen, ranks_seen[, loss_gather_ms]},
    per_rank{kernel_ms[], weak_efficiency[]},
    extra{label: numbers only}, details[, watchdog]

These tests push round 5's real 31 KB record, and an 8-rank version of it, through that compaction."""
import copy
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (numpy only at import time: no torch, no GPU, no librscm_gpu.so)

TOP = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
       "config", "roofline", "roofline_fp64_valu", "cpu_baseline", "collective", "per_rank", "extra", "details"}
EXTRA_KEYS = {"rate", "ms", "s", "frac", "fp64_frac", "traffic", "weak_efficiency", "speedup", "exchange_share_of_iteration", "failed", "error"}


def _r5():
    with open(os.path.join(ROOT, "profiles", "r5_bench_line.json")) as f:
        return json.load(f)


def _strict(text):
    def refuse(name):
        raise AssertionError(f"non-finite constant {name} in the line")
    return json.loads(text, parse_constant=refuse)


def _eight_ranks(rec):
    """Round 5's N = 1 record reshaped as the --gpus 8 record: every per-rank list eight long, every scale_* extra with eight ranks'
    own times and facts, RCCL named as the backend, a loss gather, no CPU leg (rank 0 takes it at N = 1 only)."""
    rec = copy.deepcopy(rec)
    rec["n_gpus"] = 8
    rec["value"] *= 8
    rec["per_rank"] = {"kernel_ms": [2.30 + 0.01 * k for k in range(8)], "weak_efficiency": [0.99 - 0.001 * k for k in range(8)],
                       "note": "kernel_ms: HIP events around each rank's own K launches / K; ..." * 3}
    rec["collective"] = {"backend": "rccl (torch.distributed 'nccl')", "world": 8, "rccl_ranks_seen": 8, "ranks_seen": 8, "tensors_on": "cuda",
                         "loss_gather": {"what": "per-member log-likelihood scored on the device, all-gathered" * 2, "ms": 3.1, "bytes_per_rank": 800000,
                                         "members_gathered": 800000, "finite": 772000}}
    rec["config"]["failed_members_per_rank"] = [3429] * 8
    rec["cpu_baseline"] = None
    for key, e in rec["extra"].items():
        if not key.startswith("scale_"):
            continue
        e["ranks"] = 8
        e["weak_efficiency"] = 0.97
        e["rank0_alone_s"] = e["wall_s"] * 0.97
        pr = e["per_rank"]
        pr["own_s"] = pr["own_s"] * 8
        pr["facts"] = pr["facts"] * 8
        for k in ("kernel_ms", "run_s"):
            if k in pr:
                pr[k] = pr[k] * 8
    # the N = 1-only extras do not exist at N = 8; a failed one would
    for key in [k for k in rec["extra"] if not k.startswith("scale_")]:
        del rec["extra"][key]
    rec["extra"]["scale_calibrate_sharded_1e5"]["exchange_share_of_iteration"] = 0.18
    rec["extra"]["scale_calibrate_sharded_1e5"]["speedup"] = 5.1
    return rec


def test_round_5s_record_compacts_to_a_line_the_driver_keeps_whole():
    rec = _r5()
    assert len(json.dumps(rec)) > 25_000                       # the record that was lost
    line = bench.compact_line(rec, "bench_details.json")
    text = bench.dumps_line(line)
    assert "\n" not in text and len(text.encode()) <= bench.LINE_LIMIT_BYTES <= 6000, len(text)
    back = _strict(text)
    assert set(back) == TOP
    assert back["metric"] == rec["metric"] and back["unit"] == "member-years/s" and back["dtype"] == "f64" and back["n_gpus"] == 1
    assert back["value"] == pytest.approx(rec["value"], rel=1e-5) and back["ms_per_step"] == pytest.approx(rec["ms_per_step"], rel=1e-5)
    assert back["roofline"]["frac"] == pytest.approx(rec["roofline"]["frac"], rel=1e-5) and back["roofline"]["bound"] == "hbm"
    assert back["roofline"]["achieved"] / back["roofline"]["peak"] == pytest.approx(back["roofline"]["frac"], rel=1e-4)
    assert back["roofline"]["kernel"] == "two_layer_kernel" and back["roofline"]["traffic"] == pytest.approx(1.2067e9)
    assert back["roofline_fp64_valu"]["parallelism_bound"]["ms"] == pytest.approx(2.2129, rel=1e-3)
    assert back["cpu_baseline"]["value"] == pytest.approx(rec["cpu_baseline"]["value"], rel=1e-5)
    assert back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["cores"] == 16 and isinstance(back["cpu_baseline"]["sample"], str)
    assert set(back["config"]) == {"workload", "members_per_gpu", "years", "arithmetic_mode", "failed_members"}
    assert set(back["extra"]) == set(rec["extra"])              # every extra is still there ...
    for key, e in back["extra"].items():                        # ... as numbers only
        assert set(e) <= EXTRA_KEYS and e, (key, e)
        assert all(isinstance(v, (int, float)) for v in e.values()), (key, e)
        assert "rate" in e and ("ms" in e or "s" in e), (key, e)
    assert back["extra"]["exact_1e6"]["frac"] == pytest.approx(rec["extra"]["exact_1e6"]["roofline"]["frac"], rel=1e-5)
    assert back["extra"]["scale_exact_1e6"]["ms"] == pytest.approx(max(rec["extra"]["scale_exact_1e6"]["per_rank"]["kernel_ms"]), rel=1e-5)


def test_the_eight_rank_line_fits_too():
    rec = _eight_ranks(_r5())
    text = bench.dumps_line(bench.compact_line(rec, "bench_details.json"))
    assert len(text.encode()) <= bench.LINE_LIMIT_BYTES_8_RANKS <= 8000, len(text)
    back = _strict(text)
    assert back["n_gpus"] == 8 and len(back["per_rank"]["kernel_ms"]) == 8 and len(back["per_rank"]["weak_efficiency"]) == 8
    assert back["collective"] == {"backend": "rccl (torch.distributed 'nccl')", "world": 8, "rccl_ranks_seen": 8, "ranks_seen": 8, "loss_gather_ms": 3.1}
    assert back["cpu_baseline"] is None
    for key in ("scale_exact_1e6", "scale_coupled_fast_1e6", "scale_configs3_share", "scale_calibrate_sharded_1e5_per_gpu"):
        assert back["extra"][key]["weak_efficiency"] == 0.97
    assert back["extra"]["scale_calibrate_sharded_1e5"]["speedup"] == 5.1
    # even with every N = 1 extra of round 5 kept beside eight ranks' lists (more than bench.py ever prints at N = 8) it fits
    rec2 = _eight_ranks(_r5())
    rec2["extra"].update({k: v for k, v in _r5()["extra"].items() if not k.startswith("scale_")})
    assert len(bench.dumps_line(bench.compact_line(rec2, "bench_details.json")).encode()) <= 8000


def test_non_finite_numbers_and_failed_extras_stay_strict_json():
    rec = _r5()
    rec["roofline"]["traffic"] = float("nan")
    rec["per_rank"]["weak_efficiency"] = [float("inf")]
    rec["extra"]["udeb_1e5"]["kernel_ms"] = float("nan")
    rec["extra"]["fast_1e6"] = {"error": "RuntimeError: " + "x" * 500}
    rec["cpu_baseline"] = {"error": "OSError: oracle/librscm_oracle.so missing" + "y" * 400}
    rec["watchdog"] = "side measurements unfinished after 540 s; present: 3"
    text = bench.dumps_line(bench.compact_line(rec, None))
    back = _strict(text)                                       # json.dumps(allow_nan=False) inside dumps_line, re-read strictly here
    assert back["roofline"]["traffic"] is None and back["per_rank"]["weak_efficiency"] == [None]
    assert back["extra"]["udeb_1e5"]["ms"] is None and len(back["extra"]["fast_1e6"]["error"]) <= 80
    assert len(back["cpu_baseline"]["error"]) <= 120 and "details" not in back and back["watchdog"].startswith("side measurements")
    assert len(text.encode()) <= bench.LINE_LIMIT_BYTES
    # the details file is strict JSON as well
    assert bench._jsonable({"a": [float("nan"), 1.5, {"b": float("-inf")}]}) == {"a": [None, 1.5, {"b": None}]}


def test_details_are_written_beside_the_line(tmp_path, capfd):
    rec = _r5()
    path = str(tmp_path / "bench_details.json")
    assert bench.write_details(rec, path) == path
    assert json.load(open(path))["extra"]["udeb_65536"]["roofline"]["note"].startswith("7 output rows")   # the prose lives here
    assert "bench.py details: {" in capfd.readouterr().err
    # a cwd that cannot be written costs the file, not the line
    assert bench.write_details(rec, str(tmp_path / "no" / "such" / "dir" / "d.json")) is None


def test_cpu_baseline_is_taken_before_any_extra():
    """A watchdog print must still carry `cpu_baseline`: in main() the CPU leg runs right after the headline, before the watchdog
    is started and before the first side measurement."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    cpu, dog, scale, first_side = (main.index('out["cpu_baseline"] = cpu_baseline()'), main.index("start_watchdog(args.extras_budget"),
                                   main.index("scale_extras(args, rank"), main.index('side(label, lambda'))
    assert cpu < dog < scale < first_side


def test_a_hung_side_measurement_ends_with_the_line_and_a_non_zero_exit_code():
    """the watchdog used to end the process with exit code 0, so a GPU hang after the headline read as a clean run.  Now
    every rank's watchdog exits with WATCHDOG_EXIT (3), and the launcher relays rank 0's line and returns 3.  Exercised on the CPU:
    two ranks over gloo whose 'side measurement' never returns."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RSCM_BENCH_SELFTEST_HANG"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--extras-budget", "1.5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == bench.WATCHDOG_EXIT == 3, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(rows) == 1
    line = json.loads(rows[0])
    assert line["n_gpus"] == 2 and "unfinished" in line["watchdog"]
    assert "watchdog ended rank(s) [0, 1]" in r.stderr
