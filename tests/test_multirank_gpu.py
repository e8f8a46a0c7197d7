"""The N > 1 path with REAL ensembles: several ranks share the one GPU of the box, collectives over gloo.

These are in the driver's tier (`-m gpu`): sharded ensemble == single process with real kernels on two ranks, the sharded
sampler (plain and with a graph as the evaluator) == the single-rank chain bit for bit on two ranks, and bench.py started as
`python bench.py --gpus 2` (its own launcher) printing a valid, compact line.  They start their ranks as
CHILD processes (subprocess), exactly as tests/test_gpu_parity.py::test_external_stream_and_async_run starts its child
from the same pytest process -- what the pool forbids is replacing a GPU process by exec, not starting children -- and they
keep the number of processes on the card at three (pytest + two ranks; the pool allows six).

The wider rehearsal (four ranks) stays under its own marker and its own pytest process:

    python -m pytest tests/test_multirank_gpu.py -m gpu_ranks -q

Without a GPU everything here skips.  What is asserted is computed inside the workers (scripts/rehearse_*.py): sharded ==
single-process, bit for bit."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus() -> int:
    import torch
    return torch.cuda.device_count()   # does not initialise the GPU


def _launch(script, ranks, port, out, extra=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               RSCM_BENCH_BACKEND="gloo", RSCM_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "scripts", script), "--out", str(out), *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [json.load(open(os.path.join(out, f"rank{k}.json"))) for k in range(ranks)]


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_sharded_ensemble_equals_single_process(tmp_path):
    for res in _launch("rehearse_two_ranks.py", 2, 29541, tmp_path, ["--members", "30001"]):
        assert res["world"] == 2 and res["ok"], res
        c = res["checks"]
        assert c["lhs_params_bit_equal"] and c["status_bit_equal"] and c["loglik_bit_equal"] and c["calibrate_batch_bit_equal"]


@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
@pytest.mark.parametrize("ranks", [pytest.param(2, marks=[pytest.mark.gpu, pytest.mark.gpu_ranks]), pytest.param(4, marks=pytest.mark.gpu_ranks)])
def test_sharded_sampler_reproduces_the_single_rank_chain(tmp_path, ranks):
    """rscm_sampler_create_sharded: the walkers split over the ranks, each half-step's blocks all-gathered -- the single-rank chain,
    bit for bit (positions, log-probabilities, acceptance counters).  The 2-rank case is in the driver's tier (pytest + two ranks =
    three processes on the card); 4 ranks stay under `gpu_ranks` only.  Reference: crates/rscm-calibrate/src/sampler/ensemble.rs:143-177,
    496-547 (the half-ensemble evaluation the ranks split)."""
    for res in _launch("rehearse_sharded_sampler.py", ranks, 29543 + ranks, tmp_path, ["--walkers", "4096", "20000", "--sweeps", "3"]):
        assert res["world"] == ranks and res["ok"], res
        for case in res["cases"]:
            assert case["positions_bit_equal"] and case["log_probs_bit_equal"] and case["counters_equal"]


@pytest.mark.gpu
@pytest.mark.gpu_ranks
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_sharded_graph_sampler_reproduces_the_single_rank_chain(tmp_path):
    """rscm_sampler_create_graph with n_ranks = 2: a graph of four linked ensembles as the evaluator, the walkers split over the
    ranks -- the single-rank chain, bit for bit."""
    for res in _launch("rehearse_sharded_sampler.py", 2, 29551, tmp_path, ["--graph", "--walkers", "2048", "--sweeps", "3"]):
        assert res["world"] == 2 and res["ok"] and res["evaluator"] == "graph", res
        for case in res["cases"]:
            assert case["positions_bit_equal"] and case["log_probs_bit_equal"] and case["counters_equal"]


def _bench_line(argv, env_extra, launcher=False, expect_code=0):
    """(line, details): the ONE compact line bench.py prints -- under 8000 bytes, the driver keeps 8 KB of stdout -- and the full
    record it wrote to --details."""
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29561"]
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "bench_details.json")
        cmd += [os.path.join(ROOT, "bench.py"), *argv, "--details", path]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == expect_code, r.stdout[-3000:] + r.stderr[-3000:]
        rows = [x for x in r.stdout.splitlines() if x.lstrip().startswith("{")]
        assert len(rows) == 1, r.stdout[-3000:]
        assert len(rows[0].encode()) < 8000, len(rows[0])
        line = json.loads(rows[0])
        assert line["details"] == path
        with open(path) as f:
            details = json.load(f)
    assert details["value"] == pytest.approx(line["value"], rel=1e-5) and set(details["extra"]) == set(line["extra"])
    return line, details


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_bench_two_ranks_with_and_without_a_launcher():
    """The three ways the driver may start the bench -- `python bench.py --gpus 1`, `python bench.py --gpus 2` (bench.py starts its
    ranks itself) and under torch.distributed.run -- each print ONE valid line; with two ranks the whole-job value is about twice
    one rank's share of it (both ranks on this box's one GPU: they take turns on the card), and every rank's own kernel time is in
    the line.  Collectives over gloo here; on the driver's 8-GPU node the same code path runs RCCL."""
    small = ["--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--members", "20000"]
    one, full = _bench_line(["--gpus", "1", *small], {})
    assert one["n_gpus"] == 1 and one["value"] > 0 and one["per_rank"]["kernel_ms"][0] > 0
    assert one["roofline"]["frac"] > 0 and one["roofline"]["kernel_ms"] > 0 and one["cpu_baseline"] is None
    # (a few per cent of the Latin hypercube's members run away -- lambda0 - a Ts < 0 -- and are flagged: part of the workload)
    assert one["collective"]["world"] == 1 and full["check"]["failed_members_rank0"] < 0.1 * 20000
    knobs = {"RSCM_BENCH_BACKEND": "gloo", "RSCM_BENCH_DEVICE": "0"}
    for launcher in (False, True):
        two, full = _bench_line(["--gpus", "2", *small], knobs, launcher=launcher)
        assert two["n_gpus"] == 2 and two["value"] > 0 and two["scaling"] == "weak"
        assert two["collective"]["world"] == 2 and two["collective"]["ranks_seen"] == 2 and two["collective"]["loss_gather_ms"] > 0
        assert len(two["per_rank"]["kernel_ms"]) == 2 and min(two["per_rank"]["kernel_ms"]) > 0
        assert two["config"]["members_per_gpu"] == 20000 and two["config"]["failed_members"] == full["config"]["failed_members"]
        two = full   # the facts below live in the details
        g = two["collective"]["loss_gather"]   # configs[4]'s exchange: per-member losses scored on the device, all-gathered over the ranks
        assert "error" not in g and g["members_gathered"] == 40000 and g["finite"] > 0.9 * 40000 and g["ms"] > 0
        assert two["check"]["failed_members_rank0"] < 0.1 * 20000
        assert len(two["config"]["failed_members_per_rank"]) == 2 and two["config"]["failed_members"] == sum(two["config"]["failed_members_per_rank"])


SCALE_KEYS = ("scale_exact_1e6", "scale_coupled_fast_1e6", "scale_configs3_share", "scale_calibrate_sharded_1e5",
              "scale_calibrate_sharded_1e5_per_gpu")
SCALE_SMALL = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--members", "20000", "--scale-members", "30000",
               "--share-members", "1024", "--share-years", "2", "--scale-walkers", "4096", "--scale-sweeps", "3"]


def _check_scale_extras(line, details, ranks):
    for key in SCALE_KEYS:   # the line: numbers only
        e = line["extra"][key]
        assert "error" not in e and e["rate"] > 0 and (e.get("ms") or e.get("s")) > 0, (key, e)
        assert all(isinstance(v, (int, float)) for v in e.values()), (key, e)
        if ranks > 1 and key != "scale_calibrate_sharded_1e5":
            assert 0 < e["weak_efficiency"] < 1.5, (key, e)
    extra = details["extra"]
    for key in SCALE_KEYS:
        assert key in extra, sorted(extra)
        e = extra[key]
        assert "error" not in e, (key, e)
        assert e["ranks"] == ranks and e["wall_s"] > 0
        assert len(e["per_rank"]["own_s"]) == ranks and min(e["per_rank"]["own_s"]) > 0
        if ranks > 1:
            assert e["rank0_alone_s"] > 0
    for key in ("scale_exact_1e6", "scale_coupled_fast_1e6"):
        e = extra[key]
        assert e["member_years_per_s"] > 0 and len(e["per_rank"]["kernel_ms"]) == ranks and min(e["per_rank"]["kernel_ms"]) > 0
        assert e["roofline"]["frac"] > 0 and e["members_per_gpu"] == 30000
        if ranks > 1:
            assert 0 < e["weak_efficiency"] < 1.5    # both ranks on one card: about a half
    share = extra["scale_configs3_share"]
    assert share["parity_anchor_all_ranks"] and share["failed_members"] == 0 and len(share["per_rank"]["run_s"]) == ranks
    # the ranks hold different blocks of one draw: their ensemble means differ
    if ranks > 1:
        means = [f["warming_end_K_mean"] for f in share["per_rank"]["facts"]]
        assert means[0] != means[1]
    for key, scaling in (("scale_calibrate_sharded_1e5", "strong"), ("scale_calibrate_sharded_1e5_per_gpu", "weak")):
        c = extra[key]
        assert c["scaling"] == scaling and c["device_ms_per_iteration"] > 0 and 0.05 < c["per_rank"]["facts"][0]["acceptance_rate"] < 0.95
        assert c["walkers"] == 4096 * (ranks if scaling == "weak" else 1)
        if ranks > 1:
            assert c["exchange_ms_per_iteration"] > 0 and 0 < c["exchange_share_of_iteration"] < 1
            assert c["per_rank"]["facts"][0]["exchange_bytes_per_half_step"] > 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_bench_two_ranks_measures_every_multi_gpu_config():
    """`bench.py --gpus 2` runs the configs that are defined on more than one GPU on BOTH ranks, barrier to barrier with the
    max-over-ranks wall time and every rank's own time: 1e6-per-GPU two-layer and coupled FAST (here 30 000), each rank's configs[3]
    share (its own block of one draw, parity anchor on every rank), and the device sampler sharded over the process group (walkers
    split over the ranks, and per GPU) with the exchange's share of an iteration -- and the same keys exist at N = 1."""
    knobs = {"RSCM_BENCH_BACKEND": "gloo", "RSCM_BENCH_DEVICE": "0"}
    two, full2 = _bench_line(["--gpus", "2", *SCALE_SMALL], knobs)
    assert two["n_gpus"] == 2 and two["value"] > 0
    _check_scale_extras(two, full2, 2)
    one, full1 = _bench_line(["--gpus", "1", *SCALE_SMALL, "--scale-only"], {})
    _check_scale_extras(one, full1, 1)


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_bench_watchdog_keeps_the_headline_when_side_measurements_do_not_finish():
    """Once the headline is measured the line is complete; should a side measurement not come back (a collective that hangs on an
    8-GPU node would otherwise take the headline with it), the watchdog prints the line as it stands and ends the process with
    exit code 3 (bench.WATCHDOG_EXIT: line valid, extras unfinished -- never a clean 0).  Here the budget is simply too short for
    the extras."""
    line, _ = _bench_line(["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--members", "20000", "--extras-budget", "1.5"], {},
                          expect_code=3)
    assert line["value"] > 0 and line["n_gpus"] == 1 and "watchdog" in line and "unfinished" in line["watchdog"]
    assert isinstance(line["extra"], dict) and "calibrate_graph_device_1e5_fast" not in line["extra"]
    two, _ = _bench_line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--members", "20000", "--extras-budget", "1.5"],
                         {"RSCM_BENCH_BACKEND": "gloo", "RSCM_BENCH_DEVICE": "0"}, expect_code=3)
    assert two["value"] > 0 and two["n_gpus"] == 2 and "watchdog" in two
