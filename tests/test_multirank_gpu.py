"""The N > 1 path with REAL ensembles: several ranks share the one GPU of the box, collectives over gloo.

Two of these tests are in the driver's tier (`-m gpu`): sharded ensemble == single process with real kernels on two ranks,
and bench.py started as `python bench.py --gpus 2` (its own launcher) printing a valid line.  They start their ranks as
CHILD processes (subprocess), exactly as tests/test_gpu_parity.py::test_external_stream_and_async_run starts its child
from the same pytest process -- what the pool forbids is replacing a GPU process by exec, not starting children -- and they
keep the number of processes on the card at three (pytest + two ranks; the pool allows six).

The wider rehearsals (four ranks, the sharded samplers) stay under their own marker and their own pytest process:

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


@pytest.mark.gpu_ranks
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
@pytest.mark.parametrize("ranks", [2, 4])
def test_sharded_sampler_reproduces_the_single_rank_chain(tmp_path, ranks):
    for res in _launch("rehearse_sharded_sampler.py", ranks, 29543 + ranks, tmp_path, ["--walkers", "4096", "20000", "--sweeps", "3"]):
        assert res["world"] == ranks and res["ok"], res
        for case in res["cases"]:
            assert case["positions_bit_equal"] and case["log_probs_bit_equal"] and case["counters_equal"]


@pytest.mark.gpu_ranks
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_sharded_graph_sampler_reproduces_the_single_rank_chain(tmp_path):
    """rscm_sampler_create_graph with n_ranks = 2: a graph of four linked ensembles as the evaluator, the walkers split over the
    ranks -- the single-rank chain, bit for bit."""
    for res in _launch("rehearse_sharded_sampler.py", 2, 29551, tmp_path, ["--graph", "--walkers", "2048", "--sweeps", "3"]):
        assert res["world"] == 2 and res["ok"] and res["evaluator"] == "graph", res
        for case in res["cases"]:
            assert case["positions_bit_equal"] and case["log_probs_bit_equal"] and case["counters_equal"]


def _bench_line(argv, env_extra, launcher=False):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29561"]
    cmd += [os.path.join(ROOT, "bench.py"), *argv]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = [x for x in r.stdout.splitlines() if x.lstrip().startswith("{")]
    assert len(rows) == 1, r.stdout[-3000:]
    return json.loads(rows[0])


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_bench_two_ranks_with_and_without_a_launcher():
    """The three ways the driver may start the bench -- `python bench.py --gpus 1`, `python bench.py --gpus 2` (bench.py starts its
    ranks itself) and under torch.distributed.run -- each print ONE valid line; with two ranks the whole-job value is about twice
    one rank's share of it (both ranks on this box's one GPU: they take turns on the card), and every rank's own kernel time is in
    the line.  Collectives over gloo here; on the driver's 8-GPU node the same code path runs RCCL."""
    small = ["--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--members", "20000"]
    one = _bench_line(["--gpus", "1", *small], {})
    assert one["n_gpus"] == 1 and one["value"] > 0 and one["per_rank"]["kernel_ms"][0] > 0
    # (a few per cent of the Latin hypercube's members run away -- lambda0 - a Ts < 0 -- and are flagged: part of the workload)
    assert one["collective"]["world"] == 1 and one["check"]["failed_members_rank0"] < 0.1 * 20000
    knobs = {"RSCM_BENCH_BACKEND": "gloo", "RSCM_BENCH_DEVICE": "0"}
    for launcher in (False, True):
        two = _bench_line(["--gpus", "2", *small], knobs, launcher=launcher)
        assert two["n_gpus"] == 2 and two["value"] > 0 and two["scaling"] == "weak"
        assert two["collective"]["world"] == 2 and two["collective"]["ranks_seen"] == 2
        assert len(two["per_rank"]["kernel_ms"]) == 2 and min(two["per_rank"]["kernel_ms"]) > 0
        assert two["config"]["members_per_gpu"] == 20000
        g = two["collective"]["loss_gather"]   # configs[4]'s exchange: per-member losses scored on the device, all-gathered over the ranks
        assert "error" not in g and g["members_gathered"] == 40000 and g["finite"] > 0.9 * 40000 and g["ms"] > 0
        assert two["check"]["failed_members_rank0"] < 0.1 * 20000
