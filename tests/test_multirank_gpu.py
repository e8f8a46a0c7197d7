"""The N > 1 path with REAL ensembles: several ranks share the one GPU of the box, collectives over gloo.

These tests start `torch.distributed.run` children, so they live under their own marker (`gpu_ranks`)
and are run as their own pytest process, which itself never touches the GPU:

    python -m pytest tests/test_multirank_gpu.py -m gpu_ranks -q

(`-m gpu` does not select them: that tier runs in one process that has initialised the GPU long before it
would get here, and a GPU process must not exec other programs on this pool.)  Without a GPU they skip.
What they assert is computed inside the workers (scripts/rehearse_*.py): sharded == single-process, bit
for bit."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu_ranks


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


@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_sharded_ensemble_equals_single_process(tmp_path):
    for res in _launch("rehearse_two_ranks.py", 2, 29541, tmp_path, ["--members", "30001"]):
        assert res["world"] == 2 and res["ok"], res
        c = res["checks"]
        assert c["lhs_params_bit_equal"] and c["status_bit_equal"] and c["loglik_bit_equal"] and c["calibrate_batch_bit_equal"]


@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
@pytest.mark.parametrize("ranks", [2, 4])
def test_sharded_sampler_reproduces_the_single_rank_chain(tmp_path, ranks):
    for res in _launch("rehearse_sharded_sampler.py", ranks, 29543 + ranks, tmp_path, ["--walkers", "4096", "20000", "--sweeps", "3"]):
        assert res["world"] == ranks and res["ok"], res
        for case in res["cases"]:
            assert case["positions_bit_equal"] and case["log_probs_bit_equal"] and case["counters_equal"]


@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_sharded_graph_sampler_reproduces_the_single_rank_chain(tmp_path):
    """rscm_sampler_create_graph with n_ranks = 2: a graph of four linked ensembles as the evaluator, the walkers split over the
    ranks -- the single-rank chain, bit for bit."""
    for res in _launch("rehearse_sharded_sampler.py", 2, 29551, tmp_path, ["--graph", "--walkers", "2048", "--sweeps", "3"]):
        assert res["world"] == 2 and res["ok"] and res["evaluator"] == "graph", res
        for case in res["cases"]:
            assert case["positions_bit_equal"] and case["log_probs_bit_equal"] and case["counters_equal"]
