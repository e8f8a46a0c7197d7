"""CPU tier: the N>1 path (member sharding + gathers) with a real 2-rank gloo group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_total", [10, 1001])
def test_two_rank_gloo_sharding(n_total, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + n_total % 97),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n_total), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    results = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert sorted(x["rank"] for x in results) == [0, 1]
    for x in results:
        assert x["world"] == 2
        assert x["ll_ok"] and x["st_ok"] and x["sum_ok"] and x["params_ok"] and x["single_gather"]
        assert x["reduce"] == {"count": 4, "mean": 0.5, "min": -1.0, "max": 1.0}


def _bench(*argv, env_extra=None, strip=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in strip}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus N` with no RANK / WORLD_SIZE in the environment becomes the launcher: N children of itself, rank 0's
    line relayed as the ONLY thing on stdout.  --rendezvous-only: the ranks meet over gloo and run the contract's barrier, the
    max-over-ranks all-reduce and the rank report, with no GPU work (there is none here)."""
    r = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--rendezvous-only")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(rows) == 1, r.stdout
    line = json.loads(rows[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["value"] is None and line["rendezvous_only"]
    assert line["collective"] == {"backend": "gloo", "world": 2, "rccl_ranks_seen": 0, "ranks_seen": 2, "tensors_on": "cpu"}
    assert line["per_rank"]["kernel_ms"] == [10.0, 20.0]            # every rank's own figure reached rank 0, in rank order
    eff = line["per_rank"]["weak_efficiency"]
    assert len(eff) == 2 and eff[0] < eff[1] <= 1.0 + 1e-9          # the rank that waited shows
    # the all-ranks side measurements' procedure (bench.py scale_measure, sleeps for work): barrier to barrier, the slowest rank
    # sets the wall time, rank 0's time alone beside it; a rank that fails -- before the barriers or inside the timed region --
    # turns the extra into a report on every rank instead of hanging the job
    t = line["scale_selftest"]
    h = t["healthy"]
    assert h["ranks"] == 2 and h["units_per_rank"] == 3 and len(h["per_rank"]["own_s"]) == 2
    assert h["per_rank"]["own_s"][0] < h["per_rank"]["own_s"][1] <= h["wall_s"] + 1e-3
    assert 0.2 < h["weak_efficiency"] < 0.9 and h["rank0_alone_s"] < h["wall_s"]     # rank 0 sleeps 5 ms per unit, rank 1 10 ms
    assert [f["units"] for f in h["per_rank"]["facts"]] == [1 + 3 + 3, 1 + 3]        # warm-up + alone (rank 0 only) + timed
    assert "prepare failed on rank(s) 1" in t["prepare_fails"]["error"] and "self-test" in t["prepare_fails"]["error"]
    assert "rank(s) 1" in t["body_fails"]["error"] and "body raised" in t["body_fails"]["error"]


def test_bench_under_a_launcher_keeps_the_launchers_ranks():
    """With RANK / WORLD_SIZE set (torch.distributed.run's environment) bench.py is a rank, not a launcher."""
    cmd_env = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"]
    r = subprocess.run(cmd, env=dict(os.environ, **cmd_env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.lstrip().startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["collective"]["ranks_seen"] == 2


def test_bench_launcher_reports_a_failed_rank():
    """A rank that fails (here: no GPU for the real run) makes the launcher exit non-zero, print no result line and name the rank."""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu-baseline", "--members", "2000",
               env_extra={"RSCM_BENCH_BACKEND": "gloo", "RSCM_BENCH_DEVICE": "0"})   # (on a one-GPU box: both ranks on its GPU)
    if r.returncode == 0:
        pytest.skip("a GPU is present: the real two-rank run succeeded")
    assert r.stdout.strip() == ""
    assert "exited with" in r.stderr


def test_bench_world_size_mismatch_is_refused():
    r = _bench("--gpus", "2", "--rendezvous-only", env_extra={"RANK": "0", "WORLD_SIZE": "1"}, strip=())
    assert r.returncode != 0 and "disagree" in (r.stderr + r.stdout)
