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
