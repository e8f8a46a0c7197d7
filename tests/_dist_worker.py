"""Worker for tests/test_distributed_cpu.py: one rank of a 2-process gloo group.

The compute itself needs a GPU, so here the rank's ensemble is a stand-in with the Ensemble
interface whose outputs are known functions of the GLOBAL member id; what is under test is the
product's N>1 path: shard arithmetic, global-offset plumbing, all-gather ordering with ragged
shards, summary reduction."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rscm_amd.distributed import ShardedEnsemble, gather_members, reduce_summary, shard_bounds  # noqa: E402


class FakeEnsemble:
    def __init__(self, count, device):
        self.n_members = count
        self.device = device
        self.gid = None

    def sample_lhs(self, seed, low, high, member_offset, n_total):
        self.gid = np.arange(member_offset, member_offset + self.n_members)
        self.n_total = n_total

    def set_params(self, soa):
        self.params = soa

    def rewind(self):
        pass

    def run(self):
        pass

    def loglik(self, *a, **k):
        return -0.5 * self.gid.astype(np.float64) ** 2

    def status(self):
        return (self.gid % 7 == 0).astype(np.uint8)

    def status_device(self):  # the stand-in has no device: the gather takes host arrays as well
        return self.status()

    def get_params(self):
        return self.params

    def summary(self, var, tidx):
        x = self.gid.astype(np.float64)
        if len(x) == 0:
            return {"count": 0, "mean": float("nan"), "min": float("inf"), "max": float("-inf")}
        return {"count": len(x), "mean": x.mean(), "min": x.min(), "max": x.max()}


def main():
    n_total = int(sys.argv[1])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    se = ShardedEnsemble(n_total, FakeEnsemble)
    assert (se.offset, se.count) == shard_bounds(n_total, rank, world)
    se.sample_lhs(1, None, None)
    se.run()
    ll = se.loglik_global([], [], [], [])
    st = se.status_global()
    sm = se.summary_global(1, 0)
    glob = np.arange(6 * n_total, dtype=np.float64).reshape(6, n_total)
    se.set_params_global(glob)
    ok = bool(np.array_equal(se.ensemble.params, glob[:, se.offset:se.offset + se.count]))
    gid = np.arange(n_total, dtype=np.float64)
    res = {
        "rank": rank, "world": world,
        "ll_ok": bool(np.array_equal(ll, -0.5 * gid ** 2)),
        "st_ok": bool(np.array_equal(st, (np.arange(n_total) % 7 == 0).astype(np.uint8))),
        "sum_ok": sm["count"] == n_total and abs(sm["mean"] - gid.mean()) < 1e-9
                  and sm["min"] == 0.0 and sm["max"] == n_total - 1,
        "params_ok": ok,
        "single_gather": bool(np.array_equal(gather_members(ll[se.offset:se.offset + se.count], n_total), ll)),
        "reduce": reduce_summary({"count": 2, "mean": float(rank), "min": -rank, "max": rank}),
    }
    # one file per rank: stdout of the two ranks can interleave
    with open(os.path.join(sys.argv[2], f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
