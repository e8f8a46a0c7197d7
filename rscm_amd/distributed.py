"""Member sharding across the GPUs of one node.

Members are independent (each ``run`` builds its own model,
crates/rscm-calibrate/src/model_runner.rs:257-266), so the ensemble shards embarrassingly:
rank ``g`` of ``G`` owns the contiguous global members ``[offset, offset+count)``, keeps its SoA
buffers in its own HBM and runs the same kernels.  There is NO data-path collective.  The only
exchanges, over ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo``
in the CPU tests), are:

* ``gather_members``: all-gather of one small per-member vector (log-likelihood, status) --
  8 B per member, e.g. 8 MB for 1e6 members;
* ``reduce_summary``: all-reduce of count/sum/min/max.

Full time series are never gathered: 12 GB into one GPU's seven xGMI links would serialise on
rank 0 for no benefit; each rank copies its own shard to the host if asked.
Parameters need no scatter either: ``Ensemble.sample_lhs`` is counter-based on the global
member id, so every rank generates its own rows of one global Latin hypercube.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(offset, count) of rank's contiguous member block; blocks differ by at most one member."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad rank/world/n_total")
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _dist():
    import torch.distributed as dist
    return dist


def is_distributed() -> bool:
    """An initialised process group of more than one rank -- or of one rank when
    ``RSCM_FORCE_DISTRIBUTED=1`` (rehearsal of the collective paths, e.g. RCCL on a one-GPU box)."""
    try:
        d = _dist()
        if not (d.is_available() and d.is_initialized()):
            return False
        return d.get_world_size() > 1 or os.environ.get("RSCM_FORCE_DISTRIBUTED") == "1"
    except Exception:
        return False


def _device_for_backend():
    import torch
    d = _dist()
    if d.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_members(local, n_total: int, group=None) -> np.ndarray:
    """All-gather a per-member vector sharded by ``shard_bounds`` into the global order.
    Every rank returns the full ``[n_total]`` array on the host.

    ``local`` is a numpy array or a ``DeviceVector`` (``Ensemble.loglik(..., on_device=True)``,
    ``status_device()``).  With the ``nccl`` backend a device vector is gathered where it lies: one
    device-to-device copy into the padded send buffer, the RCCL all-gather, one copy of the gathered
    vector to the host -- no host round trip of the shard.  With ``gloo`` (CPU rehearsals) the shard
    comes to the host first."""
    from .ensemble import DeviceVector
    on_device = isinstance(local, DeviceVector)
    if not is_distributed():
        host = local.to_host() if on_device else np.ascontiguousarray(local).copy()
        if len(host) != n_total:
            raise ValueError("single process: local shard must be the whole ensemble")
        return host
    import torch
    d = _dist()
    world, rank = d.get_world_size(group), d.get_rank(group)
    off, cnt = shard_bounds(n_total, rank, world)
    if len(local) != cnt:
        raise ValueError(f"rank {rank}: shard has {len(local)} members, expected {cnt}")
    dev = _device_for_backend()
    max_cnt = shard_bounds(n_total, 0, world)[1]
    if on_device and dev.type == "cuda":
        src = torch.as_tensor(local, device=dev)            # zero-copy view of the library's buffer
        mine = torch.zeros(max_cnt, dtype=src.dtype, device=dev)
        mine[:cnt].copy_(src)
    else:
        host = local.to_host() if on_device else np.ascontiguousarray(local)
        pad = np.zeros(max_cnt, dtype=host.dtype)
        pad[:cnt] = host
        mine = torch.from_numpy(pad).to(dev)
    gathered = torch.empty(world * max_cnt, dtype=mine.dtype, device=dev)
    d.all_gather_into_tensor(gathered, mine, group=group)
    parts = gathered.cpu().numpy().reshape(world, max_cnt)
    out = np.empty(n_total, dtype=parts.dtype)
    for r in range(world):
        o, c = shard_bounds(n_total, r, world)
        out[o:o + c] = parts[r, :c]
    return out


def reduce_summary(local: Dict[str, float], group=None) -> Dict[str, float]:
    """Combine ``Ensemble.summary`` dicts (count, mean, min, max) over ranks."""
    cnt = float(local["count"])
    s = local["mean"] * cnt if cnt else 0.0
    if not is_distributed():
        return dict(local)
    import torch
    d = _dist()
    dev = _device_for_backend()
    acc = torch.tensor([cnt, s], dtype=torch.float64, device=dev)
    mn = torch.tensor([local["min"]], dtype=torch.float64, device=dev)
    mx = torch.tensor([local["max"]], dtype=torch.float64, device=dev)
    d.all_reduce(acc, op=d.ReduceOp.SUM, group=group)
    d.all_reduce(mn, op=d.ReduceOp.MIN, group=group)
    d.all_reduce(mx, op=d.ReduceOp.MAX, group=group)
    total = float(acc[0].item())
    return {"count": int(total), "mean": float(acc[1].item()) / total if total else float("nan"),
            "min": float(mn.item()), "max": float(mx.item())}


class ShardedEnsemble:
    """One global ensemble of ``n_total`` members, this rank holding its block on its GPU.

    ``factory(count, device)`` must return a configured ``rscm_amd.Ensemble`` of ``count`` members
    (forcing and initial values set); parameters are then drawn on the device from one global
    Latin hypercube, or set from the rank's slice of a global matrix.
    """

    def __init__(self, n_total: int, factory, rank: Optional[int] = None,
                 world: Optional[int] = None, device: Optional[int] = None):
        r, lr, w = env_rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.n_total = int(n_total)
        self.offset, self.count = shard_bounds(self.n_total, self.rank, self.world)
        self.ensemble = factory(self.count, lr if device is None else device)

    def sample_lhs(self, seed: int, low, high) -> None:
        self.ensemble.sample_lhs(seed, low, high, self.offset, self.n_total)

    def set_params_global(self, soa: np.ndarray) -> None:
        self.ensemble.set_params(np.ascontiguousarray(soa[:, self.offset:self.offset + self.count]))

    def run(self) -> None:
        self.ensemble.rewind()
        self.ensemble.run()

    def loglik_global(self, obs_var, obs_tidx, obs_value, obs_sigma, normalize=False) -> np.ndarray:
        local = self.ensemble.loglik(obs_var, obs_tidx, obs_value, obs_sigma, normalize, on_device=True)
        return gather_members(local, self.n_total)

    def status_global(self) -> np.ndarray:
        return gather_members(self.ensemble.status_device(), self.n_total)

    def params_global(self) -> np.ndarray:
        """The global ``[P][n_total]`` parameter matrix (one gather per row)."""
        P = self.ensemble.get_params()
        return np.stack([gather_members(np.ascontiguousarray(row), self.n_total) for row in P])

    def summary_global(self, var, tidx: int) -> Dict[str, float]:
        return reduce_summary(self.ensemble.summary(var, tidx))
