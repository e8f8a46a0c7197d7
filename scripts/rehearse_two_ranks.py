"""Rehearsal of the N > 1 path with REAL ensembles on a one-GPU box: two (or more) ranks started by
torch.distributed.run share GPU 0, the collectives run over gloo on the CPU (RCCL refuses two ranks
on one device).  What it checks, on every rank, against the same work done by one process:

  * ShardedEnsemble.sample_lhs: the ranks' parameter blocks are the rows of ONE global Latin
    hypercube (bit for bit the single-process draw);
  * status_global / loglik_global (device-resident shard -> all-gather): bit for bit;
  * summary_global: count / min / max exact, the mean to rounding (the partial sums associate
    differently);
  * calibrate.ModelRunner.log_likelihood_batch sharded over the ranks: bit for bit;
  * a ragged split (n_total not divisible by the world size).

Launch (its own gpurun command -- the launcher itself never touches the GPU):
    RSCM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29517 scripts/rehearse_two_ranks.py --out gpurun_out/two_ranks
Rank 0 prints one JSON line and every rank writes <out>/rank<k>.json; exit code 0 iff all checks hold.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

T0, T1 = 1750, 2500
LOW = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0])
HIGH = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])
SEED = 20260327


def f_syn(t):
    return 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2.0 * np.pi * (t - 1750.0) / 11.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=100_001)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "two_ranks"))
    args = ap.parse_args()
    import torch.distributed as dist
    import rscm_amd
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.distributed import ShardedEnsemble, shard_bounds
    from rscm_amd.two_layer import TwoLayerBuilder

    dist.init_process_group(os.environ.get("RSCM_BENCH_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    device = int(os.environ.get("RSCM_BENCH_DEVICE", "0"))
    n_total = args.members
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    bounds = np.append(t, t[-1] + 1.0)
    F = f_syn(t)

    def factory(count, _device):
        e = rscm_amd.Ensemble(rscm_amd.KIND_TWO_LAYER, count, bounds, device=device)
        e.set_forcing(F)
        e.set_initial("Surface Temperature", 0.0)
        e.set_initial("Deep Ocean Temperature", 0.0)
        return e

    obs_t = np.arange(100, 271, 10, dtype=np.int32)
    obs_v = np.ones(len(obs_t), dtype=np.int32)
    obs_val = 0.8 + 0.004 * obs_t
    obs_sig = np.full(len(obs_t), 0.1)
    checks = {}

    # ---- the sharded run
    se = ShardedEnsemble(n_total, factory, device=device)
    assert (se.offset, se.count) == shard_bounds(n_total, rank, world)
    se.sample_lhs(SEED, LOW, HIGH)
    P_sharded = se.params_global()
    se.run()
    st_sharded = se.status_global()
    ll_sharded = se.loglik_global(obs_v, obs_t, obs_val, obs_sig)
    sm_sharded = se.summary_global("Surface Temperature", 270)
    own_rows = se.ensemble.get_series("Surface Temperature", 750, 751)[0]

    # ---- the same work in one process (every rank does it, on the shared GPU)
    one = factory(n_total, device)
    one.sample_lhs(SEED, LOW, HIGH)
    P_one = one.get_params()
    one.run()
    st_one = one.status()
    ll_one = one.loglik(obs_v, obs_t, obs_val, obs_sig)
    sm_one = one.summary("Surface Temperature", 270)
    rows_one = one.get_series("Surface Temperature", 750, 751)[0]
    one.close()

    def same_bits(a, b):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        return bool(a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)))

    checks["lhs_params_bit_equal"] = same_bits(P_sharded, P_one)
    checks["status_bit_equal"] = same_bits(st_sharded, st_one)
    checks["loglik_bit_equal"] = same_bits(ll_sharded, ll_one)
    checks["own_block_series_bit_equal"] = same_bits(own_rows, rows_one[se.offset:se.offset + se.count])
    checks["summary_count_min_max_equal"] = bool(sm_sharded["count"] == sm_one["count"] and sm_sharded["min"] == sm_one["min"]
                                                 and sm_sharded["max"] == sm_one["max"])
    checks["summary_mean_rel_diff"] = float(abs(sm_sharded["mean"] - sm_one["mean"]) / abs(sm_one["mean"]))
    checks["summary_mean_ok"] = bool(checks["summary_mean_rel_diff"] < 1e-13)
    checks["finite_members"] = int(np.isfinite(ll_one).sum())
    se.ensemble.close()

    # ---- the calibration front: ModelRunner.log_likelihood_batch shards the batch over the ranks
    axis = core.TimeAxis.from_values(t)
    fixed = dict(lambda0=1.1, a=0.05, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (core.ModelBuilder().with_device(device).with_time_axis(axis)
         .with_rust_component(TwoLayerBuilder.from_parameters(fixed).build())
         .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(F, axis, "W/m^2", core.InterpolationStrategy.Linear))
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    names = list(fixed)
    runner = cal.ModelRunner(b, names, ["Surface Temperature"])
    target = cal.Target()
    for k, yr in enumerate(range(1850, 2021, 10)):
        target.add_observation("Surface Temperature", float(yr), float(obs_val[k]), 0.1)
    batch = (LOW + np.random.default_rng(5).random((20_001, 6)) * (HIGH - LOW))
    ll_batch = runner.log_likelihood_batch(batch, target, cal.GaussianLikelihood())   # sharded + gathered
    runner.close()
    ref = factory(len(batch), device)
    ref.set_params_aos(batch)
    ll_ref = ref.run_loglik(obs_v, obs_t, obs_val, obs_sig)
    ref.close()
    checks["calibrate_batch_bit_equal"] = same_bits(ll_batch, ll_ref)

    ok = bool(all(v for k, v in checks.items() if isinstance(v, bool)))
    res = {"rank": rank, "world": world, "backend": dist.get_backend(), "n_total": n_total, "shard": [se.offset, se.count],
           "ok": ok, "checks": checks}
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, f"rank{rank}.json"), "w") as f:
        json.dump(res, f, indent=1)
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        res["all_ranks_ok"] = all(flags)
        print(json.dumps(res))
    dist.destroy_process_group()
    sys.exit(0 if all(flags) else 1)


if __name__ == "__main__":
    main()
