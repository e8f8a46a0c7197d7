"""The RCCL (backend "nccl") code paths on a one-GPU box: a ONE-rank process group, treated as distributed
(RSCM_FORCE_DISTRIBUTED=1).  RCCL refuses two ranks on one device, so this is as close as a single GPU gets
to the 8-GPU run: communicator creation, device-resident all-gathers (zero-copy torch views of the library's
buffers through __cuda_array_interface__), the sharded sampler's exchange on device memory, bench.py's
barrier / all-reduce.  Checks: the results equal the plain single-process ones bit for bit.

    RSCM_FORCE_DISTRIBUTED=1 python scripts/rehearse_rccl_one_rank.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29537")
os.environ["RSCM_FORCE_DISTRIBUTED"] = "1"


def main():
    import torch
    import torch.distributed as dist
    import rscm_amd
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.distributed import ShardedEnsemble, is_distributed
    from rscm_amd.two_layer import TwoLayerBuilder
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert is_distributed() and dist.get_backend() == "nccl"
    t = np.arange(1750.0, 2501.0)
    b = np.append(t, t[-1] + 1.0)
    F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2.0 * np.pi * (t - 1750.0) / 11.0)
    lo = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0])
    hi = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])

    def factory(count, _dev):
        e = rscm_amd.Ensemble(rscm_amd.KIND_TWO_LAYER, count, b, device=0)
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        return e

    n = 50_000
    obs_t = np.arange(100, 271, 10, dtype=np.int32)
    obs_v = np.ones(len(obs_t), dtype=np.int32)
    obs_val, obs_sig = 0.8 + 0.004 * obs_t, np.full(len(obs_t), 0.1)
    se = ShardedEnsemble(n, factory, rank=0, world=1, device=0)
    se.sample_lhs(20260327, lo, hi)
    se.run()
    ll = se.loglik_global(obs_v, obs_t, obs_val, obs_sig)          # DeviceVector -> torch view -> RCCL all-gather
    st = se.status_global()
    sm = se.summary_global(1, 270)
    want_ll = se.ensemble.loglik(obs_v, obs_t, obs_val, obs_sig)
    checks = {"loglik_bit_equal": bool(np.array_equal(ll.view(np.uint64), want_ll.view(np.uint64))),
              "status_equal": bool(np.array_equal(st, se.ensemble.status())),
              "summary_equal": bool(sm == se.ensemble.summary(1, 270))}
    se.ensemble.close()
    # the sharded sampler's device-side exchange
    axis = core.TimeAxis.from_values(t)
    names = ["lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep"]
    defaults = dict(lambda0=1.0, a=0.0, efficacy=1.0, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    bld = (core.ModelBuilder().with_time_axis(axis).with_rust_component(TwoLayerBuilder.from_parameters(defaults).build())
           .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(F, axis, "W/m^2", core.InterpolationStrategy.Linear))
           .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    runner = cal.ModelRunner(bld, names, ["Surface Temperature"])
    truth = runner.run([defaults[k] for k in names])["Surface Temperature"]
    target = cal.Target()
    for yr in range(1850, 2021, 10):
        target.add_observation("Surface Temperature", float(yr), truth[float(yr)], 0.1)
    params = cal.ParameterSet()
    for k, a_, b_ in zip(names, lo, hi):
        params.add(k, cal.Uniform(float(a_), float(b_)))
    dev = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    W = 20_000
    pos = params.sample_random(W, np.random.default_rng(3))
    sharded = dev.run(3, cal.WalkerInit.explicit(pos), n_walkers=W, seed=9)              # driven mode, exchange on the device
    acc_s = dev.n_accepted.copy()
    plain = dev.run(3, cal.WalkerInit.explicit(pos), n_walkers=W, seed=9, shard=False)
    checks["sampler_positions_bit_equal"] = bool(np.array_equal(np.stack(sharded._samples).view(np.uint64), np.stack(plain._samples).view(np.uint64)))
    checks["sampler_log_probs_bit_equal"] = bool(np.array_equal(np.stack(sharded._log_probs).view(np.uint64), np.stack(plain._log_probs).view(np.uint64)))
    checks["sampler_counters_equal"] = bool(np.array_equal(acc_s, dev.n_accepted))
    # ModelRunner.log_likelihood_batch: shard (the whole batch here) + device all-gather
    batch = lo + np.random.default_rng(5).random((10_001, 6)) * (hi - lo)
    got = runner.log_likelihood_batch(batch, target, cal.GaussianLikelihood())
    os.environ["RSCM_FORCE_DISTRIBUTED"] = "0"
    want = runner.log_likelihood_batch(batch, target, cal.GaussianLikelihood())
    checks["calibrate_batch_bit_equal"] = bool(np.array_equal(got.view(np.uint64), want.view(np.uint64)))
    runner.close()
    ok = all(checks.values())
    print(json.dumps({"backend": "nccl", "world": 1, "ok": ok, "checks": checks}))
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
