#!/usr/bin/env python3
"""BASELINE.json configs[4]: the rscm-calibrate loop with a 1e5-member ensemble per iteration.

One stretch-move iteration of `walkers` walkers = two half-ensemble evaluations; every evaluation
is one fused run+likelihood launch per GPU (no series written) followed by an all-gather of the
per-member log-likelihood over RCCL (8 B per member).  All ranks run the same sampler with the
same seed, so proposals are identical everywhere and nothing is scattered.

    python scripts/bench_calibrate.py --walkers 100000 --iterations 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bench_calibrate.py
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walkers", type=int, default=100_000)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--device-sampler", action="store_true",
                    help="keep proposals, priors and the accept step on the GPU (rscm_sampler_*); single rank")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.two_layer import TwoLayerBuilder

    t = np.arange(1750.0, 2501.0)
    F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2 * np.pi * (t - 1750.0) / 11.0)
    fixed = dict(lambda0=1.1, a=0.05, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (core.ModelBuilder().with_device(local_rank).with_time_axis(core.TimeAxis.from_values(t))
         .with_rust_component(TwoLayerBuilder.from_parameters(fixed).build())
         .with_exogenous_variable("Effective Radiative Forcing",
                                  core.Timeseries(F, core.TimeAxis.from_values(t), "W/m^2", core.InterpolationStrategy.Linear))
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    names = ["lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep"]
    runner = cal.ModelRunner(b, names, ["Surface Temperature"])
    truth = runner.run([fixed[k] for k in names])["Surface Temperature"]
    target = cal.Target()
    for yr in range(1850, 2021, 10):  # Ts observations 1850..2020 step 10, sigma 0.1 K
        target.add_observation("Surface Temperature", float(yr), truth[float(yr)], 0.1)
    params = cal.ParameterSet()
    for k, (lo, hi) in zip(names, [(0.8, 1.5), (0.0, 0.1), (1.0, 1.8), (0.5, 1.0), (5.0, 15.0), (50.0, 200.0)]):
        params.add(k, cal.Uniform(lo, hi))
    if args.device_sampler:
        # one independent ensemble of walkers per GPU (seed = rank): nothing is exchanged while
        # sampling, the chains are pooled afterwards
        sampler = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    else:
        sampler = cal.EnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    rng = np.random.default_rng(20260327)  # same on every rank
    sampler.run(2, cal.WalkerInit.from_prior(), n_walkers=args.walkers, rng=rng)  # warm-up
    start = cal.WalkerInit.explicit(params.sample_random(args.walkers, rng))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kw = dict(thin=args.iterations, seed=rank) if args.device_sampler else {}  # device: fetch the first sweep only
    chain = sampler.run(args.iterations, start, n_walkers=args.walkers, rng=rng, **kw)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        evals = args.walkers * (args.iterations + 1) * (world if args.device_sampler else 1)
        print(json.dumps({
            "metric": "calibration loop, model evaluations/s (751-point two-layer runs incl. likelihood)",
            "value": evals / dt, "unit": "member-runs/s", "n_gpus": world, "walkers": args.walkers,
            "iterations": args.iterations, "s_per_iteration": dt / args.iterations,
            "member_years_per_s": evals * 750 / dt, "acceptance_rate": sampler.acceptance_rate(),
            "sampler": "device" if args.device_sampler else "host",
            "device_ms_per_iteration": (sampler.device_ms / args.iterations) if args.device_sampler else None,
            "mean_log_prob_last": float(chain.flat_log_probs(len(chain) - 1).mean())}))
    runner.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
