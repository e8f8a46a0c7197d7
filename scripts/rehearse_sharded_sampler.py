"""The sharded device stretch-move sampler (BASELINE.json configs[4]: rscm_sampler_create_sharded,
calibrate.DeviceEnsembleSampler under torch.distributed) rehearsed on a one-GPU box: R ranks share GPU 0,
collectives over gloo.  Every rank runs the sampler sharded over the ranks and, for comparison, the whole
ensemble on its own (shard=False): positions, log probabilities and acceptance counters of every kept
sweep must agree bit for bit -- the chain does not depend on the number of ranks.

    RSCM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29521 scripts/rehearse_sharded_sampler.py --out gpurun_out/sharded_sampler
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

T0, T1 = 1750, 2500
LOW = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0])
HIGH = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])
NAMES = ["lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep"]


def f_syn(t):
    return 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2.0 * np.pi * (t - 1750.0) / 11.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walkers", type=int, nargs="+", default=[4096, 100_000])
    ap.add_argument("--sweeps", type=int, default=4)
    ap.add_argument("--graph", action="store_true", help="a graph of linked ensembles (CarbonCycle -> CO2ERF -> Sum -> TwoLayer) as the "
                                                         "evaluator (rscm_sampler_create_graph) instead of one two-layer ensemble")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sharded_sampler"))
    args = ap.parse_args()
    import torch.distributed as dist
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.two_layer import TwoLayerBuilder

    dist.init_process_group(os.environ.get("RSCM_BENCH_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    device = int(os.environ.get("RSCM_BENCH_DEVICE", "0"))
    t = np.arange(T0, T1 + 1, dtype=np.float64)
    axis = core.TimeAxis.from_values(t)
    defaults = dict(lambda0=1.0, a=0.0, efficacy=1.0, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (core.ModelBuilder().with_device(device).with_time_axis(axis)
         .with_rust_component(TwoLayerBuilder.from_parameters(defaults).build())
         .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(f_syn(t), axis, "W/m^2", core.InterpolationStrategy.Linear))
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    if args.graph:
        from rscm_amd.components import CarbonCycleBuilder, CO2ERFBuilder
        tg = np.arange(1750.0, 1951.0)
        axis = core.TimeAxis.from_values(tg)
        schema = core.VariableSchema()
        for n in ["Emissions|CO2|Anthropogenic", "Surface Temperature", "Deep Ocean Temperature", "Atmospheric Concentration|CO2",
                  "Cumulative Land Uptake", "Cumulative Emissions|CO2", "Effective Radiative Forcing|CO2", "Effective Radiative Forcing|Other"]:
            schema.add_variable(n, "")
        schema.add_aggregate("Effective Radiative Forcing", "W/m^2", "Sum", ["Effective Radiative Forcing|CO2", "Effective Radiative Forcing|Other"])
        emis = np.interp(tg, [1750.0, 1850.0, 1950.0], [1.0, 1.5, 4.0])
        b = (core.ModelBuilder().with_device(device).with_time_axis(axis).with_schema(schema)
             .with_rust_component(CarbonCycleBuilder.from_parameters(dict(tau=25.0, conc_pi=278.0, alpha_temperature=0.05)).build())
             .with_rust_component(CO2ERFBuilder.from_parameters(dict(erf_2xco2=3.7, conc_pi=278.0)).build())
             .with_rust_component(TwoLayerBuilder.from_parameters(dict(defaults, lambda0=1.1, efficacy=1.2)).build())
             .with_exogenous_variable("Emissions|CO2|Anthropogenic", core.Timeseries(emis, axis, "", core.InterpolationStrategy.Linear))
             .with_exogenous_variable("Effective Radiative Forcing|Other", core.Timeseries(0.2 * np.sin(tg / 9.0), axis, "", core.InterpolationStrategy.Linear))
             .with_initial_values({"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0, "Atmospheric Concentration|CO2": 278.0,
                                   "Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
        names = ["TwoLayer.lambda0", "tau"]
        runner = cal.ModelRunner(b, names, ["Surface Temperature", "Atmospheric Concentration|CO2"])
        assert runner._graph
        truth = runner.run([1.25, 30.0])
        target = cal.Target()
        for yr in range(1800, 1941, 10):
            target.add_observation("Surface Temperature", float(yr), truth["Surface Temperature"][float(yr)], 0.005)
            target.add_observation("Atmospheric Concentration|CO2", float(yr), truth["Atmospheric Concentration|CO2"][float(yr)], 0.1)
        params = cal.ParameterSet().add("TwoLayer.lambda0", cal.Uniform(0.8, 1.6)).add("tau", cal.Uniform(15.0, 45.0))
    else:
        runner = cal.ModelRunner(b, NAMES, ["Surface Temperature"])
        truth = runner.run([defaults[k] for k in NAMES])["Surface Temperature"]
        target = cal.Target()
        for yr in range(1850, 2021, 10):
            target.add_observation("Surface Temperature", float(yr), truth[float(yr)], 0.1)
        params = cal.ParameterSet()
        for k, lo, hi in zip(NAMES, LOW, HIGH):
            params.add(k, cal.Uniform(float(lo), float(hi)))
    dev = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    cases = []
    for W in args.walkers:
        pos = params.sample_random(W, np.random.default_rng(2026 + W))
        t0 = time.perf_counter()
        sharded = dev.run(args.sweeps, cal.WalkerInit.explicit(pos), n_walkers=W, seed=17)
        wall_sharded = time.perf_counter() - t0
        acc_s, prop_s = dev.n_accepted.copy(), dev.n_proposed.copy()
        single = dev.run(args.sweeps, cal.WalkerInit.explicit(pos), n_walkers=W, seed=17, shard=False)
        acc_1, prop_1 = dev.n_accepted.copy(), dev.n_proposed.copy()
        xs, x1 = np.stack(sharded._samples), np.stack(single._samples)
        ls, l1 = np.stack(sharded._log_probs), np.stack(single._log_probs)
        moved = float((xs[-1] != pos).any(axis=1).mean())
        cases.append({"walkers": W, "sweeps": args.sweeps,
                      "positions_bit_equal": bool(np.array_equal(xs.view(np.uint64), x1.view(np.uint64))),
                      "log_probs_bit_equal": bool(np.array_equal(ls.view(np.uint64), l1.view(np.uint64))),
                      "counters_equal": bool(np.array_equal(acc_s, acc_1) and np.array_equal(prop_s, prop_1)),
                      "every_walker_proposed_each_sweep": bool((prop_s == args.sweeps).all()),
                      "fraction_of_walkers_moved": moved,
                      "acceptance_rate": float(acc_s.sum() / prop_s.sum()),
                      "ms_per_sweep_sharded_incl_exchange": wall_sharded / args.sweeps * 1e3})
    runner.close()
    ok = bool(all(c["positions_bit_equal"] and c["log_probs_bit_equal"] and c["counters_equal"]
                  and c["every_walker_proposed_each_sweep"] and 0.0 < c["fraction_of_walkers_moved"] < 1.0 for c in cases))
    res = {"rank": rank, "world": world, "backend": dist.get_backend(), "ok": ok, "evaluator": "graph" if args.graph else "two-layer ensemble", "cases": cases}
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
