"""Throughput of every ensemble kind on one MI355X (HIP events around rscm_ens_run, best of a few
passes), with the algorithmic HBM bytes per member-year each kind moves.  Run on the GPU box:

    python scripts/bench_kinds.py > gpurun_out/kinds.json
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd  # noqa: E402
from rscm_amd import _lib as L  # noqa: E402

T0, T1 = 1750, 2500
t = np.arange(T0, T1 + 1, dtype=np.float64)
b = np.append(t, t[-1] + 1.0)
T = len(t)
yr = t - T0
years = T - 1
HBM_PEAK = 8000.0  # GB/s


def lhs(e, defaults, ranges, names):
    lo = np.array(defaults, dtype=float)
    hi = lo.copy()
    for k, (a_, b_) in ranges.items():
        j = names.index(k)
        lo[j], hi[j] = a_, b_
    e.sample_lhs(20260327, lo, hi)


def timed(e, passes=3):
    ms = []
    for _ in range(passes):
        e.rewind()
        e.run()
        ms.append(e.last_run_ms())
    return min(ms)


ONLY = set(sys.argv[1:])   # e.g. `python scripts/bench_kinds.py ocean_carbon ocean_carbon_fast`


def case(name, kind, members, defaults, ranges, names, inputs, initial, bytes_per_my, note, mode=0):
    if ONLY and name not in ONLY:
        return None
    with rscm_amd.Ensemble(kind, members, b) as e:
        e.set_mode(mode)
        lhs(e, defaults, ranges, names)
        e.set_forcing(inputs)
        for v, x in initial.items():
            e.set_initial(v, x)
        ms = timed(e)
        assert not e.status().any()
    my = members * years
    return {"members": members, "kernel_ms": round(ms, 3), "member_years_per_s": my / (ms * 1e-3),
            "alg_bytes_per_member_year": bytes_per_my, "hbm_frac": bytes_per_my * my / (ms * 1e-3) / 1e9 / HBM_PEAK,
            "note": note}


out = {}
ramp = lambda a, r: a * r ** yr  # noqa: E731
out["ozone_forcing"] = case("ozone_forcing", L.KIND_OZONE_FORCING, 1_000_000, L.OZ_DEFAULTS, dict(strat_o3_scale=(-0.005, -0.004), trop_radeff=(0.03, 0.04)),
                            L.OZ_PARAM_NAMES, np.stack([1000.0 + yr, 700.0 + 2 * yr, 0.05 * yr, 0.5 * yr, 0.1 * yr, 0.005 * yr]), {}, 24,
                            "one f64 pow + one log per member-year")
out["aerosol_direct"] = case("aerosol_direct", L.KIND_AEROSOL_DIRECT, 1_000_000, L.AD_DEFAULTS, dict(sox_coefficient=(-0.004, -0.003), bc_coefficient=(0.007, 0.009)),
                             L.AD_PARAM_NAMES, np.stack([1 + 0.1 * yr, 2.5 + 0.01 * yr, 10 + 0.03 * yr, 10 + 0.05 * yr]), {}, 32, "no transcendental")
out["aerosol_indirect"] = case("aerosol_indirect", L.KIND_AEROSOL_INDIRECT, 1_000_000, L.AI_DEFAULTS, dict(cloud_albedo_coefficient=(-1.2, -0.8)),
                               L.AI_PARAM_NAMES, np.stack([1 + 0.1 * yr, 10 + 0.03 * yr]), {}, 8, "one log per member-year")
out["ch4_chemistry"] = case("ch4_chemistry", L.KIND_CH4_CHEMISTRY, 1_000_000, L.CH4_DEFAULTS, dict(tau_oh=(8.0, 11.0), ch4_self_feedback=(-0.4, -0.2)),
                            L.CH4_PARAM_NAMES, np.stack([150 + 0.4 * yr, 0.004 * yr, 5 + 0.04 * yr, 200 + 0.5 * yr, 50 + 0.1 * yr]), {1: 722.0}, 16,
                            "4 Prather passes: 4 pow + 1 exp + ~14 divisions per member-year")
out["n2o_chemistry"] = case("n2o_chemistry", L.KIND_N2O_CHEMISTRY, 1_000_000, L.N2O_DEFAULTS, dict(tau_n2o=(110.0, 160.0), lifetime_feedback=(-0.08, -0.01)),
                            L.N2O_PARAM_NAMES, (0.01 * yr)[None], {1: 270.0}, 16, "4 passes: 4 pow + ~10 divisions per member-year")
out["co2_budget"] = case("co2_budget", L.KIND_CO2_BUDGET, 1_000_000, L.CB_DEFAULTS, dict(gtc_per_ppm=(2.0, 2.3)), L.CB_PARAM_NAMES,
                         np.stack([0.02 * yr, np.full(T, 0.5), 0.004 * yr, 0.006 * yr]), {1: 278.0}, 24, "no transcendental")
out["terrestrial_carbon"] = case("terrestrial_carbon", L.KIND_TERRESTRIAL_CARBON, 1_000_000, L.TC_DEFAULTS, dict(beta=(0.3, 0.9), npp_pi=(55.0, 75.0)), L.TC_PARAM_NAMES,
                                 np.stack([ramp(278.0, 1.001), 0.004 * yr, np.full(T, 0.3)]), {1: 884.86, 2: 92.77, 3: 1681.53, 4: 836.0}, 40,
                                 "1 log + 5 exp + 4 divisions per member-year")
out["ocean_carbon"] = case("ocean_carbon", L.KIND_OCEAN_CARBON, 262_144, L.OC_PRESETS["3D-GFDL"], dict(gas_exchange_tau=(6.0, 10.0), mixed_layer_depth=(45.0, 60.0)),
                           L.OC_PARAM_NAMES, np.stack([np.minimum(ramp(278.0, 1.003), 1100.0), np.minimum(0.006 * yr, 4.0)]), {1: 278.0, 2: 0.0},
                           3.0e6 * 8 / 3 / years + 24 + 96, "6000-month IRF convolution: 3.0e6 pulse reads (8 B, once per three years) and 7.2e7 f64 ops per member over 750 years")
out["ocean_carbon_fast"] = case("ocean_carbon_fast", L.KIND_OCEAN_CARBON, 262_144, L.OC_PRESETS["3D-GFDL"], dict(gas_exchange_tau=(6.0, 10.0), mixed_layer_depth=(45.0, 60.0)),
                                L.OC_PARAM_NAMES, np.stack([np.minimum(ramp(278.0, 1.003), 1100.0), np.minimum(0.006 * yr, 4.0)]), {1: 278.0, 2: 0.0},
                                12 * 16 + 24, "RSCM_MODE_FAST: O(T) recurrence -- 60 lags explicit, 21 fitted modes; 12 pulses written + 12 leaving pulses "
                                "read + 3 output rows per member-year", mode=1)
out["halocarbon"] = case("halocarbon", L.KIND_HALOCARBON, 100_000, L.HC_DEFAULTS, {"CFC-11.lifetime": (45.0, 60.0), "br_multiplier": (45.0, 75.0)}, L.HC_PARAM_NAMES,
                         np.tile(20.0 + 0.05 * yr, (41, 1)), {v: 10.0 for v in range(1, 42)}, 41 * 8 * 2 + 32,
                         "41 species: series written once, read once for the aggregates")
out["carbon_cycle"] = case("carbon_cycle", L.KIND_CARBON_CYCLE, 1_000_000, (25.0, 278.0, 0.05), dict(tau=(15.0, 40.0), alpha_temperature=(0.0, 0.1)), L.CC_PARAM_NAMES,
                           np.stack([0.02 * yr, 0.004 * yr]), {1: 278.0, 2: 0.0, 3: 0.0}, 24, "RK4, 10 sub-steps: 1 exp + 40 divisions per member-year")
out["carbon_cycle_fast"] = case("carbon_cycle_fast", L.KIND_CARBON_CYCLE, 1_000_000, (25.0, 278.0, 0.05), dict(tau=(15.0, 40.0), alpha_temperature=(0.0, 0.1)),
                                L.CC_PARAM_NAMES, np.stack([0.02 * yr, 0.004 * yr]), {1: 278.0, 2: 0.0, 3: 0.0}, 24,
                                "RSCM_MODE_FAST: the RK4 step of the linear box in closed form, 1 exp, no division", mode=1)
out["co2_erf"] = case("co2_erf", L.KIND_CO2_ERF, 1_000_000, (3.7, 278.0), dict(erf_2xco2=(3.0, 4.5)), L.CE_PARAM_NAMES, ramp(278.0, 1.001)[None], {}, 8,
                      "one log per member-year")
agg_in = np.full((8, T), np.nan)
agg_in[:3] = np.stack([0.004 * yr, 0.002 * yr, -0.001 * yr])
out["aggregate_sum"] = case("aggregate_sum", L.KIND_AGGREGATE, 1_000_000, (0.0,) + (1.0,) * 8, {}, L.AG_PARAM_NAMES, agg_in, {}, 8, "Sum of three contributors, five NaN rows skipped")
print(json.dumps({k: v for k, v in out.items() if v is not None}, indent=1))
