"""End to end (run on the GPU box): calibrate the climate sensitivity of the emissions-driven MAGICC
graph against MAGICC7's own SSP245 temperatures (tests/golden/magicc7_emissions_driven.json, generated
with core_climatesensitivity = 3.0), with rscm-calibrate's affine-invariant sampler driving batches
of members through the ten linked components on the GPU.
    python scripts/calibrate_magicc_chain.py [--fast] [--device-sampler [--walkers 4096] [--iterations 120]]
--device-sampler: the whole stretch-move loop on the device with the GRAPH as the evaluator (rscm_sampler_create_graph): no host
round trip per sweep; the host sampler's run (64 walkers) is printed beside it for the posterior comparison."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rscm_amd.calibrate as cal  # noqa: E402
import rscm_amd.core as core  # noqa: E402
from rscm_amd import magicc as B  # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "magicc7_emissions_driven.json")))
V = {k: np.array(v) for k, v in g["variables"].items()}
t = np.array(g["years"], dtype=float)


def sectors(base):
    return V[f"{base}|MAGICC Fossil and Industrial"] + V[f"{base}|MAGICC AFOLU"]


exo = {"Emissions|CO2|Fossil": V["Emissions|CO2"], "Emissions|CO2|Land Use": 0.0 * t, "Emissions|CH4": V["Emissions|CH4"],
       "Emissions|N2O": V["Emissions|N2O"], "Emissions|NOx": sectors("Emissions|NOx"), "Emissions|CO": sectors("Emissions|CO"),
       "Emissions|NMVOC": sectors("Emissions|NMVOC"), "Emissions|SOx": sectors("Emissions|SOx"),
       "Emissions|BC": sectors("Emissions|BC"), "Emissions|OC": sectors("Emissions|OC"), "EESC": 0.0 * t}
co2_0, ch4_0, n2o_0 = (float(V[f"Atmospheric Concentrations|{s}"][0]) for s in ("CO2", "CH4", "N2O"))
init = {"Atmospheric Concentration|CO2": co2_0, "Atmospheric Concentration|CH4": ch4_0, "Atmospheric Concentration|N2O": n2o_0,
        "Surface Temperature": 0.0, "Ocean Surface pCO2": co2_0, "Cumulative Ocean Uptake": 0.0, "Carbon Pool|Plant": 884.86,
        "Carbon Pool|Detritus": 92.77, "Carbon Pool|Soil": 1681.53, "Carbon Pool|Humus": 836.0, "Effective Radiative Forcing": 0.0}
contributors = ["Effective Radiative Forcing|CO2", "Effective Radiative Forcing|CH4", "Effective Radiative Forcing|N2O",
                "Effective Radiative Forcing|O3|Stratospheric", "Effective Radiative Forcing|O3|Tropospheric",
                "Effective Radiative Forcing|O3|Temperature Feedback", "Effective Radiative Forcing|Aerosol|Direct",
                "Effective Radiative Forcing|Aerosol|Indirect"]
schema = core.VariableSchema()
for n in list(exo) + [k for k in init if k not in ("Surface Temperature", "Effective Radiative Forcing")] + contributors + [
        "Heat Uptake", "Ocean Heat Content", "Sea Surface Temperature", "Carbon Flux|Terrestrial", "Carbon Flux|Ocean",
        "Emissions|CO2|Net", "Airborne Fraction|CO2", "Lifetime|CH4", "Lifetime|N2O"]:
    schema.add_variable(n, "")
schema.add_variable("Surface Temperature", "K", core.GridType.FourBox)
schema.add_aggregate("Effective Radiative Forcing", "W/m^2", "Sum", contributors)
axis = core.TimeAxis.from_values(t)
builder = core.ModelBuilder().with_time_axis(axis).with_schema(schema).with_initial_values(init)
for c in (B.CH4ChemistryBuilder.from_parameters({"ch4_pi": ch4_0}), B.N2OChemistryBuilder.from_parameters({"n2o_pi": n2o_0}),
          B.GhgForcingBuilder.from_parameters({"method": "Ipcctar", "delq2xco2": 3.71, "co2_pi": co2_0, "ch4_pi": ch4_0, "n2o_pi": n2o_0}),
          B.OzoneForcingBuilder.from_parameters({}), B.AerosolDirectBuilder.from_parameters({}), B.AerosolIndirectBuilder.from_parameters({}),
          B.ClimateUDEBBuilder.from_parameters({"ecs": 3.0, "rf_2xco2": 3.71}), B.TerrestrialCarbonBuilder.from_parameters({}),
          B.OceanCarbonBuilder.from_parameters({}), B.CO2BudgetBuilder.from_parameters({})):
    builder.with_rust_component(c.build())
for name, vals in exo.items():
    builder.with_exogenous_variable(name, core.Timeseries(vals, axis, "", core.InterpolationStrategy.Linear))

# the scalar view of the FourBox surface temperature is what MAGICC7's global mean compares with
FAST = "--fast" in sys.argv   # RSCM_MODE_FAST: OceanCarbon's O(T) recurrence instead of the literal convolution
runner = cal.ModelRunner(builder, ["ClimateUDEB.ecs", "ClimateUDEB.kappa"], ["Surface Temperature"], execution_order="topological",
                         mode=1 if FAST else 0)
target = cal.Target()
for year in range(1900, 2100, 10):          # MAGICC7's year n is our index n + 1 (upstream's comparison)
    target.add_observation("Surface Temperature", float(year + 1), float(V["Surface Temperature"][year - 1750]), 0.05)
params = cal.ParameterSet().add("ClimateUDEB.ecs", cal.Uniform(1.5, 6.0)).add("ClimateUDEB.kappa", cal.Uniform(0.3, 2.0))
sampler = cal.EnsembleSampler(params, runner, cal.GaussianLikelihood(), target)


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


t0 = time.perf_counter()
n_iter, walkers = 120, 64
chain = sampler.run(n_iter, cal.WalkerInit.from_prior(), n_walkers=walkers, rng=np.random.default_rng(7))
dt = time.perf_counter() - t0
flat = chain.flat_samples(discard=60)
lo, med, hi = np.percentile(flat[:, 0], [5, 50, 95])
print(f"host sampler: {n_iter} stretch-move iterations x {walkers} walkers = {n_iter * walkers} runs of the ten-component graph over 350 years in {dt:.1f} s "
      f"({n_iter * walkers / dt:.0f} model runs/s)")
print(f"ECS | MAGICC7 SSP245 temperatures (generated with ECS 3.0): median {med:.2f} K [5-95 %: {lo:.2f}, {hi:.2f}]; "
      f"kappa median {np.median(flat[:, 1]):.2f} cm^2/s; acceptance {sampler.acceptance_rate():.2f}")
if "--device-sampler" in sys.argv:
    dev = cal.DeviceEnsembleSampler(params, runner, cal.GaussianLikelihood(), target)
    d_iter, d_walkers = arg("--iterations", 120), arg("--walkers", 4096)
    t0 = time.perf_counter()
    dchain = dev.run(d_iter, cal.WalkerInit.from_prior(), n_walkers=d_walkers, seed=7)
    ddt = time.perf_counter() - t0
    dflat = dchain.flat_samples(discard=d_iter // 2)
    dlo, dmed, dhi = np.percentile(dflat[:, 0], [5, 50, 95])
    print(f"device sampler over the graph: {d_iter} iterations x {d_walkers} walkers = {d_iter * d_walkers} runs in {ddt:.1f} s "
          f"({d_iter * d_walkers / ddt:.0f} model runs/s, {dev.device_ms / d_iter:.1f} ms of device time per sweep)")
    print(f"ECS median {dmed:.2f} K [5-95 %: {dlo:.2f}, {dhi:.2f}]; kappa median {np.median(dflat[:, 1]):.2f} cm^2/s; "
          f"acceptance {dev.acceptance_rate():.2f}; host posterior ECS {med:.2f} [{lo:.2f}, {hi:.2f}]")
    print(json.dumps({"host": {"runs": n_iter * walkers, "seconds": dt, "ecs_median": med, "ecs_5_95": [lo, hi], "kappa_median": float(np.median(flat[:, 1]))},
                      "device": {"runs": d_iter * d_walkers, "seconds": ddt, "walkers": d_walkers, "iterations": d_iter, "ms_per_sweep": dev.device_ms / d_iter,
                                 "ecs_median": dmed, "ecs_5_95": [dlo, dhi], "kappa_median": float(np.median(dflat[:, 1])),
                                 "acceptance": dev.acceptance_rate()}}))
runner.close()
