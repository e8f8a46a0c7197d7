"""BASELINE.json configs[3] (run on the GPU box): the emissions-driven MAGICC graph of the reference's
tests/regression/test_ghg_forcing.py -- ten rscm-magicc components, Sum of eight forcings, FourBox
transforms -- as linked ensembles, N members x 750 years, ClimateUDEB and OceanCarbon at 12
sub-steps per year.  Members differ in ECS, ocean diffusivity and the CO2 fertilisation factor."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd.core as core  # noqa: E402
from rscm_amd import _lib as L  # noqa: E402
from rscm_amd import magicc as B  # noqa: E402


def chain_inputs(years: int = 750, steps_per_year: int = 1):
    """(time values, exogenous series, initial values, aggregate contributors) of the benchmark
    scenario on an axis of `steps_per_year` model steps per year (12: the monthly axis of
    BASELINE.json configs[3])."""
    t = 1750.0 + np.arange(years * steps_per_year + 1) / float(steps_per_year)
    yrs = t - 1750.0
    ramp = np.minimum(yrs / 300.0, 1.0)
    # fossil CO2: up to 6 GtC/yr in 2050, down to 1 GtC/yr by 2150 and flat afterwards (~1600 GtC in all)
    fossil = np.interp(yrs, [0.0, 300.0, 400.0, 1e9], [0.0, 6.0, 1.0, 1.0])
    exo = {"Emissions|CO2|Fossil": fossil, "Emissions|CO2|Land Use": 0.5 + 0.0 * yrs, "Emissions|CH4": 200.0 + 200.0 * ramp,
           "Emissions|N2O": 7.0 + 5.0 * ramp, "Emissions|NOx": 10.0 + 30.0 * ramp, "Emissions|CO": 300.0 + 500.0 * ramp,
           "Emissions|NMVOC": 60.0 + 100.0 * ramp, "Emissions|SOx": 2.0 + 40.0 * ramp, "Emissions|BC": 2.5 + 5.0 * ramp,
           "Emissions|OC": 10.0 + 20.0 * ramp, "EESC": 1400.0 + 0.0 * yrs}
    init = {"Atmospheric Concentration|CO2": 278.0, "Atmospheric Concentration|CH4": 722.0, "Atmospheric Concentration|N2O": 270.0,
            "Surface Temperature": 0.0, "Ocean Surface pCO2": 278.0, "Cumulative Ocean Uptake": 0.0,
            "Carbon Pool|Plant": 884.86, "Carbon Pool|Detritus": 92.77, "Carbon Pool|Soil": 1681.53, "Carbon Pool|Humus": 836.0,
            "Effective Radiative Forcing": 0.0}
    contributors = ["Effective Radiative Forcing|CO2", "Effective Radiative Forcing|CH4", "Effective Radiative Forcing|N2O",
                    "Effective Radiative Forcing|O3|Stratospheric", "Effective Radiative Forcing|O3|Tropospheric",
                    "Effective Radiative Forcing|O3|Temperature Feedback", "Effective Radiative Forcing|Aerosol|Direct",
                    "Effective Radiative Forcing|Aerosol|Indirect"]
    return t, exo, init, contributors


def chain_components():
    """The ten rscm-magicc components in the registration order of the reference's emissions-driven
    model (tests/regression/test_ghg_forcing.py:399-563)."""
    return [B.CH4ChemistryBuilder.from_parameters({}).build(), B.N2OChemistryBuilder.from_parameters({}).build(),
            B.GhgForcingBuilder.from_parameters({"method": "Ipcctar"}).build(), B.OzoneForcingBuilder.from_parameters({}).build(),
            B.AerosolDirectBuilder.from_parameters({}).build(), B.AerosolIndirectBuilder.from_parameters({}).build(),
            B.ClimateUDEBBuilder.from_parameters({}).build(), B.TerrestrialCarbonBuilder.from_parameters({}).build(),
            B.OceanCarbonBuilder.from_parameters({}).build(), B.CO2BudgetBuilder.from_parameters({}).build()]


def build_chain(members: int, years: int = 750, order: str = "topological", steps_per_year: int = 1, device: int = 0,
                member_offset: int = 0, members_total: int = None, **build_kwargs):
    """The emissions-driven graph for `members` members; returns the GraphModel, ready to run.  `member_offset` / `members_total`:
    these members are the block [member_offset, member_offset + members) of one seeded draw of `members_total` members (one rank's
    share of a sharded ensemble; default: the whole draw)."""
    t, exo, init, contributors = chain_inputs(years, steps_per_year)
    schema = core.VariableSchema()
    for n in list(exo) + [k for k in init if k not in ("Surface Temperature", "Effective Radiative Forcing")] + contributors + [
            "Heat Uptake", "Ocean Heat Content", "Sea Surface Temperature", "Carbon Flux|Terrestrial", "Carbon Flux|Ocean",
            "Emissions|CO2|Net", "Airborne Fraction|CO2", "Lifetime|CH4", "Lifetime|N2O"]:
        schema.add_variable(n, "")
    schema.add_variable("Surface Temperature", "K", core.GridType.FourBox)
    schema.add_aggregate("Effective Radiative Forcing", "W/m^2", "Sum", contributors)
    comps = chain_components()
    axis = core.TimeAxis.from_values(t)
    bld = core.ModelBuilder().with_device(device).with_time_axis(axis).with_schema(schema).with_initial_values(init)
    for c in comps:
        bld.with_rust_component(c)
    for name, vals in exo.items():
        bld.with_exogenous_variable(name, core.Timeseries(vals, axis, "", core.InterpolationStrategy.Linear))
    model = bld.build(n_members=members, execution_order=order, **build_kwargs)
    model._chain_components = comps
    rng = np.random.default_rng(20260327)
    total = members if members_total is None else members_total
    if not (0 <= member_offset and member_offset + members <= total):
        raise ValueError("member block outside the draw")
    block = slice(member_offset, member_offset + members)
    ud = model.ensembles["ClimateUDEB"]
    P = ud.get_params()
    P[L.UD_PARAM_NAMES.index("ecs")] = rng.uniform(2.0, 4.5, total)[block]
    P[L.UD_PARAM_NAMES.index("kappa")] = rng.uniform(0.5, 1.2, total)[block]
    ud.set_params(P)
    tc = model.ensembles["TerrestrialCarbon"]
    P = tc.get_params()
    P[L.TC_PARAM_NAMES.index("beta")] = P[L.TC_PARAM_NAMES.index("beta")] * rng.uniform(0.7, 1.3, total)[block]
    tc.set_params(P)
    return model


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=100_000)
    ap.add_argument("--years", type=int, default=750)
    ap.add_argument("--order", default="topological")
    ap.add_argument("--fast", action="store_true", help="RSCM_MODE_FAST: fused multiply-adds in the ocean convolution")
    args = ap.parse_args()
    t0 = time.perf_counter()
    model = build_chain(args.members, args.years, args.order)
    if args.fast:
        model.set_mode(L.MODE_FAST)
    free, total = L.mem_info(0)
    print(f"built {len(model._order)} linked ensembles for {args.members} members in {time.perf_counter()-t0:.1f} s; "
          f"HBM in use {(total-free)/2**30:.1f} GiB; order: {model._order}", flush=True)
    t0 = time.perf_counter()
    model.run()
    dt = time.perf_counter() - t0
    my = args.members * args.years
    print(f"run: {dt*1e3:.0f} ms = {my/dt:.3g} member-years/s ({len(model._order) * args.years} launches)", flush=True)
    ts = model.ensembles["Transform:Surface Temperature"].summary(1, args.years)
    co2 = model.ensembles["CO2Budget"].summary(1, args.years)
    print(f"global-mean warming at the end: mean {ts['mean']:.3f} K [{ts['min']:.3f}, {ts['max']:.3f}], finite members {ts['count']}; "
          f"CO2 mean {co2['mean']:.1f} ppm [{co2['min']:.1f}, {co2['max']:.1f}]", flush=True)
    model.close()
