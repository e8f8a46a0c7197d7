"""GPU tier: the reference-shaped Python surface (ModelBuilder / TwoLayerBuilder / ModelRunner /
EnsembleSampler / PointEstimator) driving the HIP path, checked against the oracle's generic
component-by-component stepper and the reference's own API-level tests."""
import math
import os

import numpy as np
import pytest

from tests.helpers import assert_bit_equal, f_syn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P_DEFAULT = dict(lambda0=1.0, a=0.0, efficacy=1.0, eta=0.7, heat_capacity_surface=8.0,
                 heat_capacity_deep=100.0)
RTOL = 1e-11


@pytest.fixture(scope="module")
def api():
    from rscm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1
    import rscm_amd.calibrate as cal
    import rscm_amd.config as cfg
    import rscm_amd.core as core
    from rscm_amd.components import CarbonCycleBuilder, CO2ERFBuilder
    from rscm_amd.two_layer import TwoLayerBuilder

    class A:
        pass
    a = A()
    a.core, a.cal, a.cfg = core, cal, cfg
    a.TwoLayerBuilder, a.CarbonCycleBuilder, a.CO2ERFBuilder = TwoLayerBuilder, CarbonCycleBuilder, CO2ERFBuilder
    return a


@pytest.fixture(scope="module")
def rm():
    from oracle import reference_model
    return reference_model


def _collection(api, erf, t0, t1):
    c, S = api.core, api.core.InterpolationStrategy
    axis = c.TimeAxis.from_bounds(np.array([t0, t1]))
    coll = c.TimeseriesCollection()
    coll.add_timeseries("Effective Radiative Forcing", c.Timeseries(np.array([erf]), axis, "W/m^2", S.Previous))
    coll.add_timeseries("Surface Temperature", c.Timeseries(np.array([0.0]), axis, "K", S.Previous))
    coll.add_timeseries("Deep Ocean Temperature", c.Timeseries(np.array([0.0]), axis, "K", S.Previous))
    return coll


def test_two_layer_component_solve_reference_properties(api, known):
    """crates/rscm-two-layer/src/component.rs:299-406 through TwoLayerBuilder...build().solve()."""
    from oracle import cbind
    k = known["two_layer_properties"]
    comp = api.TwoLayerBuilder.from_parameters(k["params"]).build()

    def solve(erf):
        out = comp.solve(k["t0"], k["t1"], _collection(api, erf, k["t0"], k["t1"]))
        return out["Surface Temperature"].as_scalar()

    t = solve(k["erf_positive"])
    assert 0.0 < t < k["positive_upper_bound"]
    assert abs(solve(0.0)) < k["erf_zero_abs_tol"]
    assert solve(k["erf_negative"]) < 0.0
    small, large = solve(k["ratio_erf_small"]), solve(k["ratio_erf_large"])
    assert abs(large / small - k["ratio_expected"]) < k["ratio_tol"]
    p = [k["params"][n] for n in ("lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep")]
    assert t == cbind.two_layer_solve(p, k["erf_positive"], k["t0"], k["t1"], k["step"], 0.0, 0.0)[0]
    # the 10-year single solve of tests/test_calibration_integration.py:50-68 (100 RK4 sub-steps)
    true = dict(lambda0=1.1, efficacy=1.3, a=0.05, eta=0.7, heat_capacity_deep=100.0, heat_capacity_surface=8.0)
    got = api.TwoLayerBuilder.from_parameters(true).build().solve(2000, 2010, _collection(api, 3.0, 2000.0, 2010.0))
    pv = [true[n] for n in ("lambda0", "a", "efficacy", "eta", "heat_capacity_surface", "heat_capacity_deep")]
    assert got["Surface Temperature"].as_scalar() == cbind.two_layer_solve(pv, 3.0, 2000.0, 2010.0, 0.1, 0.0, 0.0)[0]


def test_config1_model_vs_generic_stepper(api, rm):
    """BASELINE configs[0]: configs/two-layer defaults, 1750-2100 annual, ERF attached by the
    harness; every value bit-identical to the oracle's generic ModelBuilder/Model restatement
    (including the <=1 ulp forward-extrapolation quirk of the last resampled forcing point)."""
    c = api.core
    conf = api.cfg.load_config(os.path.join(ROOT, "configs/two-layer/defaults.toml"))
    b = api.cfg.two_layer_builder(conf)
    t = b._axis.values()
    assert len(t) == 351
    F = f_syn(t)
    erf = c.Timeseries(F, c.TimeAxis.from_values(t), "W/m^2", c.InterpolationStrategy.Linear)
    model = b.with_exogenous_variable("Effective Radiative Forcing", erf).build()
    assert model.variable_sources()[("Effective Radiative Forcing", "TwoLayer")] == "Exogenous"
    assert model.current_time() == 1750.0
    model.step()
    model.step()
    assert model.current_time() == 1752.0 and model.current_time_bounds() == (1752.0, 1753.0)
    model.run()
    assert model.finished()
    with pytest.raises(RuntimeError, match="time_index"):
        model.step()
    got = model.timeseries()
    p = conf["components"]["climate"]["parameters"]
    ref = rm.ModelBuilder(axis=rm.TimeAxis.from_values(t),
                          components=[rm.TwoLayer(p["lambda0"], p["a"], p["efficacy"], p["eta"],
                                                  p["heat_capacity_surface"], p["heat_capacity_deep"])],
                          initial_values={"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0},
                          exogenous={"Effective Radiative Forcing":
                                     rm.ExoSeries(list(F), rm.TimeAxis.from_values(t), "Linear")}).build()
    ref.run()
    for name in ("Effective Radiative Forcing", "Surface Temperature", "Deep Ocean Temperature"):
        assert_bit_equal(got.get_timeseries_by_name(name).values(), ref.data[name], name)
    assert sorted(got.names()) == sorted(ref.data)
    model.close()
    # like the reference builder, no forcing attached -> NaN outputs after index 0
    bare = api.cfg.build_model(conf)
    bare.run()
    ts = bare.timeseries().get_timeseries_by_name("Surface Temperature").values()
    assert ts[0] == 0.0 and np.isnan(ts[1:]).all()
    bare.close()


def _coupled_builder(api, t, tl=None, cc=None):
    c = api.core
    schema = c.VariableSchema()
    for n, u in (("Emissions|CO2|Anthropogenic", "GtC / yr"), ("Surface Temperature", "K"),
                 ("Atmospheric Concentration|CO2", "ppm"), ("Cumulative Land Uptake", "Gt C"),
                 ("Cumulative Emissions|CO2", "Gt C"), ("Effective Radiative Forcing|CO2", "W/m^2"),
                 ("Deep Ocean Temperature", "K")):
        schema.add_variable(n, u)
    schema.add_aggregate("Effective Radiative Forcing", "W/m^2", "Sum", ["Effective Radiative Forcing|CO2"])
    years = np.array([1750.0, 1850.0, 1950.0, 2000.0, 2020.0, 2050.0, 2100.0])
    vals = np.array([0.0, 0.5, 3.0, 7.0, 10.0, 5.0, 1.0])
    emis = c.Timeseries(vals, c.TimeAxis.from_bounds(np.concatenate([years, [2101.0]])), "GtC / yr",
                        c.InterpolationStrategy.Linear)
    tl = tl or dict(lambda0=1.1, a=0.0, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    cc = cc or dict(tau=25.0, conc_pi=278.0, alpha_temperature=0.1)
    b = (c.ModelBuilder().with_time_axis(c.TimeAxis.from_values(t)).with_schema(schema)
         .with_rust_component(api.CarbonCycleBuilder.from_parameters(cc).build())
         .with_rust_component(api.CO2ERFBuilder.from_parameters(dict(erf_2xco2=3.7, conc_pi=278.0)).build())
         .with_rust_component(api.TwoLayerBuilder.from_parameters(tl).build())
         .with_exogenous_variable("Emissions|CO2|Anthropogenic", emis)
         .with_initial_values({"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
                               "Atmospheric Concentration|CO2": 278.0, "Surface Temperature": 0.0,
                               "Deep Ocean Temperature": 0.0}))
    return b, (years, vals), tl, cc


def test_notebook_coupled_model_vs_generic_stepper(api, rm):
    """docs/notebooks/coupled_model.py:357-510 (feedback-coupled model), 1750-2100."""
    t = np.arange(1750.0, 2101.0)
    b, (years, vals), tl, cc = _coupled_builder(api, t)
    model = b.build()
    model.run()
    got = model.timeseries()
    ref = rm.ModelBuilder(
        axis=rm.TimeAxis.from_values(t),
        components=[rm.CarbonCycle(cc["tau"], cc["conc_pi"], cc["alpha_temperature"]),
                    rm.CO2ERF(3.7, 278.0), rm.TwoLayer(*[tl[k] for k in api.core.TL_PARAM_ORDER])],
        aggregates=[("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2"])],
        exogenous={"Emissions|CO2|Anthropogenic":
                   rm.ExoSeries(list(vals), rm.TimeAxis.from_bounds(list(years) + [2101.0]), "Linear")},
        initial_values={"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
                        "Atmospheric Concentration|CO2": 278.0, "Surface Temperature": 0.0,
                        "Deep Ocean Temperature": 0.0}).build()
    ref.run()
    assert_bit_equal(got.get_timeseries_by_name("Emissions|CO2|Anthropogenic").values(),
                     ref.data["Emissions|CO2|Anthropogenic"], "resampled emissions")
    for name, want in ref.data.items():
        g = got.get_timeseries_by_name(name).values()
        w = np.array(want)
        assert (np.isnan(g) == np.isnan(w)).all(), name
        ok = ~np.isnan(w)
        assert (np.abs(g[ok] - w[ok]) <= RTOL * np.maximum(1.0, np.abs(w[ok]))).all(), name
    assert math.isnan(got.get_timeseries_by_name("Effective Radiative Forcing").values()[0])
    # physically sensible: CO2 rises above pre-industrial and the surface warms
    assert got.get_timeseries_by_name("Atmospheric Concentration|CO2").values()[-1] > 300.0
    assert got.get_timeseries_by_name("Surface Temperature").values()[-1] > 0.5
    model.close()


def test_two_components_of_one_type_the_one_that_runs_last_stands(api, rm):
    """builder.rs:531-559 accepts a second provider of a variable (the later component becomes its owner, with
    an edge from the earlier one); each writes index n+1 every step and the write of whichever runs last in the
    breadth-first order stands (runtime.rs:504-527).  Two TwoLayer components with different parameters in the
    coupled graph, the extra one registered second and registered last, against the generic stepper over one
    shared collection."""
    t = np.arange(1750.0, 1901.0)
    tl_a = dict(lambda0=0.9, a=0.02, efficacy=1.1, eta=0.6, heat_capacity_surface=7.0, heat_capacity_deep=90.0)
    finals = []
    for at in (1, 4):
        b, (years, vals), tl, cc = _coupled_builder(api, t)
        b._components.insert(at, api.TwoLayerBuilder.from_parameters(tl_a).build())
        model = b.build()
        model.run()
        got = model.timeseries()
        comps = [rm.CarbonCycle(cc["tau"], cc["conc_pi"], cc["alpha_temperature"]), rm.CO2ERF(3.7, 278.0),
                 rm.TwoLayer(*[tl[k] for k in api.core.TL_PARAM_ORDER])]
        comps.insert(at, rm.TwoLayer(*[tl_a[k] for k in api.core.TL_PARAM_ORDER]))
        ref = rm.ModelBuilder(
            axis=rm.TimeAxis.from_values(t), components=comps,
            aggregates=[("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2"])],
            exogenous={"Emissions|CO2|Anthropogenic":
                       rm.ExoSeries(list(vals), rm.TimeAxis.from_bounds(list(years) + [2101.0]), "Linear")},
            initial_values={"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
                            "Atmospheric Concentration|CO2": 278.0, "Surface Temperature": 0.0,
                            "Deep Ocean Temperature": 0.0}).build()
        ref.run()
        for name, want in ref.data.items():
            g = got.get_timeseries_by_name(name).values()
            w = np.array(want)
            assert (np.isnan(g) == np.isnan(w)).all(), (at, name)
            ok = ~np.isnan(w)
            assert (np.abs(g[ok] - w[ok]) <= RTOL * np.maximum(1.0, np.abs(w[ok]))).all(), (at, name)
        finals.append(got.get_timeseries_by_name("Surface Temperature").values()[-1])
        assert finals[-1] > 0.05
        model.close()
    assert finals[0] != finals[1]  # the survivor differs between the two registrations


def test_coupled_model_with_two_preindustrial_concentrations(api, rm):
    """CarbonCycle.conc_pi != CO2ERF.conc_pi (builder.rs allows any parameters): the fused coupled kernel
    carries one conc_pi row, so the model is built as the same graph of linked ensembles -- and has to
    reproduce the generic stepper like the fused one does."""
    c = api.core
    t = np.arange(1750.0, 1901.0)
    b, (years, vals), tl, cc = _coupled_builder(api, t)
    b._components[1] = api.CO2ERFBuilder.from_parameters(dict(erf_2xco2=3.7, conc_pi=285.0)).build()
    model = b.build()
    assert isinstance(model, c.GraphModel)
    model.run()
    got = model.timeseries()
    ref = rm.ModelBuilder(
        axis=rm.TimeAxis.from_values(t),
        components=[rm.CarbonCycle(cc["tau"], cc["conc_pi"], cc["alpha_temperature"]),
                    rm.CO2ERF(3.7, 285.0), rm.TwoLayer(*[tl[k] for k in api.core.TL_PARAM_ORDER])],
        aggregates=[("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2"])],
        exogenous={"Emissions|CO2|Anthropogenic":
                   rm.ExoSeries(list(vals), rm.TimeAxis.from_bounds(list(years) + [2101.0]), "Linear")},
        initial_values={"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
                        "Atmospheric Concentration|CO2": 278.0, "Surface Temperature": 0.0,
                        "Deep Ocean Temperature": 0.0}).build()
    ref.run()
    for name, want in ref.data.items():
        g = got.get_timeseries_by_name(name).values()
        w = np.array(want)
        assert (np.isnan(g) == np.isnan(w)).all(), name
        ok = ~np.isnan(w)
        assert (np.abs(g[ok] - w[ok]) <= RTOL * np.maximum(1.0, np.abs(w[ok]))).all(), name
    assert got.get_timeseries_by_name("Effective Radiative Forcing|CO2").values()[1] < 0.0   # 278 ppm against a 285 ppm baseline
    model.close()


@pytest.mark.parametrize("operation", ["Sum", "Weighted", "Mean"])
def test_aggregate_of_more_than_eight_contributors(api, operation):
    """compute_aggregate (schema.rs:760-802) folds any number of contributors; an aggregate ensemble takes
    eight, so eleven run as two chained stages whose additions keep the declaration order: the same bits
    as one running sum (a Mean: that sum, a chained count of the non-NaN contributors, and their quotient).
    Some contributors are NaN at some times (skipped), one row is NaN throughout."""
    c = api.core
    t = np.arange(1850.0, 1881.0)
    axis = c.TimeAxis.from_values(t)
    rng = np.random.default_rng(4)
    names = [f"Effective Radiative Forcing|Part{k}" for k in range(11)]
    series = {n: rng.normal(0.3, 1.0, len(t)) for n in names}
    series[names[2]][5:9] = np.nan
    series[names[9]][:] = np.nan
    series[names[10]][20] = np.nan
    weights = list(rng.uniform(0.2, 2.0, 11)) if operation == "Weighted" else None
    schema = c.VariableSchema()
    for n in names + ["Surface Temperature", "Deep Ocean Temperature"]:
        schema.add_variable(n, "")
    schema.add_aggregate("Effective Radiative Forcing", "W/m^2", operation, names, weights)
    fixed = dict(lambda0=1.1, a=0.05, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (c.ModelBuilder().with_time_axis(axis).with_schema(schema)
         .with_rust_component(api.TwoLayerBuilder.from_parameters(fixed).build())
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    for n in names:
        b = b.with_exogenous_variable(n, c.Timeseries(series[n], axis, "W/m^2", c.InterpolationStrategy.Previous))
    model = b.build()
    assert isinstance(model, c.GraphModel) and "Aggregator:Effective Radiative Forcing#0" in model._order
    model.run()
    got = model.timeseries()
    assert not any("#" in n for n in got.names())
    erf = got.get_timeseries_by_name("Effective Radiative Forcing").values()
    want = np.full(len(t), np.nan)
    for n in range(len(t) - 1):      # contributors at n + 1, in declaration order, NaN skipped
        acc, cnt = 0.0, 0
        for k, name in enumerate(names):
            v = series[name][n + 1]
            if not np.isnan(v):
                acc = acc + (v * weights[k] if weights else v)
                cnt += 1
        want[n + 1] = (acc / float(cnt) if operation == "Mean" else acc) if cnt else np.nan
    assert_bit_equal(erf, want, f"{operation} of eleven contributors")
    assert np.isfinite(got.get_timeseries_by_name("Surface Temperature").values()[1:]).all()
    model.close()


@pytest.mark.parametrize("operation", ["Sum", "Mean", "Weighted"])
def test_fourbox_aggregate_per_region_and_its_scalar_view(api, operation):
    """A FourBox-typed aggregate (AggregatorComponent::solve, schema.rs:902-923): compute_aggregate per region over the contributors'
    values of that region at n + 1.  Two FourBox producers (AerosolDirect's regional forcing, FourBoxOceanHeatUptake's regional heat
    uptake) are aggregated region by region; TwoLayer declares its forcing as a scalar and reads the aggregate through the read
    transform (sum over the regions of value x weight, state/aggregating.rs:162-176).  Checked: every region of the aggregate is the
    running combination of the contributors' regions in declaration order, bit for bit; the scalar view is the weighted sum of the
    regions; the TwoLayer run equals a stand-alone ensemble forced with that scalar series as an upstream output; the collection
    holds the aggregate as a FourBox timeseries."""
    import rscm_amd as ra
    from rscm_amd import magicc as B
    from rscm_amd.components import FourBoxOceanHeatUptakeBuilder
    c = api.core
    t = np.arange(1850.0, 1891.0)
    axis = c.TimeAxis.from_values(t)
    yr = t - 1850.0
    direct, uptake, agg = "Effective Radiative Forcing|Aerosol|Direct", "Heat Uptake|Ocean", "Effective Radiative Forcing"
    exo = {"Emissions|SOx": 2.0 + 1.5 * yr, "Emissions|BC": 2.5 + 0.1 * yr, "Emissions|OC": 10.0 + 0.3 * yr, "Emissions|NOx": 10.0 + 0.5 * yr,
           "Effective Radiative Forcing|Aggregated": 0.5 + 0.04 * yr + 0.2 * np.sin(yr / 3.0)}
    schema = c.VariableSchema()
    for n in list(exo) + ["Surface Temperature", "Deep Ocean Temperature"]:
        schema.add_variable(n, "")
    schema.add_variable(direct, "", c.GridType.FourBox)
    schema.add_variable(uptake, "", c.GridType.FourBox)
    weights = [1.0, -0.5] if operation == "Weighted" else None
    schema.add_aggregate(agg, "", operation, [direct, uptake], weights, grid_type=c.GridType.FourBox)
    fixed = dict(lambda0=1.1, a=0.05, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    ratios = dict(northern_ocean_ratio=1.1, northern_land_ratio=0.8, southern_ocean_ratio=1.2, southern_land_ratio=0.9)
    b = (c.ModelBuilder().with_time_axis(axis).with_schema(schema)
         .with_rust_component(B.AerosolDirectBuilder.from_parameters({}).build())
         .with_rust_component(FourBoxOceanHeatUptakeBuilder.from_parameters(ratios).build())
         .with_rust_component(api.TwoLayerBuilder.from_parameters(fixed).build())
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    for n, v in exo.items():
        b = b.with_exogenous_variable(n, c.Timeseries(v, axis, "", c.InterpolationStrategy.Linear))
    model = b.build(execution_order="topological")
    assert isinstance(model, c.GraphModel)
    mine = [n for n in model._order if n == f"Transform:{agg}" or n.startswith(f"Aggregator:{agg}|box")]
    assert mine == [f"Aggregator:{agg}|box{k}" for k in range(4)] + [f"Transform:{agg}"]
    model.run()
    ad, ohu = model.ensembles["AerosolDirect"], model.ensembles["FourBoxOceanHeatUptake"]
    view = np.zeros(len(t))
    for k in range(4):
        a_k, u_k = ad.get_series(1 + k)[:, 0], ohu.get_series(1 + k)[:, 0]
        got = model.get_series(f"{agg}|box{k}")[:, 0]
        want = np.full(len(t), np.nan)
        if operation == "Sum":
            want[1:] = (0.0 + a_k[1:]) + u_k[1:]
        elif operation == "Mean":
            want[1:] = ((0.0 + a_k[1:]) + u_k[1:]) / 2.0
        else:
            want[1:] = (0.0 + a_k[1:] * 1.0) + u_k[1:] * -0.5
        assert_bit_equal(got, want, f"{operation}, region {k}")
        view = view + np.nan_to_num(got) * 0.25 if k else 0.0 + np.nan_to_num(got) * 0.25
    scalar = model.get_series(agg)[:, 0]
    assert_bit_equal(scalar[1:], view[1:], "the scalar view: sum over the regions of value x 0.25")
    # TwoLayer read that scalar at n + 1 (an upstream output)
    bounds = np.append(t, t[-1] + 1.0)
    with ra.Ensemble(ra.KIND_TWO_LAYER, 1, bounds) as e:
        e.set_params(np.array([[fixed[k]] for k in fixed]))
        e.set_forcing(np.nan_to_num(scalar), None, ra.SRC_UPSTREAM)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run()
        assert_bit_equal(model.get_series("Surface Temperature")[:, 0], e.get_series(1)[:, 0], "TwoLayer forced by the FourBox aggregate's scalar view")
    coll = model.timeseries()
    fb = coll.get_fourbox_timeseries_by_name(agg)
    assert fb is not None and not any("|box" in n for n in coll.names())
    model.close()
    # a contributor that is not a FourBox variable is refused like the reference refuses a grid-type mismatch
    bad = c.VariableSchema()
    bad.add_variable("Emissions|SOx", "")
    bad.add_variable(direct, "", c.GridType.FourBox)
    bad.add_aggregate("X", "", "Sum", [direct, "Emissions|SOx"], None, grid_type=c.GridType.FourBox)
    with pytest.raises(ValueError, match="Grid type mismatch"):
        bad.validate()


def _tl_runner(api, t, F, names, outputs=("Surface Temperature",), mode=0):
    c = api.core
    erf = c.Timeseries(F, c.TimeAxis.from_values(t), "W/m^2", c.InterpolationStrategy.Linear)
    fixed = dict(lambda0=1.1, a=0.05, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0, heat_capacity_deep=100.0)
    b = (c.ModelBuilder().with_time_axis(c.TimeAxis.from_values(t))
         .with_rust_component(api.TwoLayerBuilder.from_parameters(fixed).build())
         .with_exogenous_variable("Effective Radiative Forcing", erf)
         .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    return api.cal.ModelRunner(b, names, list(outputs), mode=mode), fixed


def test_model_runner_batch_vs_oracle(api):
    from oracle import cbind
    t = np.arange(1850.0, 2021.0)
    F = f_syn(t)
    runner, fixed = _tl_runner(api, t, F, ["lambda0", "a"], ("Surface Temperature", "Deep Ocean Temperature"))
    assert runner.param_names == ["lambda0", "a"]
    rng = np.random.default_rng(5)
    sets = np.column_stack([rng.uniform(0.8, 1.5, 37), rng.uniform(0.0, 0.1, 37)])
    outs = runner.run_batch(sets)
    assert len(outs) == 37
    # the resampled forcing the model actually sees (last point through the extrapolation formula)
    c = api.core
    Fm = c.Timeseries(F, c.TimeAxis.from_values(t), "", c.InterpolationStrategy.Linear).interpolate_into(
        c.TimeAxis.from_values(t)).values()
    P = np.repeat(np.array([fixed[k] for k in api.core.TL_PARAM_ORDER])[:, None], 37, axis=1)
    P[0], P[1] = sets[:, 0], sets[:, 1]
    want_ts, want_td = cbind.two_layer_run(cbind.bounds_from_values(t), P, Fm, 0.0, 0.0)
    for i, o in enumerate(outs):  # order-preserving
        assert list(o) == ["Surface Temperature", "Deep Ocean Temperature"]
        assert list(o["Surface Temperature"]) == list(t)
        assert_bit_equal(np.array(list(o["Surface Temperature"].values())), want_ts[:, i])
        assert_bit_equal(np.array(list(o["Deep Ocean Temperature"].values())), want_td[:, i])
    single = runner.run(sets[3])
    assert single == outs[3]
    with pytest.raises(ValueError, match="Expected 2 parameters, got 3"):
        runner.run([1.0, 0.0, 2.0])
    with pytest.raises(ValueError, match="unknown model parameter"):
        _tl_runner(api, t, F, ["nope"])
    with pytest.raises(KeyError, match="missing variable"):
        _tl_runner(api, t, F, ["a"], ("Nope",))
    # device likelihood == host likelihood over the extracted outputs
    lik = api.cal.GaussianLikelihood()
    target = api.cal.Target()
    for yr in range(1850, 2021, 10):
        target.add_observation("Surface Temperature", float(yr), outs[0]["Surface Temperature"][float(yr)] + 0.03, 0.1)
    target.add_observation("Deep Ocean Temperature", 2000.0, 0.2, 0.05)
    dev = runner.log_likelihood_batch(sets, target, lik)
    host = np.array([lik.ln_likelihood(o, target) for o in outs])
    assert np.allclose(dev, host, rtol=1e-13, atol=0)
    # an observation at a time the model does not have -> every member fails (-inf)
    bad = api.cal.Target().add_observation("Surface Temperature", 1700.0, 0.0, 0.1)
    assert (runner.log_likelihood_batch(sets, bad, lik) == -np.inf).all()
    runner.close()


def test_calibration_recovers_parameters(api):
    """The workflow of tests/test_calibration_integration.py (synthetic truth -> point estimate
    -> MCMC), with every batch evaluated in one launch."""
    cal = api.cal
    t = np.arange(1850.0, 2021.0)
    F = f_syn(t)
    runner, fixed = _tl_runner(api, t, F, ["lambda0", "a"])
    truth = runner.run([1.1, 0.05])["Surface Temperature"]
    target = cal.Target()
    for yr in range(1860, 2021, 10):
        target.add_observation("Surface Temperature", float(yr), truth[float(yr)], 0.05)
    params = cal.ParameterSet().add("lambda0", cal.Uniform(0.8, 1.5)).add("a", cal.Uniform(0.0, 0.1))
    lik = cal.GaussianLikelihood()
    rng = np.random.default_rng(42)
    est = cal.PointEstimator(params, runner, lik, target)
    res = est.optimize(cal.Optimizer.random_search(), n_samples=2000, rng=rng)
    assert est.n_evaluations == 2000 and res.n_evaluations == 2000 and res.converged
    best = dict(zip(est.param_names, res.best_params))
    assert 0.8 <= best["lambda0"] <= 1.5 and 0.0 <= best["a"] <= 0.1
    assert math.isfinite(res.best_log_likelihood)
    assert res.best_log_likelihood > -5.0  # close to the truth (lnL = 0 there)
    sampler = cal.EnsembleSampler(params, runner, lik, target)
    assert sampler.default_n_walkers == 32
    chain = sampler.run(150, cal.WalkerInit.from_prior(), thin=1, rng=rng)
    assert chain.total_iterations == 150 and chain.thin == 1 and chain.param_names == ["lambda0", "a"]
    flat = chain.flat_samples(discard=75)
    assert flat.shape == (75 * 32, 2)
    assert ((flat[:, 0] >= 0.8) & (flat[:, 0] <= 1.5) & (flat[:, 1] >= 0.0) & (flat[:, 1] <= 0.1)).all()
    # the data constrain lambda0 - a*T; both posterior means land near the truth
    assert abs(flat[:, 0].mean() - 1.1) < 0.1 and abs(flat[:, 1].mean() - 0.05) < 0.03
    assert 0.05 < sampler.acceptance_rate() < 0.95
    with pytest.raises(ValueError, match="even"):
        sampler.run(1, cal.WalkerInit.from_prior(), n_walkers=5)
    runner.close()


def test_model_toml_round_trip_like_the_reference(api):
    """tests/test_model.py::test_model_serialisation of the reference, on the GPU front: step once,
    to_toml, Model.from_toml, same graph and current time, and the rest of the run gives the same
    series -- here bit for bit -- for a fused model and for a graph of linked ensembles."""
    c = api.core
    t = np.arange(1750.0, 1781.0)
    axis = c.TimeAxis.from_values(t)
    erf = c.Timeseries(np.asarray([1.0] * len(t)), axis, "W / m^2", c.InterpolationStrategy.Next)
    tl = dict(lambda0=1.1, a=0.02, efficacy=1.2, eta=0.7, heat_capacity_deep=100.0, heat_capacity_surface=8.0)
    model = (c.ModelBuilder().with_time_axis(axis).with_rust_component(api.TwoLayerBuilder.from_parameters(tl).build())
             .with_exogenous_variable("Effective Radiative Forcing", erf)
             .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}).build())
    assert "TwoLayer" in model.as_dot()
    model.step()
    text = model.to_toml()
    assert "[[components]]" in text and 'type = "TwoLayer"' in text
    new_model = c.Model.from_toml(text)
    assert new_model.as_dot() == model.as_dot() and new_model.current_time() == model.current_time() == 1751.0
    model.run()
    new_model.run()
    assert_bit_equal(new_model.timeseries().get_timeseries_by_name("Surface Temperature").values(),
                     model.timeseries().get_timeseries_by_name("Surface Temperature").values(), "fused model")
    model.close()
    new_model.close()
    # a graph model: the notebook's chain with a second aggregate, checkpointed mid-run
    b, _, _, _ = _coupled_builder(api, t)
    b._schema.add_aggregate("Diagnostic", "K", "Mean", ["Surface Temperature", "Deep Ocean Temperature"])
    g = b.build(n_members=2)
    assert isinstance(g, c.GraphModel)
    for _ in range(7):
        g.step()
    h = c.Model.from_toml(g.to_toml())
    assert isinstance(h, c.GraphModel) and h.time_index == 7 and h.as_dot() == g.as_dot() and h.n_members == 2
    g.run()
    h.run()
    for name in ("Surface Temperature", "Atmospheric Concentration|CO2", "Effective Radiative Forcing", "Diagnostic"):
        assert_bit_equal(h.get_series(name)[7:], g.get_series(name)[7:], name)
    with pytest.raises(ValueError, match="not a model written by rscm_amd"):
        c.Model.from_toml("[model]\nformat = \"something else\"\n")
    g.close()
    h.close()


def test_debug_info_shows_the_execution_order_and_sources(api):
    """Model::debug_info (docs/notebooks/debugging_inspection.py of the reference): plain / rich / json."""
    import json
    c = api.core
    t = np.arange(1750.0, 1756.0)
    b, _, _, _ = _coupled_builder(api, t)
    fused = b.build()
    info = json.loads(fused.debug_info("json"))
    assert [e["name"] for e in info["components"]] == ["CarbonCycle", "CO2ERF", "Aggregator:Effective Radiative Forcing", "TwoLayer"]
    cc = info["components"][0]
    assert {i["name"]: i["source"] for i in cc["inputs"]} == {"Emissions|CO2|Anthropogenic": "exo", "Surface Temperature": "exo"}
    assert len(cc["states"]) == 3 and info["components"][3]["inputs"][0]["source"] == "upstream"
    plain = fused.debug_info()
    assert "[3] TwoLayer" in plain and "<- Effective Radiative Forcing (upstream)" in plain and "<> Surface Temperature" in plain
    assert "\x1b[" in fused.debug_info("rich") and "\x1b[" not in plain
    with pytest.raises(ValueError, match="Unknown format"):
        fused.debug_info("yaml")
    fused.close()
    b._schema.add_aggregate("Diagnostic", "K", "Mean", ["Surface Temperature", "Deep Ocean Temperature"])
    g = b.build()
    names = [e["name"] for e in json.loads(g.debug_info("json"))["components"]]
    assert names == list(g._order) and "Aggregator:Diagnostic" in names
    assert "(Mean)" in g.debug_info("plain")
    g.close()
