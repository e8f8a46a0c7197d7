"""CPU tier: host logic of the product front-end (no GPU calls): time axis, interpolation and
builder graph resolution in rscm_amd.core against the reference's known answers and against the
oracle's independent restatement; priors, Latin hypercube, host likelihood and chain diagnostics
in rscm_amd.calibrate; member sharding arithmetic."""
import math

import numpy as np
import pytest

from oracle import reference_model as rm
from rscm_amd import calibrate as cal
from rscm_amd import core
from rscm_amd.components import CarbonCycleBuilder, CO2ERFBuilder
from rscm_amd.distributed import shard_bounds
from rscm_amd.two_layer import TwoLayerBuilder

S = core.InterpolationStrategy


def test_time_axis_doctests(known):
    k = known["time_axis"]
    ta = core.TimeAxis.from_values(k["from_values"])
    assert list(ta.at_bounds(2)) == k["at_bounds_2"]
    assert ta.at(1) == k["at_1"] and ta.at(27) is None
    assert ta.contains(1.0) and not ta.contains(27.0)
    assert ta.index_of(2.0) == 1 and ta.index_of(27.0) is None
    assert len(core.TimeAxis.from_bounds(k["from_bounds"])) == 3
    with pytest.raises(ValueError):
        core.TimeAxis.from_values([2020.0, 1.0, 2021.0])


def test_interpolation_tables(known):
    k = known["interp_linear"]
    t, y = np.array(k["time"]), np.array(k["y"])
    for q, e in zip(k["targets"], k["expected"]):
        assert math.isclose(core.interpolate(S.Linear, t, y, q, False), e, rel_tol=1e-9)
    for q, e in zip(k["extrap_targets"], k["extrap_expected"]):
        assert math.isclose(core.interpolate(S.Linear, t, y, q, True), e, rel_tol=1e-9)
    for q in k["noextrap_error_targets"]:
        with pytest.raises(RuntimeError, match="Extrapolation is not allowed"):
            core.interpolate(S.Linear, np.array(k["noextrap_time"]), np.array(k["noextrap_y"]), q, False)
    k = known["interp_previous"]
    t, y = np.array(k["time"]), np.array(k["y"])
    for q, e in zip(k["extrap_targets"], k["extrap_expected"]):
        assert core.interpolate(S.Previous, t, y, q, True) == e
    k = known["interp1d_next_extrapolate"]
    assert core.interpolate(S.Next, np.array(k["years"]), np.array(k["data"]), k["query"]) == k["expected"]
    k = known["timeseries_at_time"]
    ts = core.Timeseries.from_values(k["custom_data"], k["custom_years"])
    assert ts.at_time(k["custom_query"]) == k["custom_linear_expected"]
    ts.with_interpolation_strategy(S.Previous)
    assert ts.at_time(k["custom_query"]) == k["custom_previous_expected"]


def test_product_interpolation_equals_oracle_restatement():
    rng = np.random.default_rng(0)
    src = np.sort(rng.uniform(1700, 2200, 40))
    y = rng.normal(size=40)
    q = np.concatenate([rng.uniform(1650, 2250, 300), src, src + 1e-12])
    for strat, name in ((S.Linear, "Linear"), (S.Previous, "Previous"), (S.Next, "Next")):
        for t in q:
            a = core.interpolate(strat, src, y, float(t), True)
            b = rm.interpolate(name, list(src), list(y), float(t), True)
            assert a == b or (math.isnan(a) and math.isnan(b))
    axis = core.TimeAxis.from_values(np.arange(1750.0, 1800.0))
    ts = core.Timeseries(y, core.TimeAxis.from_values(src), "x", S.Linear)
    got = ts.interpolate_into(axis).values()
    want = rm.interpolate_into("Linear", rm.TimeAxis.from_values(src), list(y),
                               rm.TimeAxis.from_values(np.arange(1750.0, 1800.0)))
    assert list(got) == want


def test_previous_resample_golden(known):
    k = known["stepper_exogenous_previous"]
    ts = core.Timeseries(k["emissions_values"], core.TimeAxis.from_bounds(k["emissions_bounds"]),
                         "GtC / yr", S.Previous)
    got = ts.interpolate_into(core.TimeAxis.from_values(k["time_values"])).values()
    assert list(got) == k["resampled_emissions"]


P_TL = dict(lambda0=1.1, a=0.0, efficacy=1.3, eta=0.7, heat_capacity_surface=8.0,
            heat_capacity_deep=100.0)


def _coupled_builder():
    schema = core.VariableSchema()
    for n, u in (("Emissions|CO2|Anthropogenic", "GtC / yr"), ("Surface Temperature", "K"),
                 ("Atmospheric Concentration|CO2", "ppm"), ("Cumulative Land Uptake", "Gt C"),
                 ("Cumulative Emissions|CO2", "Gt C"), ("Effective Radiative Forcing|CO2", "W/m^2"),
                 ("Deep Ocean Temperature", "K")):
        schema.add_variable(n, u)
    schema.add_aggregate("Effective Radiative Forcing", "W/m^2", "Sum",
                         ["Effective Radiative Forcing|CO2"])
    return (core.ModelBuilder()
            .with_time_axis(core.TimeAxis.from_values(np.arange(1750.0, 1761.0)))
            .with_schema(schema)
            .with_rust_component(CarbonCycleBuilder.from_parameters(
                dict(tau=25.0, conc_pi=278.0, alpha_temperature=0.1)).build())
            .with_rust_component(CO2ERFBuilder.from_parameters(dict(erf_2xco2=3.7, conc_pi=278.0)).build())
            .with_rust_component(TwoLayerBuilder.from_parameters(P_TL).build()))


def test_builder_sources_match_reference_classification():
    """Registration-order rule of builder.rs:470-482 on the notebook's coupled model, and the
    same answer from the oracle's independent restatement."""
    b = _coupled_builder().with_initial_values({
        "Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
        "Atmospheric Concentration|CO2": 278.0, "Surface Temperature": 0.0,
        "Deep Ocean Temperature": 0.0})
    endogenous, sources, exo, aggregates = b._resolve()
    assert sources[("Surface Temperature", "CarbonCycle")] == "Exogenous"
    assert sources[("Atmospheric Concentration|CO2", "CarbonCycle")] == "OwnState"
    assert sources[("Atmospheric Concentration|CO2", "CO2ERF")] == "UpstreamOutput"
    assert sources[("Effective Radiative Forcing", "TwoLayer")] == "UpstreamOutput"
    assert sources[("Surface Temperature", "TwoLayer")] == "OwnState"
    m = rm.ModelBuilder(
        axis=rm.TimeAxis.from_values(np.arange(1750.0, 1761.0)),
        components=[rm.CarbonCycle(25.0, 278.0, 0.1), rm.CO2ERF(3.7, 278.0), rm.TwoLayer(*P_TL.values())],
        aggregates=[("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2"])],
        initial_values={"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
                        "Atmospheric Concentration|CO2": 278.0, "Surface Temperature": 0.0,
                        "Deep Ocean Temperature": 0.0}).build()
    for key, src in m.sources.items():
        if not key[1].startswith("Aggregator"):
            assert sources[key] == src, key
    assert "Emissions|CO2|Anthropogenic" in exo


def test_builder_errors():
    with pytest.raises(ValueError, match="Missing initial value"):
        _coupled_builder()._resolve()
    with pytest.raises(ValueError, match="missing field `eta`"):
        TwoLayerBuilder.from_parameters({k: v for k, v in P_TL.items() if k != "eta"})
    with pytest.raises(NotImplementedError):
        core.ModelBuilder().with_py_component(object())
    class Foreign(core.Component):
        type_name = "Foreign"
        definitions = [("x", "", "Input"), ("y", "", "Output")]
    b = (core.ModelBuilder().with_time_axis(core.TimeAxis.from_values([0.0, 1.0, 2.0]))
         .with_rust_component(Foreign({})))
    with pytest.raises(NotImplementedError, match="has no GPU kernel"):
        b.build()


def test_graph_order_is_the_reference_bfs():
    """ModelBuilder._graph_order (the order linked ensembles are stepped in) against the oracle's
    restatement of builder.rs:448-701 + petgraph Bfs, over every registration order of the notebook's
    three components, with one and with two aggregates."""
    import itertools
    from rscm_amd.components import CarbonCycleBuilder
    make = {"CarbonCycle": lambda: CarbonCycleBuilder.from_parameters(dict(tau=25.0, conc_pi=278.0, alpha_temperature=0.1)).build(),
            "CO2ERF": lambda: CO2ERFBuilder.from_parameters(dict(erf_2xco2=3.7, conc_pi=278.0)).build(),
            "TwoLayer": lambda: TwoLayerBuilder.from_parameters(P_TL).build()}
    ref = {"CarbonCycle": lambda: rm.CarbonCycle(25.0, 278.0, 0.1), "CO2ERF": lambda: rm.CO2ERF(3.7, 278.0),
           "TwoLayer": lambda: rm.TwoLayer(*P_TL.values())}
    init = {"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0, "Atmospheric Concentration|CO2": 278.0,
            "Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}
    aggs = [[("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2"])],
            [("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2", "Effective Radiative Forcing|Other"]),
             ("Diagnostic", "Mean", ["Surface Temperature", "Deep Ocean Temperature"])]]
    for perm in itertools.permutations(make):
        for agg in aggs:
            schema = core.VariableSchema()
            for n in list(init) + ["Effective Radiative Forcing|CO2", "Effective Radiative Forcing|Other", "Emissions|CO2|Anthropogenic"]:
                schema.add_variable(n, "")
            for name, op, contributors in agg:
                schema.add_aggregate(name, "", op, contributors)
            b = core.ModelBuilder().with_time_axis(core.TimeAxis.from_values(np.arange(1750.0, 1756.0))).with_schema(schema)
            for k in perm:
                b.with_rust_component(make[k]())
            b.with_initial_values(init)
            _, _, _, aggregates = b._resolve()
            got = b._graph_order(aggregates)
            m = rm.ModelBuilder(axis=rm.TimeAxis.from_values(np.arange(1750.0, 1756.0)), components=[ref[k]() for k in perm],
                                aggregates=agg, initial_values=init,
                                schema_variables=["Effective Radiative Forcing|Other"]).build()
            want = [m.order_nodes[i].type_name for i in m._bfs() if m.order_nodes[i] is not None]
            assert got == want, (perm, got, want)


def test_graph_order_with_two_components_of_one_type():
    """Several providers of one variable (builder.rs:531-559): every component keeps its node and the edge
    from the earlier provider to the later one; the order is the oracle's breadth-first walk of that graph."""
    from rscm_amd.components import CarbonCycleBuilder
    init = {"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0, "Atmospheric Concentration|CO2": 278.0,
            "Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}
    make = {"CarbonCycle": lambda: CarbonCycleBuilder.from_parameters(dict(tau=25.0, conc_pi=278.0, alpha_temperature=0.1)).build(),
            "CO2ERF": lambda: CO2ERFBuilder.from_parameters(dict(erf_2xco2=3.7, conc_pi=278.0)).build(),
            "TwoLayer": lambda: TwoLayerBuilder.from_parameters(P_TL).build()}
    ref = {"CarbonCycle": lambda: rm.CarbonCycle(25.0, 278.0, 0.1), "CO2ERF": lambda: rm.CO2ERF(3.7, 278.0),
           "TwoLayer": lambda: rm.TwoLayer(*P_TL.values())}
    agg = [("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2"])]
    for seq in (["TwoLayer", "CarbonCycle", "CO2ERF", "TwoLayer"], ["CarbonCycle", "CO2ERF", "TwoLayer", "TwoLayer"],
                ["CO2ERF", "TwoLayer", "CO2ERF", "CarbonCycle"], ["CarbonCycle", "TwoLayer", "CarbonCycle", "CO2ERF", "TwoLayer"]):
        schema = core.VariableSchema()
        for n in list(init) + ["Effective Radiative Forcing|CO2", "Emissions|CO2|Anthropogenic"]:
            schema.add_variable(n, "")
        schema.add_aggregate(agg[0][0], "", agg[0][1], agg[0][2])
        b = core.ModelBuilder().with_time_axis(core.TimeAxis.from_values(np.arange(1750.0, 1756.0))).with_schema(schema)
        for k in seq:
            b.with_rust_component(make[k]())
        b.with_initial_values(init)
        names = b._node_names()
        assert [n.split("#")[0] for n in names] == seq and len(set(names)) == len(names)
        assert all(("#" in n) == (k < len(seq) - 1 - seq[::-1].index(t)) for k, (n, t) in enumerate(zip(names, seq)))
        _, _, _, aggregates = b._resolve()
        got = [n.split("#")[0] for n in b._graph_order(aggregates)]
        m = rm.ModelBuilder(axis=rm.TimeAxis.from_values(np.arange(1750.0, 1756.0)), components=[ref[k]() for k in seq],
                            aggregates=agg, initial_values=init).build()
        want = [m.order_nodes[i].type_name for i in m._bfs() if m.order_nodes[i] is not None]
        assert got == want, (seq, got, want)


def test_topological_order_keeps_launch_classes_together():
    """execution_order="topological" (an extension: every edge forwards) on the emissions-driven MAGICC graph: any
    such order gives the same values, so ties go to the component of the previous one's launch class -- the
    own-kernel components (ClimateUDEB, OceanCarbon) end up adjacent and a step is two fused launches plus those
    two, where the breadth-first position alone cut the light components into three runs."""
    import scripts.bench_magicc_chain as chain
    t, exo, init, contributors = chain.chain_inputs(3, 12)
    schema = core.VariableSchema()
    for n in list(exo) + [k for k in init if k not in ("Surface Temperature", "Effective Radiative Forcing")] + contributors + [
            "Heat Uptake", "Ocean Heat Content", "Sea Surface Temperature", "Carbon Flux|Terrestrial", "Carbon Flux|Ocean",
            "Emissions|CO2|Net", "Airborne Fraction|CO2", "Lifetime|CH4", "Lifetime|N2O"]:
        schema.add_variable(n, "")
    schema.add_variable("Surface Temperature", "K", core.GridType.FourBox)
    schema.add_aggregate("Effective Radiative Forcing", "W/m^2", "Sum", contributors)
    axis = core.TimeAxis.from_values(t)
    b = core.ModelBuilder().with_time_axis(axis).with_schema(schema).with_initial_values(init)
    comps = chain.chain_components()
    for c in comps:
        b.with_rust_component(c)
    for name, vals in exo.items():
        b.with_exogenous_variable(name, core.Timeseries(vals, axis, "", core.InterpolationStrategy.Linear))
    _, _, _, aggregates = b._resolve()
    bfs = b._graph_order(aggregates)
    topo = b._graph_order(aggregates, topological=True)
    assert sorted(bfs) == sorted(topo) and len(topo) == len(comps) + 1
    # every producer before its consumers
    produced_by = {}
    for c in comps:
        for name, _, kind in c.definitions:
            if kind in ("Output", "State"):
                produced_by[name] = c.type_name
    produced_by["Effective Radiative Forcing"] = "Aggregator:Effective Radiative Forcing"
    at = {n: k for k, n in enumerate(topo)}
    for c in comps:
        for name, _, kind in c.definitions:
            if kind == "Input" and name in produced_by and produced_by[name] != c.type_name:
                # (a feedback that closes a cycle is a State of its owner or comes in through the initial values: not an edge)
                assert at[produced_by[name]] < at[c.type_name] or name in init, (name, produced_by[name], c.type_name)
    for name in contributors:
        if name in produced_by:
            assert at[produced_by[name]] < at["Aggregator:Effective Radiative Forcing"]
    own = [n in core.OWN_KERNEL_TYPES for n in topo]
    runs = 1 + sum(own[k] != own[k - 1] for k in range(1, len(own)))
    assert runs == 3 and own.index(True) + 2 == len(own) - own[::-1].index(True), topo   # light ..., UDEB, ocean, light ...
    assert abs(at["ClimateUDEB"] - at["OceanCarbon"]) == 1


def test_priors_and_lhs():
    ps = cal.ParameterSet().add("x", cal.Uniform(0.0, 2.0)).add("y", cal.Uniform(-1.0, 1.0))
    assert ps.param_names == ["x", "y"]
    assert ps.log_prior([1.0, 0.0]) == -math.log(2.0) - math.log(2.0)
    assert ps.log_prior([3.0, 0.0]) == -math.inf
    with pytest.raises(ValueError):
        cal.Uniform(1.0, 1.0)
    n = 257
    s = ps.sample_lhs(n, np.random.default_rng(1))
    assert s.shape == (n, 2)
    for j, (lo, hi) in enumerate(((0.0, 2.0), (-1.0, 1.0))):
        strata = np.floor((s[:, j] - lo) / (hi - lo) * n).astype(int)
        assert np.array_equal(np.sort(strata), np.arange(n))  # parameter_set.rs:207-233
    nrm = cal.ParameterSet().add("z", cal.Normal(1.0, 2.0))
    z = nrm.sample_lhs(4001, np.random.default_rng(2))[:, 0]
    assert abs(z.mean() - 1.0) < 0.01 and abs(z.std() - 2.0) < 0.02


def test_host_likelihood_known_values(known):
    k = known["likelihood"]
    lik = cal.GaussianLikelihood()
    t = cal.Target()
    for tm, v, s in k["perfect"]["obs"]:
        t.add_observation("Temperature", tm, v, s)
    out = {"Temperature": {2020.0: 1.2, 2021.0: 1.3}}
    assert lik.ln_likelihood(out, t) == 0.0
    t = cal.Target().add_observation("Temperature", 2020.0, 1.0, 0.1)
    assert abs(lik.ln_likelihood({"Temperature": {2020.0: 1.1}}, t) + 0.5) < 1e-10
    assert lik.ln_likelihood({"Temperature": {2020.0000001: 1.1}}, t) < 0  # time_key tolerance (:268-275)
    with pytest.raises(KeyError, match="missing time"):
        lik.ln_likelihood({"Temperature": {2021.0: 1.1}}, t)
    with pytest.raises(ValueError, match="non-finite"):
        lik.ln_likelihood({"Temperature": {2020.0: math.inf}}, t)
    with pytest.raises(ValueError):
        cal.Observation(2020.0, 1.0, 0.0)


def test_chain_r_hat_and_flat_samples():
    rng = np.random.default_rng(3)
    c = cal.Chain(["a", "b"], thin=2)
    for _ in range(400):
        c.push(rng.normal(size=(8, 2)), rng.normal(size=8))
    assert c.total_iterations == 400 and len(c) == 200
    assert c.flat_samples(50).shape == (150 * 8, 2)
    assert c.flat_samples(500).shape == (0, 2)
    r = c.r_hat(0)
    assert all(0.95 < v < 1.05 for v in r.values())
    assert c.is_converged()
    # one walker stuck elsewhere -> not converged
    d = cal.Chain(["a"], thin=1)
    for _ in range(100):
        x = rng.normal(size=(4, 1))
        x[0] += 50.0
        d.push(x, np.zeros(4))
    assert not d.is_converged()


def test_stretch_move_z_range():
    """sampler/moves.rs in-file tests: z in [1/a, a]."""
    a = 2.0
    u = np.random.default_rng(4).random(10000)
    z = ((a - 1.0) * u + 1.0) ** 2 / a
    assert z.min() >= 1 / a and z.max() <= a


def test_shard_bounds_partition():
    for n, w in ((10, 3), (100000, 8), (7, 8), (0, 2), (1000001, 4)):
        blocks = [shard_bounds(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0
        assert sum(c for _, c in blocks) == n
        for (o0, c0), (o1, _) in zip(blocks, blocks[1:]):
            assert o0 + c0 == o1
        counts = [c for _, c in blocks]
        assert max(counts) - min(counts) <= 1


def test_config_loader_layers():
    """python/rscm/config/loader.py doctest + tests/test_config_two_layer_integration.py shape."""
    import os
    from rscm_amd import config as cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert cfg.deep_merge({"a": 1, "nested": {"x": 1, "y": 2}}, {"b": 2, "nested": {"y": 3}}) == \
        {"a": 1, "b": 2, "nested": {"x": 1, "y": 3}}
    base = cfg.load_config(os.path.join(root, "configs/two-layer/defaults.toml"))
    assert base["time"] == {"start": 1750, "end": 2100}
    assert base["components"]["climate"]["parameters"]["lambda0"] == 1.0
    merged = cfg.load_config_layers(os.path.join(root, "configs/two-layer/defaults.toml"),
                                    os.path.join(root, "configs/two-layer/tuning/high-ecs.toml"))
    p = merged["components"]["climate"]["parameters"]
    assert p["lambda0"] == 0.7 and p["efficacy"] == 1.3 and p["eta"] == 0.7
    assert merged["model"]["name"] == "two-layer-high-ecs" and merged["model"]["type"] == "two-layer"
    b = cfg.two_layer_builder(merged)
    assert len(b._axis) == 351 and b._axis.at(0) == 1750.0 and b._axis.at(350) == 2100.0
    with pytest.raises(ValueError, match="Unknown model type"):
        cfg.build_model({"model": {"type": "nope"}})


def test_checkpoint_files_round_trip(tmp_path):
    """save_checkpoint / load_checkpoint: a nested checkpoint as plain arrays in one .npz (nothing
    pickled), variable names with '|' and ':' intact, None and NaN preserved."""
    ck = {"time_index": 33, "order": ["A", "Transform:Surface Temperature"],
          "ensembles": {"A": {"kind": 2, "n_members": 4, "bounds": np.arange(5.0), "time_index": 33, "params": np.ones((3, 4)),
                              "state": {"Surface Temperature|NorthernOcean": np.arange(4.0)}, "history": {}, "internal": None},
                        "Transform:Surface Temperature": {"kind": 17, "n_members": 4, "bounds": np.arange(5.0), "time_index": 33,
                                                          "params": np.ones((9, 4)), "state": {"aggregate": np.array([1.0, np.nan, 2.0, 3.0])},
                                                          "history": {"x": np.ones((2, 4))}, "internal": np.arange(7.0)}}}
    core.save_checkpoint(tmp_path / "c.npz", ck)
    back = core.load_checkpoint(tmp_path / "c.npz")

    def same(a, b):
        if isinstance(a, dict):
            return isinstance(b, dict) and set(a) == set(b) and all(same(a[k], b[k]) for k in a)
        if a is None or isinstance(a, list):
            return a == b
        return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)

    assert same(ck, back) and isinstance(back["time_index"], int)
    with pytest.raises(ValueError, match="contains '/'"):
        core.save_checkpoint(tmp_path / "d.npz", {"a/b": 1})


def test_units_that_differ_by_more_than_spelling_are_refused():
    """The reference converts compatible units (builder.rs:141-338); this path has no units registry
    (every factor is 1.0), so a series supplied in another unit is refused, not silently mis-scaled.
    Spelling differences (blanks) are not a mismatch."""
    axis = core.TimeAxis.from_values(np.arange(1750.0, 1756.0))
    def builder(unit):
        return (core.ModelBuilder().with_time_axis(axis)
                .with_rust_component(TwoLayerBuilder.from_parameters(P_TL).build())
                .with_exogenous_variable("Effective Radiative Forcing", core.Timeseries(np.ones(6), axis, unit, core.InterpolationStrategy.Linear))
                .with_initial_values({"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    with pytest.raises(NotImplementedError, match="unit conversion is not available"):
        builder("mW/m^2").build()
    b = builder("W / m^2")
    assert b._exogenous_on_axis("Effective Radiative Forcing", ["Effective Radiative Forcing"]) is not None


def test_toml_writer_round_trips_through_tomli():
    """serialise.dumps / loads: nested tables, arrays of tables, NaN / inf, names with '|' and ':',
    nested number arrays -- what Model.to_toml writes."""
    from rscm_amd import serialise
    doc = {"model": {"format": "rscm_amd-model-1", "graph": True, "n_members": 2, "time_index": 3},
           "time_axis": {"bounds": np.array([1750.0, 1751.0, 1752.5])},
           "components": [{"type": "TwoLayer", "parameters": {"lambda0": 1.0, "a": 0.0}},
                          {"type": "CarbonCycle", "step_size": 0.1, "parameters": {"tau": 25.0}}],
           "initial_values": {"Surface Temperature": 0.0, "Atmospheric Concentration|CO2": 278.0},
           "state": {"ensembles": {"Aggregator:Effective Radiative Forcing": {"kind": 17, "params": [[0.0, 0.0], [1.0, 1.0]],
                                                                               "state": {"aggregate": [float("nan"), float("inf")]}, "history": {}}}}}
    back = serialise.loads(serialise.dumps(doc))
    assert back["model"] == doc["model"] and back["time_axis"]["bounds"] == [1750.0, 1751.0, 1752.5]
    assert back["components"][1] == {"type": "CarbonCycle", "step_size": 0.1, "parameters": {"tau": 25.0}}
    assert back["initial_values"]["Atmospheric Concentration|CO2"] == 278.0
    agg = back["state"]["ensembles"]["Aggregator:Effective Radiative Forcing"]
    assert agg["params"] == [[0.0, 0.0], [1.0, 1.0]] and np.isnan(agg["state"]["aggregate"][0]) and agg["state"]["aggregate"][1] == float("inf")
    assert agg["history"] == {}
    assert set(serialise.component_registry()) >= {"TwoLayer", "CarbonCycle", "CO2ERF", "ClimateUDEB", "GhgForcing", "OceanCarbon"}


def test_as_dot_lists_nodes_and_labelled_edges():
    from rscm_amd import serialise
    dot = serialise.as_dot(_coupled_builder().with_initial_values(
        {"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0, "Atmospheric Concentration|CO2": 278.0,
         "Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}))
    assert dot.startswith("digraph {") and '1 [ label = "CarbonCycle" ]' in dot and '4 [ label = "Aggregator:Effective Radiative Forcing" ]' in dot
    assert '1 -> 2 [ label = "Atmospheric Concentration|CO2" ]' in dot      # CarbonCycle -> CO2ERF
    assert '2 -> 4 [ label = "Effective Radiative Forcing|CO2" ]' in dot    # CO2ERF -> aggregate
    assert '4 -> 3 [ label = "Effective Radiative Forcing" ]' in dot        # aggregate -> TwoLayer
    assert '0 -> 1 [ label = "" ]' in dot                                   # CarbonCycle hangs off the root


def test_variable_schema_api_and_validation():
    """VariableSchema as the reference's Python API exposes it (python/rscm/_lib/core/__init__.pyi:
    322-404): definition objects with name / unit / grid_type / operation_type / contributors /
    weights, and validate() refusing what crates/rscm-core/src/schema.rs refuses."""
    s = core.VariableSchema()
    assert s.variables == {} and s.aggregates == {}
    s.add_variable("Emissions|CO2", "GtCO2/yr").add_variable("Regional Temperature", "K", core.GridType.FourBox)
    v = s.variables["Emissions|CO2"]
    assert (v.name, v.unit, v.grid_type) == ("Emissions|CO2", "GtCO2/yr", core.GridType.Scalar)
    assert s.variables["Regional Temperature"].grid_type == core.GridType.FourBox
    s.add_variable("ERF|CO2", "W/m^2").add_variable("ERF|CH4", "W / m^2")
    s.add_aggregate("Total ERF", "W/m^2", "Sum", ["ERF|CO2", "ERF|CH4"])
    a = s.aggregates["Total ERF"]
    assert (a.name, a.unit, a.operation_type, a.contributors, a.weights) == ("Total ERF", "W/m^2", "Sum", ["ERF|CO2", "ERF|CH4"], None)
    unit, op, contributors, weights = a            # the tuple view the builder uses
    assert (unit, op, contributors, weights) == ("W/m^2", "Sum", ["ERF|CO2", "ERF|CH4"], None)
    s.add_aggregate("Weighted Total", "W/m^2", "Weighted", ["ERF|CO2", "ERF|CH4"], weights=[0.6, 0.4])
    assert s.aggregates["Weighted Total"].weights == [0.6, 0.4]
    assert s.contains("ERF|CO2") and s.contains("Total ERF") and not s.contains("Nonexistent")
    s.validate()
    core.VariableSchema().validate()
    with pytest.raises(ValueError, match="weights must be provided"):
        s.add_aggregate("W", "units", "Weighted", ["ERF|CO2"])
    with pytest.raises(ValueError, match="Unknown operation"):
        s.add_aggregate("Bad", "units", "Invalid", ["ERF|CO2"])

    def failing(build, match):
        sch = core.VariableSchema()
        build(sch)
        with pytest.raises(ValueError, match=match):
            sch.validate()

    failing(lambda x: x.add_variable("A", "u").add_aggregate("T", "u", "Sum", ["A", "B"]), "Undefined contributor")
    failing(lambda x: x.add_variable("A", "W/m^2").add_variable("B", "GtCO2/yr").add_aggregate("T", "W/m^2", "Sum", ["A", "B"]), "Unit mismatch")
    failing(lambda x: x.add_variable("G", "K").add_variable("R", "K", core.GridType.FourBox).add_aggregate("T", "K", "Sum", ["G", "R"]),
            "Grid type mismatch")
    failing(lambda x: x.add_variable("A", "u").add_variable("B", "u").add_variable("C", "u")
            .add_aggregate("T", "u", "Weighted", ["A", "B", "C"], weights=[0.5, 0.5]), "Weight count mismatch")
    failing(lambda x: x.add_aggregate("A", "u", "Sum", ["B"]).add_aggregate("B", "u", "Sum", ["A"]), "Circular dependency")
    with pytest.raises(ValueError, match="sum to 1.0"):
        core.ModelBuilder().with_grid_weights(core.GridType.FourBox, [0.5, 0.5, 0.5, 0.5])
    with pytest.raises(ValueError, match="does not match FourBox grid size"):
        core.ModelBuilder().with_grid_weights(core.GridType.FourBox, [0.5, 0.5])


def test_python_component_declarations_registry_and_outputs():
    """rscm_amd.component mirrors the behaviour the reference's tests/test_typed_python_component.py
    expects of rscm.component: definitions from the declarations, generated Inputs / Outputs with
    validation, the class registry with opt-out, inheritance of declarations."""
    from rscm_amd.component import Component, Input, Output, PythonComponent, State, TimeseriesWindow
    Component._registry.clear()

    class Cycle(Component):
        emissions = Input("Emissions|CO2", unit="GtCO2")
        concentration = State("Atmospheric Concentration|CO2", unit="ppm")
        uptake = Output("Carbon Uptake", unit="GtC")

        def __init__(self, sensitivity):
            self.sensitivity = sensitivity

        def solve(self, t_current, t_next, inputs):
            e = inputs.emissions.at_start()
            return self.Outputs(concentration=inputs.concentration.at_start() + e * self.sensitivity, uptake=e * 0.5)

    class Hidden(Component, register=False):
        value = Output("Value", unit="")

    class Derived(Cycle, register=False):
        extra = Output("Extra", unit="")

    defs = Cycle(0.5).definitions()
    assert {d.name for d in defs} == {"Emissions|CO2", "Atmospheric Concentration|CO2", "Carbon Uptake"} and len(defs) == 3
    assert {d.name: d.requirement_type for d in defs}["Atmospheric Concentration|CO2"] == "State"
    assert Component.get_registered_components() == {"Cycle": Cycle} and Component.get_component("Cycle") is Cycle
    with pytest.raises(KeyError, match="No component registered with name"):
        Component.get_component("Hidden")
    assert isinstance(Cycle(0.5).Outputs(concentration=1.0, uptake=2.0), Cycle.Outputs)
    with pytest.raises(TypeError, match="Missing required output fields: uptake"):
        Cycle(0.5).Outputs(concentration=1.0)
    assert {d.name for d in Derived(1.0).definitions()} == {"Emissions|CO2", "Atmospheric Concentration|CO2", "Carbon Uptake", "Extra"}
    py = PythonComponent.build(Cycle(0.5))
    assert py.type_name == "Cycle" and py.definitions == [("Emissions|CO2", "GtCO2", "Input"), ("Carbon Uptake", "GtC", "Output"),
                                                          ("Atmospheric Concentration|CO2", "ppm", "State")]
    series = {"Emissions|CO2": np.array([[10.0], [10.0], [10.0]]), "Atmospheric Concentration|CO2": np.array([[280.0], [np.nan], [np.nan]])}
    out = py.solve_member(2020.0, 2021.0, series, 0, 0, {})
    assert out == {"Atmospheric Concentration|CO2": 285.0, "Carbon Uptake": 5.0}
    w = TimeseriesWindow(np.array([1.0, 2.0, 3.0]), 1, "UpstreamOutput")
    assert (w.at_start(), w.at_end(), w.get(), w.previous, w.current, w.at_offset(-1), w.at_offset(5)) == (2.0, 3.0, 3.0, 1.0, 2.0, 1.0, None)
    with pytest.raises(ValueError, match="no previous value"):
        TimeseriesWindow(np.array([1.0, 2.0]), 0, "Exogenous").previous
    assert TimeseriesWindow(np.array([1.0, 2.0]), 1, "UpstreamOutput").get() == 2.0   # at_end() is None at the last index
    with pytest.raises(TypeError, match="PythonComponent.build takes"):
        PythonComponent.build(object())
