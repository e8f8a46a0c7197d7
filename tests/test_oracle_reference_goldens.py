"""Pin the oracle (C restatement + Python stepper restatement) against every known-answer
test the reference holds for the hot path.  Data lives in
tests/golden/reference_known_answers.json with reference file:line citations."""
import math

import numpy as np
import pytest

from oracle import cbind
from oracle import reference_model as rm


# ------------------------------------------------------------------ time axis
def test_time_axis_doctests(known):
    k = known["time_axis"]
    ta = rm.TimeAxis.from_values(k["from_values"])
    assert list(ta.at_bounds(2)) == k["at_bounds_2"]
    assert ta.at(1) == k["at_1"] and ta.at(27) is None
    assert ta.contains(1.0) is k["contains_1"] and ta.contains(27.0) is k["contains_27"]
    assert ta.index_of(2.0) == k["index_of_2"] and ta.index_of(27.0) is None
    assert len(rm.TimeAxis.from_bounds(k["from_bounds"])) == k["from_bounds_len"]
    # C restatement agrees
    assert cbind.bounds_from_values(k["from_values"]).tolist() == ta.bounds


def test_time_axis_rejects_non_monotonic():
    # timeseries.rs:869-873 (#[should_panic] check_monotonic_values)
    with pytest.raises(AssertionError):
        rm.TimeAxis.from_values([2020.0, 1.0, 2021.0])
    with pytest.raises(ValueError):
        cbind.bounds_from_values([2020.0, 1.0, 2021.0])


# ------------------------------------------------------------------ interpolation
def test_linear_table(known):
    k = known["interp_linear"]
    for t, e in zip(k["targets"], k["expected"]):
        assert math.isclose(rm.interpolate("Linear", k["time"], k["y"], t, False), e, rel_tol=1e-9)
    for t, e in zip(k["extrap_targets"], k["extrap_expected"]):
        assert math.isclose(rm.interpolate("Linear", k["time"], k["y"], t, True), e, rel_tol=1e-9)
    for t in k["noextrap_error_targets"]:
        with pytest.raises(ValueError, match="Extrapolation is not allowed"):
            rm.interpolate("Linear", k["noextrap_time"], k["noextrap_y"], t, False)


def test_previous_table(known):
    k = known["interp_previous"]
    for t, e in zip(k["targets"], k["expected"]):
        assert rm.interpolate("Previous", k["time"], k["y"], t, False) == e
    for t, e in zip(k["extrap_targets"], k["extrap_expected"]):
        assert rm.interpolate("Previous", k["time"], k["y"], t, True) == e


def test_next_extrapolate(known):
    k = known["interp1d_next_extrapolate"]
    assert rm.interpolate("Next", k["years"], k["data"], k["query"], True) == k["expected"]


def test_timeseries_at_time(known):
    k = known["timeseries_at_time"]
    for q, e in zip(k["linear_queries"], k["linear_expected"]):
        assert rm.interpolate("Linear", k["linear_years"], k["linear_values"], q, False) == e
    with pytest.raises(ValueError):
        rm.interpolate("Linear", k["linear_years"], k["linear_values"],
                       k["linear_noextrap_error_query"], False)
    assert rm.interpolate("Linear", k["custom_years"], k["custom_data"], k["custom_query"],
                          True) == k["custom_linear_expected"]
    assert rm.interpolate("Previous", k["custom_years"], k["custom_data"], k["custom_query"],
                          True) == k["custom_previous_expected"]


# ------------------------------------------------------------------ stepper index conventions
class _Producer(rm.Component):
    type_name = "TemperatureProducer"
    defs = [rm.Req("Surface Temperature", rm.STATE)]

    def __init__(self, rate):
        self.rate = rate

    def solve(self, t0, t1, w):
        return {"Surface Temperature": w["Surface Temperature"].at_start() + self.rate}


class _Consumer(rm.Component):
    type_name = "TemperatureConsumer"
    defs = [rm.Req("Surface Temperature", rm.INPUT), rm.Req("Ocean Heat Content", rm.OUTPUT)]

    def __init__(self, cap):
        self.cap = cap

    def solve(self, t0, t1, w):
        return {"Ocean Heat Content": w["Surface Temperature"].get() * self.cap}


class _TestComponent(rm.Component):
    type_name = "TestComponent"
    defs = [rm.Req("Emissions|CO2", rm.INPUT), rm.Req("Concentrations|CO2", rm.OUTPUT)]

    def __init__(self, f):
        self.f = f

    def solve(self, t0, t1, w):
        return {"Concentrations|CO2": w["Emissions|CO2"].get() * self.f}


class _Constant(rm.Component):
    type_name = "ConstantComponent"
    defs = [rm.Req("TestOutput", rm.OUTPUT)]

    def __init__(self, v):
        self.v = v

    def solve(self, t0, t1, w):
        return {"TestOutput": self.v}


def test_producer_consumer_sources_and_values(known):
    k = known["stepper_producer_consumer"]
    m = rm.ModelBuilder(axis=rm.TimeAxis.from_values(k["time_values"]),
                        components=[_Producer(k["warming_rate"]), _Consumer(k["heat_capacity"])],
                        initial_values={"Surface Temperature": k["initial_surface_temperature"]}
                        ).build()
    for key, src in k["sources"].items():
        var, comp = key.split("|")
        assert m.sources[(var, comp)] == src
    for _ in range(k["steps"]):
        m.step()
    assert m.data["Surface Temperature"][:3] == k["surface_temperature_0_1_2"]
    assert math.isnan(m.data["Ocean Heat Content"][0])
    assert m.data["Ocean Heat Content"][1:3] == k["heat_content_1_2"]


def test_exogenous_previous_and_one_step(known):
    k = known["stepper_exogenous_previous"]
    exo = rm.ExoSeries(k["emissions_values"], rm.TimeAxis.from_bounds(k["emissions_bounds"]),
                       k["emissions_strategy"])
    m = rm.ModelBuilder(axis=rm.TimeAxis.from_values(k["time_values"]),
                        components=[_TestComponent(k["conversion_factor"])],
                        exogenous={"Emissions|CO2": exo}).build()
    assert m.sources[("Emissions|CO2", "TestComponent")] == k["source_emissions"]
    assert m.data["Emissions|CO2"] == k["resampled_emissions"]
    assert m.var_type == {"Emissions|CO2": "Exogenous", "Concentrations|CO2": "Endogenous"}
    m.step()
    assert m.time_index == k["after_one_step_time_index"]
    got = m.data["Concentrations|CO2"]
    for g, e in zip(got, k["after_one_step_concentrations"]):
        assert (math.isnan(g) and math.isnan(e)) or g == e
    assert list(m.axis.at_bounds(m.time_index)) == k["after_one_step_bounds"]
    m.step()
    assert m.axis.at(m.time_index) == k["after_two_steps_current_time"]
    m.run()
    assert m.finished()
    c = m.data["Concentrations|CO2"]
    assert math.isnan(c[0]) and all(not math.isnan(x) for x in c[1:])


def test_model_runner_constant_component(known):
    k = known["model_runner_constant"]
    m = rm.ModelBuilder(axis=rm.TimeAxis.from_values(k["time_values"]),
                        components=[_Constant(k["value"])]).build()
    m.run()
    out = rm.extract_outputs(m, ["TestOutput"])["TestOutput"]
    assert k["missing_key"] not in out
    for key, v in k["present"].items():
        assert out[key] == v


def test_missing_initial_value_for_state():
    # builder.rs:704-717
    with pytest.raises(ValueError, match="MissingInitialValue"):
        rm.ModelBuilder(axis=rm.TimeAxis.from_values([0.0, 1.0, 2.0]),
                        components=[_Producer(0.1)]).build()


# ------------------------------------------------------------------ physics known answers
def _p6(d):
    return [d["lambda0"], d["a"], d["efficacy"], d["eta"], d["heat_capacity_surface"],
            d["heat_capacity_deep"]]


def test_two_layer_reference_properties(known):
    k = known["two_layer_properties"]
    p = _p6(k["params"])

    def solve(erf):
        return cbind.two_layer_solve(p, erf, k["t0"], k["t1"], k["step"], 0.0, 0.0)[0]

    t = solve(k["erf_positive"])
    assert 0.0 < t < k["positive_upper_bound"]
    assert abs(solve(0.0)) < k["erf_zero_abs_tol"]
    assert solve(k["erf_negative"]) < 0.0
    small, large = solve(k["ratio_erf_small"]), solve(k["ratio_erf_large"])
    assert large > small
    assert abs(large / small - k["ratio_expected"]) < k["ratio_tol"]


def test_co2_erf_exact_points(known):
    k = known["co2_erf"]
    for c, e in k["points"]:
        assert abs(cbind.co2_erf(k["erf_2xco2"], k["conc_pi"], c) - e) < k["abs_tol"]
        assert rm.CO2ERF(k["erf_2xco2"], k["conc_pi"]).calculate_erf(c) == \
            cbind.co2_erf(k["erf_2xco2"], k["conc_pi"], c)


def test_carbon_cycle_analytic(known):
    """coupled_models.rs:13-141, run through the generic Python stepper AND the C solve."""
    k = known["carbon_cycle_analytic"]
    h = k["step_size_num"] / k["step_size_den"]
    axis = rm.TimeAxis.from_values(np.arange(k["t_initial"], k["t_end_exclusive"], 1.0))
    cc = rm.CarbonCycle(k["tau"], k["conc_pi"], k["alpha_temperature"], step=h)
    m = rm.ModelBuilder(
        axis=axis, components=[cc],
        initial_values={"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
                        "Atmospheric Concentration|CO2": k["conc_initial"]},
        exogenous={
            "Emissions|CO2|Anthropogenic": rm.ExoSeries(
                k["emissions_values"], rm.TimeAxis.from_bounds(k["emissions_bounds"]),
                k["emissions_strategy"]),
            "Surface Temperature": rm.ExoSeries(
                k["temperature_values"], rm.TimeAxis.from_bounds(k["temperature_bounds"]),
                k["temperature_strategy"]),
        }).build()
    m.run()
    tau, cpi, c0, t0 = k["tau"], k["conc_pi"], k["conc_initial"], k["t_initial"]

    def before(t):
        return (c0 - cpi) * math.exp(-(t - t0) / tau) + cpi

    def after(t):
        return k["emissions_level"] / k["gtc_per_ppm"] * tau * \
            (1.0 - math.exp(-(t - k["step_year"]) / tau)) + before(t)

    exp_e = [0.0 if t < k["step_year"] else k["emissions_level"] for t in axis.values()]
    assert m.data["Emissions|CO2|Anthropogenic"] == exp_e  # assert_eq! in the reference
    conc = m.data["Atmospheric Concentration|CO2"]
    for t, a in zip(axis.values(), conc):
        e = before(t) if t < k["step_year"] else after(t)
        assert abs(a - e) / abs(e) < k["rel_tol"]
    # C restatement, same step: bit-identical to the Python stepper
    y = np.array([c0, 0.0, 0.0])
    for n in range(len(axis) - 1):
        y = cbind.carbon_cycle_solve([tau, cpi, k["alpha_temperature"]], exp_e[n],
                                     k["temperature_value"], axis.bounds[n], axis.bounds[n + 1],
                                     h, y)
        assert y[0] == conc[n + 1]
        assert y[1] == m.data["Cumulative Land Uptake"][n + 1]
        assert y[2] == m.data["Cumulative Emissions|CO2"][n + 1]


def test_aggregate_doctest(known):
    for c in known["aggregate"]["cases"]:
        assert rm.compute_aggregate(c["values"], c["op"], c.get("weights")) == c["expected"]
        if c["op"] == "Sum":
            assert cbind.aggregate_sum(np.array(c["values"], dtype=float)) == c["expected"]
    assert math.isnan(rm.compute_aggregate([math.nan, math.nan], "Sum"))
    assert math.isnan(cbind.aggregate_sum(np.array([math.nan])))


def test_likelihood_known_values(known):
    k = known["likelihood"]
    for name in ("perfect", "one_sigma"):
        c = k[name]
        obs = np.array(c["obs"])
        series = np.array(c["model"], dtype=float).reshape(-1, 1)  # [T][N=1]
        ll = cbind.gaussian_loglik([series], np.zeros(len(obs), np.int32),
                                   np.arange(len(obs), dtype=np.int32), obs[:, 1], obs[:, 2])[0]
        if "abs_tol" in c:
            assert abs(ll - c["expected"]) < c["abs_tol"]
        else:
            assert ll == c["expected"]


def test_likelihood_nonfinite_member_is_minus_inf():
    # likelihood.rs:216-221 -> Err; ensemble.rs:163-172 -> -inf
    series = np.array([[1.0, np.nan], [1.0, 1.0]])
    ll = cbind.gaussian_loglik([series], np.zeros(2, np.int32), np.array([0, 1], np.int32),
                               np.array([1.0, 1.0]), np.array([0.1, 0.1]))
    assert ll[0] == 0.0 and ll[1] == -math.inf


def test_rk4_step_counts(known):
    k = known["rk4_step_counts"]
    for t0, t1, h, n in k["cases"]:
        assert cbind.rk4_nsteps(t0, t1, h) == n == rm.rk4_nsteps(t0, t1, h)
    t0, t1, den = k["one_over_120"]
    assert cbind.rk4_nsteps(t0, t1, 1.0 / den) == den
    # ivp/mod.rs:90-102: a 1/12-yr model step with h = 0.1 ends 0.0167 yr late -> rejected
    assert cbind.rk4_endtime_ok(1750.0, 1751.0, 0.1)
    assert not cbind.rk4_endtime_ok(0.0, 1.0 / 12.0, 0.1)
