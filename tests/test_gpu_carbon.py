"""CO2Budget and TerrestrialCarbon on the GPU (csrc/carbon.hip through the C ABI) against the CPU
oracle (oracle/carbon_oracle.c).

CO2Budget has no transcendental and must match bit for bit.  TerrestrialCarbon evaluates a log
and five exp per step from the device math library: |gpu - oracle| <= 1e-11 * max(1, |oracle|) on
pools of O(1e3) GtC; the implicit pool update is contractive."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-11
PI_POOLS = np.array([884.86, 92.77, 1681.53, 836.0])


@pytest.fixture(scope="module")
def ra():
    import rscm_amd
    from rscm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1
    return rscm_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import cbind
    return cbind


def _gpu(ra, kind, bounds, P, inputs, initial, scen=None, chunks=()):
    with ra.Ensemble(kind, P.shape[1], bounds) as e:
        e.set_params(P)
        e.set_forcing(inputs, scen)
        for v, x in enumerate(initial, start=1):
            e.set_initial(v, x)
        for c in chunks:
            e.run(c)
        e.run()
        assert not e.status().any()
        return np.stack([e.get_series(v) for v in sorted(v for v in e.var_ids.values() if v > 0)])


def _bounds(T):
    return np.concatenate([[1750.0], 1750.0 + np.cumsum(np.where(np.arange(T) % 5 == 2, 0.25, 1.0))])


@pytest.mark.parametrize("n", [1, 63, 1000])
def test_co2_budget_gpu_bit_exact(ra, orc, n):
    rng = np.random.default_rng(n)
    T = 301
    b = _bounds(T)
    yr = np.arange(T, dtype=float)
    inputs = np.stack([np.stack([0.03 * yr, 1.0 - 0.002 * yr, 0.01 * yr, 0.012 * yr]),
                       np.stack([np.where(yr < 50, 0.0, 5.0), np.where(yr < 50, 0.0, -6.0 + 0.05 * yr), 0.5 + 0 * yr, 0.7 + 0 * yr])])
    P = np.repeat(orc.carbon_default_params(orc.CARBON_BUDGET).reshape(-1, 1), n, axis=1)
    P[0] = rng.uniform(2.0, 2.3, n)
    c0 = rng.uniform(270.0, 420.0, n)
    scen = (np.arange(n) % 2).astype(np.int32)
    want = orc.carbon_run(orc.CARBON_BUDGET, b, P, inputs, [c0], scen=scen, threads=4)
    got = _gpu(ra, ra.KIND_CO2_BUDGET, b, P, inputs, [c0], scen=scen)
    assert np.array_equal(got, want, equal_nan=True)
    assert (want[2, 1:, 1::2] == 0.0).any() or n == 1  # the non-positive-emissions branch occurs
    assert np.array_equal(_gpu(ra, ra.KIND_CO2_BUDGET, b, P, inputs, [c0], scen=scen, chunks=(2, 90)), got, equal_nan=True)


@pytest.mark.parametrize("n", [1, 63, 1000])
def test_terrestrial_gpu_vs_oracle(ra, orc, n):
    rng = np.random.default_rng(7 * n)
    T = 301
    b = _bounds(T)
    yr = np.arange(T, dtype=float)
    inputs = np.stack([np.stack([278.0 * 1.004 ** yr, 0.012 * yr, np.where(yr > 100, 1.5, 0.2)]),
                       np.stack([np.maximum(500.0 - 2.0 * yr, 0.0), 3.0 * np.sin(yr / 15.0), 40.0 + 0 * yr])])
    P = np.repeat(orc.carbon_default_params(orc.CARBON_TERRESTRIAL).reshape(-1, 1), n, axis=1)
    names = orc.CARBON_PARAM_NAMES[orc.CARBON_TERRESTRIAL]
    for k, (lo, hi) in dict(beta=(0.3, 0.9), npp_temp_sensitivity=(0.0, 0.03), soil_temp_sensitivity=(0.08, 0.2),
                            npp_pi=(55.0, 75.0), frac_detritus_to_soil=(0.2, 0.4), respiration_pi=(10.0, 14.0)).items():
        P[names.index(k)] = rng.uniform(lo, hi, n)
    P[names.index("enable_fertilization")] = (np.arange(n) % 3 != 1).astype(float)
    P[names.index("enable_temp_feedback")] = (np.arange(n) % 4 != 2).astype(float)
    init = [PI_POOLS[k] * rng.uniform(0.9, 1.1, n) for k in range(4)]
    scen = (np.arange(n) % 2).astype(np.int32)
    want = orc.carbon_run(orc.CARBON_TERRESTRIAL, b, P, inputs, np.stack(init), scen=scen, threads=8)
    got = _gpu(ra, ra.KIND_TERRESTRIAL_CARBON, b, P, inputs, init, scen=scen)
    assert (np.isnan(got) == np.isnan(want)).all() and np.isnan(got[4, 0]).all()
    ok = ~np.isnan(want)
    err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
    assert err.max() <= TOL, f"max deviation {err.max():.3e}"
    assert (want[:4, 1:] >= 0.0).all()
    if n >= 63:  # scenario 1 (CO2 -> 0, heavy land use) drives the plant pool onto its floor at zero
        assert (want[0, -1, 1::2] == 0.0).any() and np.array_equal(got[0][want[0] == 0.0], want[0][want[0] == 0.0])
    assert np.array_equal(_gpu(ra, ra.KIND_TERRESTRIAL_CARBON, b, P, inputs, init, scen=scen, chunks=(1, 150)), got, equal_nan=True)


def test_carbon_through_the_reference_shaped_front(ra, orc):
    from rscm_amd import core
    from rscm_amd.magicc import CO2BudgetBuilder, TerrestrialCarbonBuilder
    years = np.arange(1950.0, 2011.0)
    axis = core.TimeAxis.from_bounds(np.append(years, 2011.0))
    T = len(years)
    ts = lambda v: core.Timeseries(v, axis, "", core.InterpolationStrategy.Previous)  # noqa: E731
    tc_in = {"Atmospheric Concentration|CO2": 310.0 + 1.5 * np.arange(T), "Surface Temperature": 0.015 * np.arange(T),
             "Emissions|CO2|Land Use": np.full(T, 1.2)}
    b = core.ModelBuilder().with_time_axis(axis).with_rust_component(TerrestrialCarbonBuilder.from_parameters({"beta": 0.5}).build())
    for k, v in tc_in.items():
        b = b.with_exogenous_variable(k, ts(v))
    m = b.with_initial_values({"Carbon Pool|Plant": 884.86, "Carbon Pool|Detritus": 92.77, "Carbon Pool|Soil": 1681.53,
                               "Carbon Pool|Humus": 836.0}).build()
    m.run()
    res = m.timeseries()
    m.close()
    want = orc.carbon_run(orc.CARBON_TERRESTRIAL, np.append(years, 2011.0), orc.carbon_default_params(orc.CARBON_TERRESTRIAL, beta=0.5),
                          np.stack(list(tc_in.values())), PI_POOLS)
    for k, name in enumerate(("Carbon Pool|Plant", "Carbon Pool|Detritus", "Carbon Pool|Soil", "Carbon Pool|Humus", "Carbon Flux|Terrestrial")):
        got = res.get_timeseries_by_name(name).values()
        w = want[k, :, 0]
        assert (np.isnan(got) == np.isnan(w)).all() and np.nanmax(np.abs(got - w) / np.maximum(1.0, np.abs(w))) <= TOL, name
    cb_in = {"Emissions|CO2|Fossil": 2.0 + 0.1 * np.arange(T), "Emissions|CO2|Land Use": np.full(T, 1.0),
             "Carbon Flux|Terrestrial": np.full(T, 1.5), "Carbon Flux|Ocean": 1.0 + 0.02 * np.arange(T)}
    b = core.ModelBuilder().with_time_axis(axis).with_rust_component(CO2BudgetBuilder.from_parameters({}).build())
    for k, v in cb_in.items():
        b = b.with_exogenous_variable(k, ts(v))
    m = b.with_initial_values({"Atmospheric Concentration|CO2": 311.0}).build()
    m.run()
    res = m.timeseries()
    m.close()
    want = orc.carbon_run(orc.CARBON_BUDGET, np.append(years, 2011.0), orc.carbon_default_params(orc.CARBON_BUDGET), np.stack(list(cb_in.values())), [311.0])
    for k, name in enumerate(("Atmospheric Concentration|CO2", "Emissions|CO2|Net", "Airborne Fraction|CO2")):
        assert np.array_equal(res.get_timeseries_by_name(name).values(), want[k, :, 0], equal_nan=True), name


def test_carbon_full_size_properties(ra, orc):
    """1e6 members x 751 years.  CO2Budget: the concentration is the initial value plus the running
    sum of net emissions over gtc_per_ppm -- checked against numpy's cumulative sum to rounding;
    TerrestrialCarbon: the change of the four pools equals the accumulated net flux (the reference's
    tests/conservation.rs property) for every member."""
    n, T = 1_000_000, 751
    rng = np.random.default_rng(9)
    b = np.arange(T + 1, dtype=float) + 1750.0
    yr = np.arange(T, dtype=float)
    cb = np.stack([0.02 * yr, np.full(T, 0.5), 0.004 * yr, 0.006 * yr])
    P = np.repeat(orc.carbon_default_params(orc.CARBON_BUDGET).reshape(-1, 1), n, axis=1)
    P[0] = rng.uniform(2.0, 2.3, n)
    with ra.Ensemble(ra.KIND_CO2_BUDGET, n, b) as e:
        e.set_params(P)
        e.set_forcing(cb)
        e.set_initial(1, 278.0)
        e.run()
        last = e.get_series(1, T - 1, T)[0]
    net = (cb[0] + cb[1]) - (cb[2] + cb[3])
    assert np.abs(last - (278.0 + net[: T - 1].sum() / P[0])).max() < 1e-9

    tc = np.stack([278.0 * 1.001 ** yr, 0.004 * yr, np.full(T, 0.3)])
    Q = np.repeat(orc.carbon_default_params(orc.CARBON_TERRESTRIAL).reshape(-1, 1), n, axis=1)
    Q[2] = rng.uniform(0.3, 0.9, n)
    with ra.Ensemble(ra.KIND_TERRESTRIAL_CARBON, n, b) as e:
        e.set_params(Q)
        e.set_forcing(tc)
        for v in range(4):
            e.set_initial(v + 1, PI_POOLS[v])
        e.run()
        final = sum(e.get_series(v, T - 1, T)[0] for v in (1, 2, 3, 4))
        flux = np.zeros(n)
        for t0 in range(1, T, 125):  # accumulate the flux rows in slabs to bound host memory
            flux += e.get_series(5, t0, min(t0 + 125, T)).sum(axis=0)
    # trapezoidal pools with the plant pool above its floor conserve carbon to rounding
    assert np.abs((final - PI_POOLS.sum()) - flux).max() < 1e-6
    pick = rng.choice(n, 16, replace=False)
    want = orc.carbon_run(orc.CARBON_TERRESTRIAL, b, Q[:, pick].copy(), tc, PI_POOLS)
    assert np.abs(final[pick] - want[:4, T - 1].sum(axis=0)).max() <= TOL * 4000.0


def test_terrestrial_zero_and_infinite_turnover_times(ra, orc):
    """Pools with a zero pre-industrial size (turnover time 0: k = 1 / 0) and with no inflow (the
    reference's fall-back lifetimes), and an infinite pool (turnover time inf: k = 0): the hoisted
    reciprocals must give what IEEE division gives in the reference."""
    rng = np.random.default_rng(3)
    n, T = 64, 41
    b = np.arange(T + 1, dtype=float) + 1850.0
    yr = np.arange(T, dtype=float)
    inputs = np.stack([278.0 * 1.004 ** yr, 0.012 * yr, np.where(yr > 10, 1.5, 0.2)])
    P = np.repeat(orc.carbon_default_params(orc.CARBON_TERRESTRIAL).reshape(-1, 1), n, axis=1)
    names = orc.CARBON_PARAM_NAMES[orc.CARBON_TERRESTRIAL]
    P[names.index("humus_pool_pi"), 0:8] = 0.0
    P[names.index("soil_pool_pi"), 8:16] = 0.0
    P[names.index("plant_pool_pi"), 16:24] = float("inf")
    P[names.index("detritus_pool_pi"), 24:32] = float("inf")
    P[names.index("npp_pi"), 32:40] = 0.0          # no inflow anywhere: the fall-back lifetimes
    init = [PI_POOLS[k] * rng.uniform(0.9, 1.1, n) for k in range(4)]
    want = orc.carbon_run(orc.CARBON_TERRESTRIAL, b, P, inputs, np.stack(init))
    with ra.Ensemble(ra.KIND_TERRESTRIAL_CARBON, n, b) as e:
        e.set_params(P)
        e.set_forcing(inputs)
        for v, x in enumerate(init, start=1):
            e.set_initial(v, x)
        e.run()
        got = np.stack([e.get_series(v) for v in range(1, 6)])
    assert (np.isnan(got) == np.isnan(want)).all()
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf])
    ok = np.isfinite(want)
    err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
    assert err.max() <= TOL, f"max deviation {err.max():.3e}"
    assert np.isfinite(want[:, 1:, 40:]).all() and np.isfinite(want[:4, 0]).all()


def test_terrestrial_member_constants_follow_the_parameters(ra, orc):
    """TerrestrialCarbon's turnover times (parameters/terrestrial_carbon.rs:103-168) are formed once per parameter set on the device
    (launch_terrestrial_derive): a second parameter set on the same handle, uniform rows (scalar loads of element 0) and varied ones
    all give the oracle's pools."""
    rng = np.random.default_rng(3)
    T, n = 120, 130
    b = _bounds(T)
    yr = np.arange(T, dtype=float)
    inputs = np.stack([278.0 * 1.004 ** yr, 0.012 * yr, np.where(yr > 50, 1.5, 0.2)])[None]
    names = orc.CARBON_PARAM_NAMES[orc.CARBON_TERRESTRIAL]
    base = np.repeat(orc.carbon_default_params(orc.CARBON_TERRESTRIAL).reshape(-1, 1), n, axis=1)
    init = np.stack([np.full(n, PI_POOLS[k]) for k in range(4)])

    def varied(seed):
        r = np.random.default_rng(seed)
        P = base.copy()
        for k, (lo, hi) in dict(npp_pi=(55.0, 75.0), plant_pool_pi=(700.0, 1000.0), soil_pool_pi=(1400.0, 1900.0), respiration_pi=(10.0, 14.0),
                                frac_npp_to_plant=(0.3, 0.5), frac_soil_to_humus=(0.0, 0.05)).items():
            if k in names:
                P[names.index(k)] = r.uniform(lo, hi, n)
        return P

    with ra.Ensemble(ra.KIND_TERRESTRIAL_CARBON, n, b) as e:
        e.set_forcing(inputs)
        for v in range(4):
            e.set_initial(v + 1, init[v])
        for what, P in (("varied", varied(1)), ("another set", varied(2)), ("uniform", base), ("varied again", varied(1))):
            e.set_params(P)
            e.rewind()
            e.run()
            got = np.stack([e.get_series(v) for v in range(1, 6)])
            want = orc.carbon_run(orc.CARBON_TERRESTRIAL, b, P, inputs, init, threads=4)
            ok = ~np.isnan(want)
            assert (np.isnan(got) == np.isnan(want)).all(), what
            err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
            assert err.max() <= TOL, f"{what}: max deviation {err.max():.3e}"


@pytest.mark.parametrize("mode", [0, 1])
def test_carbon_cycle_kind_on_its_own_in_both_modes(ra, orc, mode):
    """RSCM_KIND_CARBON_CYCLE stand-alone (table inputs: emissions and a prescribed temperature) against the oracle's single-step
    solve chained over the axis, in RSCM_MODE_EXACT (the reference's expression order) and RSCM_MODE_FAST (the closed-form RK4 step of
    the linear box, carbon_body.hpp): 1e-11 on concentration and cumulative uptake, cumulative emissions bit for bit in either."""
    rng = np.random.default_rng(9)
    T, n = 90, 200
    b = 1750.0 + np.concatenate([[0.0], np.cumsum(np.where(np.arange(T) % 4 == 1, 0.5, 1.0))])   # steps of 1 and 1/2 year: 10 and 5 sub-steps
    E = np.abs(rng.normal(4.0, 2.0, T))
    temp = np.cumsum(rng.normal(0.02, 0.05, T))
    P = np.stack([rng.uniform(15.0, 40.0, n), np.full(n, 278.0), rng.uniform(0.0, 0.1, n)])
    with ra.Ensemble(ra.KIND_CARBON_CYCLE, n, b) as e:
        e.set_mode(mode)
        e.set_params(P)
        e.set_forcing(np.stack([E, temp]))
        for v, x in ((1, 278.0), (2, 0.0), (3, 0.0)):
            e.set_initial(v, x)
        e.run(7)
        e.run()
        got = np.stack([e.get_series(v) for v in (1, 2, 3)])
        assert not e.status().any()
    want = np.empty_like(got)
    want[:, 0] = got[:, 0]
    for i in range(n):
        y = np.array([278.0, 0.0, 0.0])
        for k in range(T - 1):
            y = orc.carbon_cycle_solve(P[:, i], E[k], temp[k], b[k], b[k + 1], 0.1, y)
            want[:, k + 1, i] = y
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err[:2].max() <= 1e-11, f"mode {mode}: {err[:2].max():.3e}"
    assert np.array_equal(got[2].view(np.uint64), want[2].view(np.uint64)), "cumulative emissions"
