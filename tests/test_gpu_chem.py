"""CH4Chemistry and N2OChemistry on the GPU (csrc/chem.hip through the C ABI) against the CPU
oracle (oracle/chem_oracle.c).

Tolerance: |gpu - oracle| <= 1e-11 * max(1, |oracle|).  Each model step evaluates four f64 pow (and
for CH4 one exp) from the device math library, <= 1-2 ulp from glibc's, on concentrations of
O(1e3) ppb; the recurrence is contractive (lifetimes of 10-140 years), so the differences do not
grow over the run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-11


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


def _gpu(ra, kind, bounds, P, inputs, c0, scen=None, chunks=()):
    with ra.Ensemble(kind, P.shape[1], bounds) as e:
        e.set_params(P)
        e.set_forcing(inputs, scen)
        e.set_initial(1, c0)
        for c in chunks:
            e.run(c)
        e.run()
        assert not e.status().any()
        return e.get_series(1), e.get_series(2)


def _close(got, want, what):
    assert (np.isnan(got) == np.isnan(want)).all(), what
    ok = ~np.isnan(want)
    err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
    assert err.max() <= TOL, f"{what}: max deviation {err.max():.3e}"


def _ch4_case(orc, n, T, rng):
    yr = np.arange(T, dtype=float)
    inputs = np.stack([np.stack([150.0 + 0.8 * yr, 0.008 * yr, 5.0 + 0.1 * yr, 200.0 + yr, 50.0 + 0.2 * yr]),
                       np.stack([400.0 - 0.5 * yr, np.sin(yr / 9.0), 40.0 - 0.05 * yr, 600.0 - yr, 120.0 - 0.1 * yr])])
    P = np.repeat(orc.chem_default_params(orc.CHEM_CH4).reshape(-1, 1), n, axis=1)
    names = orc.CHEM_PARAM_NAMES[orc.CHEM_CH4]
    for k, (lo, hi) in dict(tau_oh=(8.0, 11.0), ch4_self_feedback=(-0.4, -0.2), oh_sensitivity_scale=(0.5, 0.9),
                            natural_emissions=(180.0, 240.0), temp_sensitivity=(0.02, 0.04), tau_soil=(120.0, 180.0)).items():
        P[names.index(k)] = rng.uniform(lo, hi, n)
    P[names.index("include_temp_feedback")] = (np.arange(n) % 3 != 0).astype(float)
    P[names.index("include_emissions_feedback")] = (np.arange(n) % 4 != 1).astype(float)
    return P, inputs, rng.uniform(650.0, 900.0, n)


def _n2o_case(orc, n, T, rng):
    yr = np.arange(T, dtype=float)
    inputs = np.stack([(0.02 * yr)[None], (8.0 - 0.01 * yr)[None]])
    P = np.repeat(orc.chem_default_params(orc.CHEM_N2O).reshape(-1, 1), n, axis=1)
    names = orc.CHEM_PARAM_NAMES[orc.CHEM_N2O]
    for k, (lo, hi) in dict(tau_n2o=(110.0, 160.0), lifetime_feedback=(-0.08, 0.0), natural_emissions=(9.0, 13.0)).items():
        P[names.index(k)] = rng.uniform(lo, hi, n)
    P[names.index("strat_delay")] = (np.arange(n) % 5).astype(float)  # 0 (treated as 1), 1, 2, 3, 4
    return P, inputs, rng.uniform(265.0, 300.0, n)


@pytest.mark.parametrize("kind_name", ["CHEM_CH4", "CHEM_N2O"])
@pytest.mark.parametrize("n", [1, 63, 1000])
def test_chem_gpu_vs_oracle(ra, orc, kind_name, n):
    kind = getattr(orc, kind_name)
    rng = np.random.default_rng(31 * kind + n)
    T = 301
    bounds = np.concatenate([[1750.0], 1750.0 + np.cumsum(np.where(np.arange(T) % 7 == 3, 0.5, 1.0))])  # uneven steps
    P, inputs, c0 = (_ch4_case if kind == orc.CHEM_CH4 else _n2o_case)(orc, n, T, rng)
    scen = (np.arange(n) % 2).astype(np.int32)
    wc, wl = orc.chem_run(kind, bounds, P, inputs, c0, scen=scen, threads=8)
    gc, gl = _gpu(ra, kind, bounds, P, inputs, c0, scen=scen)
    assert np.array_equal(gc[0], c0) and np.isnan(gl[0]).all()
    _close(gc, wc, f"{kind_name} n={n} concentration")
    _close(gl, wl, f"{kind_name} n={n} lifetime")
    # resume: the launch boundaries fall inside the lag window of the N2O delay
    rc, rl = _gpu(ra, kind, bounds, P, inputs, c0, scen=scen, chunks=(1, 3, 120))
    assert np.array_equal(rc, gc, equal_nan=True) and np.array_equal(rl, gl, equal_nan=True)
    g0c, _ = _gpu(ra, kind, bounds, P, inputs[:1], c0)
    w0c, _ = orc.chem_run(kind, bounds, P, inputs[:1], c0)
    _close(g0c, w0c, f"{kind_name} n={n} one scenario")


def test_chem_through_the_reference_shaped_front(ra, orc):
    from rscm_amd import core
    from rscm_amd.magicc import CH4ChemistryBuilder, N2OChemistryBuilder
    years = np.arange(1900.0, 1961.0)
    axis = core.TimeAxis.from_bounds(np.append(years, 1961.0))
    T = len(years)
    ts = lambda v: core.Timeseries(v, axis, "", core.InterpolationStrategy.Previous)  # noqa: E731
    ch4_in = {"Emissions|CH4": 200.0 + 3.0 * np.arange(T), "Surface Temperature": 0.01 * np.arange(T),
              "Emissions|NOx": np.full(T, 20.0), "Emissions|CO": np.full(T, 400.0), "Emissions|NMVOC": np.full(T, 90.0)}
    b = core.ModelBuilder().with_time_axis(axis).with_rust_component(CH4ChemistryBuilder.from_parameters({"tau_oh": 9.6}).build())
    for k, v in ch4_in.items():
        b = b.with_exogenous_variable(k, ts(v))
    with pytest.raises(ValueError, match="Missing initial value"):
        b.build()
    m = b.with_initial_values({"Atmospheric Concentration|CH4": 900.0}).build()
    m.run()
    res = m.timeseries()
    m.close()
    wc, wl = orc.chem_run(orc.CHEM_CH4, np.append(years, 1961.0), orc.chem_default_params(orc.CHEM_CH4, tau_oh=9.6),
                          np.stack(list(ch4_in.values())), 900.0)
    _close(res.get_timeseries_by_name("Atmospheric Concentration|CH4").values(), wc[:, 0], "front CH4")
    _close(res.get_timeseries_by_name("Lifetime|CH4").values(), wl[:, 0], "front CH4 lifetime")

    e = 0.1 * np.arange(T)
    m = (core.ModelBuilder().with_time_axis(axis)
         .with_rust_component(N2OChemistryBuilder.from_parameters({"strat_delay": 3}).build())
         .with_exogenous_variable("Emissions|N2O", ts(e))
         .with_initial_values({"Atmospheric Concentration|N2O": 285.0}).build())
    m.run()
    res = m.timeseries()
    m.close()
    wc, wl = orc.chem_run(orc.CHEM_N2O, np.append(years, 1961.0), orc.chem_default_params(orc.CHEM_N2O, strat_delay=3), e[None, None, :], 285.0)
    _close(res.get_timeseries_by_name("Atmospheric Concentration|N2O").values(), wc[:, 0], "front N2O")
    with pytest.raises(ValueError, match="strat_delay"):
        N2OChemistryBuilder.from_parameters({"strat_delay": 1.5})


def test_chem_full_size_properties(ra, orc):
    """1e6 members x 751 years: with the feedbacks off the CH4 update is linear in the burden, so the
    concentration relaxes towards E_total * tau / ppb_to_tg whatever the start; members with equal
    parameters agree exactly; sampled members match the oracle."""
    n, T = 1_000_000, 751
    rng = np.random.default_rng(5)
    bounds = np.arange(T + 1, dtype=float) + 1750.0
    inputs = np.stack([np.full(T, 300.0), np.zeros(T), np.zeros(T), np.zeros(T), np.zeros(T)])
    P = np.repeat(orc.chem_default_params(orc.CHEM_CH4, ch4_self_feedback=0.0, include_temp_feedback=0.0,
                                          include_emissions_feedback=0.0).reshape(-1, 1), n, axis=1)
    P[2] = rng.uniform(8.0, 11.0, n)  # tau_oh
    P[2, n // 2:] = P[2, : n // 2]
    c0 = np.tile(rng.uniform(500.0, 2500.0, n // 2), 2)
    with ra.Ensemble(ra.KIND_CH4_CHEMISTRY, n, bounds) as e:
        e.set_params(P)
        e.set_forcing(inputs)
        e.set_initial(1, c0)
        e.run()
        last = e.get_series(1, T - 1, T)[0]
        mid = e.get_series(1, 100, 101)[0]
        life = e.get_series(2, T - 1, T)[0]
    k = 1.0 / P[2] + (1 / 150.0 + 1 / 120.0 + 1 / 200.0)
    assert np.abs(last - (300.0 + 209.0) / k / 2.75).max() < 1e-6  # equilibrium E/k, reached long before 2500
    assert np.abs(life - 1.0 / k).max() < 1e-12
    assert np.array_equal(last[: n // 2], last[n // 2:]) and np.array_equal(mid[: n // 2], mid[n // 2:])
    pick = rng.choice(n, 32, replace=False)
    wc, _ = orc.chem_run(orc.CHEM_CH4, bounds, P[:, pick].copy(), inputs, c0[pick])
    assert np.abs(mid[pick] - wc[100]).max() <= TOL * 3000.0


def _close_any(got, want, what, tol=TOL):
    """Tolerance on the finite entries; infinities and NaNs must sit in the same places."""
    assert (np.isnan(got) == np.isnan(want)).all(), f"{what}: NaN placement"
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf]), f"{what}: infinities"
    ok = np.isfinite(want)
    err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
    assert not err.size or err.max() <= tol, f"{what}: max deviation {err.max():.3e}"


@pytest.mark.parametrize("kind_name", ["CHEM_CH4", "CHEM_N2O"])
def test_chem_infinite_and_zero_lifetimes(ra, orc, kind_name):
    """Lifetimes of +inf (a sink switched off: x / inf = 0 in the reference) and of 0 (x / 0 = inf):
    the hoisted reciprocals must give what IEEE division gives (rcp(inf) = 0 and rcp(0) = inf would
    otherwise turn the refinement step into NaN)."""
    kind = getattr(orc, kind_name)
    rng = np.random.default_rng(77)
    n, T = 64, 41
    bounds = np.arange(T + 1, dtype=float) + 1850.0
    P, inputs, c0 = (_ch4_case if kind == orc.CHEM_CH4 else _n2o_case)(orc, n, T, rng)
    names = orc.CHEM_PARAM_NAMES[kind]
    inf = float("inf")
    if kind == orc.CHEM_CH4:
        for m in range(0, 16):   # no soil / stratospheric / chlorine sink at all: tau_other = inf
            for k in ("tau_soil", "tau_strat", "tau_trop_cl"):
                P[names.index(k), m] = inf
        P[names.index("tau_soil"), 16:24] = inf          # one sink off, the others on
        P[names.index("tau_oh"), 24:32] = inf            # no OH sink
        P[names.index("tau_strat"), 32:36] = 0.0         # instantaneous sink: burden / 0
        P[names.index("tau_oh"), 36:40] = 0.0
    else:
        P[names.index("tau_n2o"), 0:16] = inf
        P[names.index("tau_n2o"), 16:24] = 0.0
    wc, wl = orc.chem_run(kind, bounds, P, inputs[:1], c0)
    with ra.Ensemble(kind, n, bounds) as e:
        e.set_params(P)
        e.set_forcing(inputs[:1])
        e.set_initial(1, c0)
        e.run()
        gc, gl = e.get_series(1), e.get_series(2)
    assert np.isfinite(wc[:, :16]).all()  # the switched-off sinks give ordinary trajectories
    _close_any(gc, wc, f"{kind_name} concentration")
    _close_any(gl, wl, f"{kind_name} lifetime")
