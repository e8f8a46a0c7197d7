"""OceanCarbon on the GPU (csrc/ocean.hip through the C ABI) against the CPU oracle
(oracle/ocean_oracle.c).

The kernel keeps the reference's summation order in the monthly convolution and reads the impulse
response from a host-built table holding the same bits as per-call evaluation, so with the
temperature feedback off the results must match BIT FOR BIT.  With it on, exp() comes from the
device math library: |gpu - oracle| <= 1e-12 * max(1, |oracle|)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12


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


def _gpu(ra, bounds, P, inputs, pco2_0, cum_0, scen=None, chunks=()):
    with ra.Ensemble(ra.KIND_OCEAN_CARBON, P.shape[1], bounds) as e:
        e.set_params(P)
        e.set_forcing(inputs, scen)
        e.set_initial(1, pco2_0)
        e.set_initial(2, cum_0)
        for c in chunks:
            e.run(c)
        e.run()
        assert not e.status().any()
        return np.stack([e.get_series(v) for v in (1, 2, 3)])


def _case(orc, model, n, T, rng, **fixed):
    yr = np.arange(T, dtype=float)
    inputs = np.stack([np.stack([278.0 * 1.006 ** yr, 0.01 * yr]),
                       np.stack([np.where(yr < 30, 400.0, 300.0), np.where(yr < 30, 1.0, -0.5)])])
    P = np.repeat(orc.ocean_default_params(model, **fixed).reshape(-1, 1), n, axis=1)
    names = orc.OCEAN_PARAM_NAMES
    for k, (lo, hi) in dict(gas_exchange_tau=(6.0, 10.0), temp_sensitivity=(0.03, 0.045), mixed_layer_depth=(45.0, 80.0),
                            sst_pi=(16.0, 19.0), pco2_pi=(270.0, 285.0), delta_ospp_offsets_0=(1.4, 1.7)).items():
        P[names.index(k)] = rng.uniform(lo, hi, n)
    return P, inputs


@pytest.mark.parametrize("model", ["3D-GFDL", "2D-BERN", "HILDA"])
def test_ocean_gpu_bit_exact_without_temperature_feedback(ra, orc, model):
    rng = np.random.default_rng(17)
    n, T = 65, 61
    b = np.arange(T + 1, dtype=float) + 1750.0
    P, inputs = _case(orc, model, n, T, rng, enable_temp_feedback=0.0)
    scen = (np.arange(n) % 2).astype(np.int32)
    want = orc.ocean_run(b, P, inputs, 278.0, 0.0, scen=scen, threads=8)
    got = _gpu(ra, b, P, inputs, 278.0, 0.0, scen=scen)
    assert np.array_equal(got, want, equal_nan=True)
    # three launches, boundaries inside a year group of the convolution: the same bits
    assert np.array_equal(_gpu(ra, b, P, inputs, 278.0, 0.0, scen=scen, chunks=(1, 17)), got, equal_nan=True)
    # one step per launch (Model::step, linked graphs): the steps are paired up into split two-year
    # tiles -- the same bits again, also when single steps and longer launches alternate
    assert np.array_equal(_gpu(ra, b, P, inputs, 278.0, 0.0, scen=scen, chunks=range(1, T)), got, equal_nan=True)
    mixed = (1, 2, 3, 10, 11, 12, 13, 30, 31, 58, 59)
    assert np.array_equal(_gpu(ra, b, P, inputs, 278.0, 0.0, scen=scen, chunks=mixed), got, equal_nan=True)


@pytest.mark.parametrize("n", [1, 63, 257])
def test_ocean_gpu_vs_oracle(ra, orc, n):
    rng = np.random.default_rng(n)
    T = 81
    b = np.concatenate([[1750.0], 1750.0 + np.cumsum(np.where(np.arange(T) % 6 == 1, 0.5, 1.0))])  # uneven steps
    P, inputs = _case(orc, "3D-GFDL", n, T, rng)
    P[orc.OCEAN_PARAM_NAMES.index("enable_temp_feedback")] = (np.arange(n) % 3 != 0).astype(float)
    scen = (np.arange(n) % 2).astype(np.int32)
    p0 = rng.uniform(275.0, 290.0, n)
    want = orc.ocean_run(b, P, inputs, p0, 5.0, scen=scen, threads=8)
    got = _gpu(ra, b, P, inputs, p0, 5.0, scen=scen)
    assert (np.isnan(got) == np.isnan(want)).all() and np.isnan(got[2, 0]).all() and np.array_equal(got[0, 0], p0)
    ok = ~np.isnan(want)
    err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
    assert err.max() <= TOL, f"max deviation {err.max():.3e}"
    off = np.arange(n) % 3 == 0
    assert np.array_equal(got[:, :, off], want[:, :, off], equal_nan=True)  # members without the exp: exact


@pytest.mark.parametrize("max_hist", [0, 5, 12, 30, 100])
def test_ocean_gpu_bounded_history(ra, orc, max_hist):
    """max_history_months shorter than the run: the window slides inside a year (5), across
    exactly one year (12) and across several (30, 100); 0 switches the convolution off."""
    rng = np.random.default_rng(max_hist)
    n, T = 33, 41
    b = np.arange(T + 1, dtype=float) + 1900.0
    P, inputs = _case(orc, "HILDA", n, T, rng, enable_temp_feedback=0.0, max_history_months=max_hist)
    want = orc.ocean_run(b, P, inputs[:1], 280.0, 0.0, threads=8)
    got = _gpu(ra, b, P, inputs[:1], 280.0, 0.0)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(_gpu(ra, b, P, inputs[:1], 280.0, 0.0, chunks=(3, 20)), got, equal_nan=True)
    assert np.array_equal(_gpu(ra, b, P, inputs[:1], 280.0, 0.0, chunks=range(1, T)), got, equal_nan=True)  # split tiles


def test_ocean_through_the_reference_shaped_front(ra, orc):
    from rscm_amd import core
    from rscm_amd.magicc import OceanCarbonBuilder
    years = np.arange(1950.0, 1991.0)
    axis = core.TimeAxis.from_bounds(np.append(years, 1991.0))
    T = len(years)
    ts = lambda v: core.Timeseries(v, axis, "", core.InterpolationStrategy.Previous)  # noqa: E731
    x = {"Atmospheric Concentration|CO2": 310.0 + 1.2 * np.arange(T), "Sea Surface Temperature": 0.01 * np.arange(T)}
    b = core.ModelBuilder().with_time_axis(axis).with_rust_component(
        OceanCarbonBuilder.from_parameters({"model": "2D-BERN", "gas_exchange_tau": 8.0}).build())
    for k, v in x.items():
        b = b.with_exogenous_variable(k, ts(v))
    m = b.with_initial_values({"Ocean Surface pCO2": 300.0, "Cumulative Ocean Uptake": 0.0}).build()
    m.run()
    res = m.timeseries()
    m.close()
    want = orc.ocean_run(np.append(years, 1991.0), orc.ocean_default_params("2D-BERN", gas_exchange_tau=8.0),
                         np.stack(list(x.values())), 300.0, 0.0)
    for k, name in enumerate(("Ocean Surface pCO2", "Cumulative Ocean Uptake", "Carbon Flux|Ocean")):
        got, w = res.get_timeseries_by_name(name).values(), want[k, :, 0]
        assert (np.isnan(got) == np.isnan(w)).all() and np.nanmax(np.abs(got - w) / np.maximum(1.0, np.abs(w))) <= TOL, name
    with pytest.raises(ValueError, match="unknown variant"):
        OceanCarbonBuilder.from_parameters({"model": "4D"})
    with pytest.raises(NotImplementedError, match="irf_early"):
        OceanCarbonBuilder.from_parameters({"irf_early": {"type": "Polynomial", "coefficients": [1.0]}})


def test_ocean_error_conventions(ra, orc):
    b = np.arange(11, dtype=float)
    with ra.Ensemble(ra.KIND_OCEAN_CARBON, 4, b) as e:
        P = np.repeat(orc.ocean_default_params().reshape(-1, 1), 4, axis=1)
        Q = P.copy()
        Q[orc.OCEAN_PARAM_NAMES.index("max_history_months"), 2] = 100.0
        with pytest.raises(ra.RscmGpuError, match="same for every member"):
            e.set_params(Q)
        Q = P.copy()
        Q[orc.OCEAN_PARAM_NAMES.index("steps_per_year")] = 4.0
        with pytest.raises(ra.RscmGpuError, match="steps_per_year = 12"):
            e.set_params(Q)
        e.set_params(P)
        e.set_forcing(np.stack([np.full(10, 400.0), np.zeros(10)]))
        with pytest.raises(ra.RscmGpuError, match="no initial value"):
            e.run()
        # the flux-history ring is internal state: a parameter update that changes its length mid-run is refused
        # (it would discard the pulses of the steps taken so far); after a rewind it is accepted
        e.set_initial(1, 400.0)
        e.set_initial(2, 0.0)
        Q = P.copy()
        Q[orc.OCEAN_PARAM_NAMES.index("max_history_months")] = 36.0
        e.run(5)
        e.set_params(P)   # same ring length: fine at any time
        with pytest.raises(ra.RscmGpuError, match="rewind"):
            e.set_params(Q)
        e.rewind()
        e.set_params(Q)
        e.run()
        short = e.get_series(1)
    with ra.Ensemble(ra.KIND_OCEAN_CARBON, 4, b) as e:   # a fresh ensemble with the short window gives the same bits
        e.set_params(Q)
        e.set_forcing(np.stack([np.full(10, 400.0), np.zeros(10)]))
        e.set_initial(1, 400.0)
        e.set_initial(2, 0.0)
        e.run()
        assert np.array_equal(e.get_series(1), short, equal_nan=True)


def test_ocean_full_window_properties(ra, orc):
    """4096 members x 751 years with the full 6000-month window (the history outgrows it after
    500 years): cumulative uptake equals the time-integrated flux (the reference's
    tests/conservation.rs property), ocean pCO2 approaches the atmosphere from below, duplicate
    members agree bit for bit, and two members match the oracle's full O(T^2) convolution."""
    n, T = 4096, 751
    rng = np.random.default_rng(1)
    b = np.arange(T + 1, dtype=float) + 1750.0
    inputs = np.stack([np.minimum(278.0 + 0.6 * np.arange(T), 560.0), np.zeros(T)])
    P = np.repeat(orc.ocean_default_params(enable_temp_feedback=0.0).reshape(-1, 1), n, axis=1)
    P[4] = rng.uniform(6.0, 10.0, n)
    P[4, n // 2:] = P[4, : n // 2]
    with ra.Ensemble(ra.KIND_OCEAN_CARBON, n, b) as e:
        e.set_params(P)
        e.set_forcing(inputs)
        e.set_initial(1, 278.0)
        e.set_initial(2, 0.0)
        e.run()
        pco2 = e.get_series(1)
        cum = e.get_series(2, T - 1, T)[0]
        flux = e.get_series(3, 1, T)
        ms = e.last_run_ms()
    assert np.abs(cum - flux.sum(axis=0)).max() < 1e-8 * cum.max()
    assert (pco2[-1] < 560.0).all() and (pco2[-1] > 460.0).all()
    assert (np.diff(pco2[1:480], axis=0) > 0.0).all()  # rising until the window starts dropping the oldest pulses
    assert np.array_equal(pco2[:, : n // 2], pco2[:, n // 2:])
    want = orc.ocean_run(b, P[:, :2].copy(), inputs, 278.0, 0.0, threads=2)
    assert np.array_equal(pco2[:, :2], want[0]) and np.array_equal(cum[:2], want[1, T - 1])
    print(f"ocean 4096 x 750 yr: {ms:.1f} ms")


def test_internal_state_kinds_refuse_time_jumps(ra, orc):
    """ClimateUDEB and OceanCarbon carry the reference's ComponentState on the device: their time
    index can be rewound or left alone, not moved somewhere the state has not been."""
    b = np.arange(11, dtype=float)
    with ra.Ensemble(ra.KIND_OCEAN_CARBON, 2, b) as e:
        e.set_params(np.repeat(orc.ocean_default_params().reshape(-1, 1), 2, axis=1))
        e.set_forcing(np.stack([np.full(10, 400.0), np.zeros(10)]))
        e.set_initial(1, 278.0)
        e.set_initial(2, 0.0)
        e.run(4)
        from rscm_amd import _lib
        with pytest.raises(ra.RscmGpuError, match="internal component state"):
            _lib.check(e._lib.rscm_ens_set_time_index(e._h, 2))
        _lib.check(e._lib.rscm_ens_set_time_index(e._h, 4))
        e.run()
        first = e.get_series(1)
        e.rewind()
        e.run()
        assert np.array_equal(e.get_series(1), first)


# measured over the three presets, temperature feedback on and off, 600 years: 5e-12 / 1.6e-11 / 5e-13 (printed below)
FAST_TOL = 2e-10


def _fast_info(e):
    import ctypes as C
    from rscm_amd import _lib
    uses, err = C.c_int32(), C.c_double()
    _lib.check(e._lib.rscm_ens_ocean_fast_info(e._h, C.byref(uses), C.byref(err)))
    return bool(uses.value), err.value


@pytest.mark.parametrize("model", ["3D-GFDL", "2D-BERN", "HILDA"])
def test_ocean_fast_mode_tolerance(ra, orc, model):
    """RSCM_MODE_FAST computes the history convolution in O(T): the last 60 (2D-BERN: 120) monthly lags
    explicitly, the older ones through 21 decaying modes fitted to the scaled impulse response (host fit,
    deviation from the table <= 5e-10, here ~1e-12) with one running sum each.  Against the EXACT mode
    (which equals the CPU oracle bit for bit) over 600 years -- the 500-year window fills and pulses leave
    it -- the outputs agree to FAST_TOL = 2e-10 relative (measured <= 2e-11: printed).  Launch boundaries and
    one-step launches do not change a bit of the FAST result; joining a run that EXACT began re-forms the
    running sums from the flux history."""
    rng = np.random.default_rng(12)
    n, T = 64, 601
    b = np.arange(T + 1, dtype=float) + 1750.0
    P, _ = _case(orc, model, n, T, rng, enable_temp_feedback=0.0)
    P[orc.OCEAN_PARAM_NAMES.index("enable_temp_feedback"), ::2] = 1.0
    # a bounded pathway: beyond ~1500 ppm the fifth-order Joos polynomial leaves its fitted range
    # and amplifies any rounding difference
    yr = np.arange(T)
    inputs = np.stack([np.minimum(278.0 + 0.9 * yr, 700.0) - np.where(yr > 500, 0.5 * (yr - 500), 0.0), 0.004 * np.minimum(yr, 400)])[None]

    def run(mode, chunks=(), switch_at=None):
        with ra.Ensemble(ra.KIND_OCEAN_CARBON, n, b) as e:
            e.set_params(P)
            e.set_forcing(inputs[:1])
            e.set_initial(1, 278.0)
            e.set_initial(2, 0.0)
            e.set_mode(ra.MODE_EXACT if switch_at else mode)
            uses, fit = _fast_info(e)
            assert uses and 0.0 <= fit <= 5e-10
            for c in chunks:
                e.run(c)
            if switch_at:
                e.run(switch_at)
                e.set_mode(mode)
            e.run()
            return np.stack([e.get_series(v) for v in (1, 2, 3)]), fit

    exact, fit = run(ra.MODE_EXACT)
    fast, _ = run(ra.MODE_FAST)
    want = orc.ocean_run(b, P[:, 1:3].copy(), inputs[:1], 278.0, 0.0, threads=2)   # member 1: no temperature feedback
    assert np.array_equal(exact[:, :, 1:2], want[:, :, :1], equal_nan=True)
    ok = ~np.isnan(exact)
    err = np.abs(fast[ok] - exact[ok]) / np.maximum(1.0, np.abs(exact[ok]))
    print(f"{model}: fit deviation {fit:.2e}, FAST vs EXACT over {T - 1} years: max relative deviation {err.max():.2e}")
    assert 0.0 < err.max() <= FAST_TOL, err.max()
    assert (np.isnan(fast) == np.isnan(exact)).all()
    # launch boundaries / one step per launch: the running sums and the last pulses are carried exactly
    assert np.array_equal(run(ra.MODE_FAST, chunks=(1, 2, 17, 300, 301, 555))[0], fast, equal_nan=True)
    assert np.array_equal(run(ra.MODE_FAST, chunks=range(1, 90))[0], fast, equal_nan=True)
    # EXACT for 520 years (pulses have started leaving the window), then FAST: the sums are re-formed from the history
    joined, _ = run(ra.MODE_FAST, switch_at=520)
    assert np.array_equal(joined[:, :521], exact[:, :521], equal_nan=True)
    errj = np.abs(joined[ok] - exact[ok]) / np.maximum(1.0, np.abs(exact[ok]))
    assert errj.max() <= FAST_TOL


def test_ocean_fast_mode_falls_back_when_the_window_is_short(ra, orc):
    """max_history_months below four times the explicit part: nothing to gain, FAST keeps the tiled
    convolution with fused multiply-adds (<= 1e-12 from EXACT)."""
    rng = np.random.default_rng(5)
    n, T = 64, 81
    b = np.arange(T + 1, dtype=float) + 1750.0
    P, inputs = _case(orc, "3D-GFDL", n, T, rng, enable_temp_feedback=0.0, max_history_months=100.0)
    out = {}
    for mode in (ra.MODE_EXACT, ra.MODE_FAST):
        with ra.Ensemble(ra.KIND_OCEAN_CARBON, n, b) as e:
            e.set_params(P)
            e.set_forcing(inputs[:1])
            e.set_initial(1, 278.0)
            e.set_initial(2, 0.0)
            e.set_mode(mode)
            assert not _fast_info(e)[0]
            e.run()
            out[mode] = np.stack([e.get_series(v) for v in (1, 2, 3)])
    ok = ~np.isnan(out[ra.MODE_EXACT])
    err = np.abs(out[ra.MODE_FAST][ok] - out[ra.MODE_EXACT][ok]) / np.maximum(1.0, np.abs(out[ra.MODE_EXACT][ok]))
    assert 0.0 < err.max() <= 1e-12
