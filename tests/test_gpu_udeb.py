"""GPU tier: the ClimateUDEB kernel (RSCM_KIND_UDEB) against the CPU oracle and, through it,
against the MAGICC7 outputs the reference's regression tests hold.

Parity bar (stated): |gpu - oracle| <= 1e-9 * max(1, |oracle|) for every output and year.  The
column solver on the device uses the refined-reciprocal quotient without the exact-division
replay (rk4_device.hpp), so agreement is to rounding, not bit for bit; status codes, NaN
placement and the time indexing are exact."""
import json
import os

import numpy as np
import pytest

from tests.test_oracle_udeb import PHASED, RECORDED, W, phased_ok, scenario_inputs

pytestmark = pytest.mark.gpu
RTOL = 1e-9
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "udeb_magicc7.json")
NAMES = {"st0": "Surface Temperature|NorthernOcean", "st1": "Surface Temperature|NorthernLand",
         "st2": "Surface Temperature|SouthernOcean", "st3": "Surface Temperature|SouthernLand",
         "heat_uptake": "Heat Uptake", "ohc": "Ocean Heat Content", "sst": "Sea Surface Temperature"}


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


def _gpu(ra, bounds, P, erf, scen=None, chunks=(), mode=None):
    with ra.Ensemble(ra.KIND_UDEB, P.shape[1], bounds) as e:
        if mode is not None:
            e.set_mode(mode)
        e.set_params(P)
        e.set_forcing(erf, scen)
        for k in range(1, 5):
            e.set_initial(k, 0.0)
        for c in chunks:
            e.run(c)
        e.run()
        return {k: e.get_series(v) for k, v in NAMES.items()}, e.status()


def _assert_close(got, want, what=""):
    for k in NAMES:
        g, w = got[k], want[k]
        assert (np.isnan(g) == np.isnan(w)).all(), f"{what} {k}: NaN placement"
        ok = ~np.isnan(w)
        err = np.abs(g[ok] - w[ok]) / np.maximum(1.0, np.abs(w[ok]))
        assert err.max() <= RTOL, f"{what} {k}: max rel err {err.max():.3e}"


@pytest.mark.parametrize("name", sorted(PHASED) + sorted(RECORDED) + ["11_efficacy_ar6", "12_efficacy_ar6_1pctco2"])
def test_udeb_gpu_magicc7_scenarios(ra, orc, name):
    g = json.load(open(GOLDEN))[name]
    kw, years, erf = scenario_inputs(name, g)
    b = np.append(years, years[-1] + 1.0)
    p = orc.udeb_default_params(**kw).reshape(-1, 1).copy()
    want, wst = orc.udeb_run(b, p, erf)
    got, st = _gpu(ra, b, p, erf)
    assert st[0] == wst[0] == 0
    _assert_close(got, want, name)
    actual = np.stack([got[f"st{k}"][:, 0] for k in range(4)], axis=1) @ W
    expected = np.array(g["surface_temperature"])
    if name in PHASED:  # the reference's own acceptance test against MAGICC7, through the GPU
        ok, worst = phased_ok(actual, expected, **PHASED[name])
        assert ok, worst
    else:
        m = np.abs(expected) > 1e-6
        assert np.all(np.abs(actual[m] - expected[m]) <= 0.1 * np.abs(expected[m]))


def _ensemble_params(orc, n, seed=0, **fixed):
    rng = np.random.default_rng(seed)
    P = np.repeat(orc.udeb_default_params(**fixed).reshape(-1, 1), n, axis=1)
    idx = orc.UDEB_PARAM_NAMES.index
    for name, (lo, hi) in dict(ecs=(1.8, 5.5), kappa=(0.4, 1.6), rlo=(1.15, 1.5), k_lo=(1.0, 2.0),
                               k_ns=(0.1, 0.6), w_initial=(2.0, 5.0), w_variable_fraction=(0.3, 0.9),
                               kappa_dkdt=(-0.3, 0.0), amplify_ocean_to_land=(1.0, 1.1),
                               temp_adjust_alpha=(1.0, 1.1), feedback_cumt_sensitivity=(0.0, 0.15),
                               polar_sinking_ratio=(0.1, 0.3), k_lg=(0.05, 0.2)).items():
        P[idx(name)] = rng.uniform(lo, hi, n)
    return P


@pytest.mark.parametrize("n_members", [96, 40000])   # the two-wavefront kernel (<= 32 768 members) and the one-thread kernel
@pytest.mark.parametrize("n_layers", [20, 30, 40])
def test_udeb_gpu_other_layer_counts(ra, orc, n_layers, n_members):
    """n_layers is a parameter of the reference (parameters/climate_udeb.rs:41, validated >= 2 at climate/udeb/mod.rs:164);
    the device unrolls the column solve per layer count and is instantiated for 20, 30, 40 and 50 layers.  Same bar against
    the oracle as at 50 layers (the oracle takes any count); every other count: test_udeb_gpu_any_layer_count."""
    years = np.arange(1850.0, 1931.0)
    b = np.append(years, 1931.0)
    P = _ensemble_params(orc, n_members, seed=n_layers, n_layers=float(n_layers))
    F = np.stack([np.where(years >= 1851, 3.71, 0.0), 3.71 * np.log(np.where(years > 1850, 1.01 ** (years - 1850), 1.0)) / np.log(2.0)])
    scen = (np.arange(n_members) % 2).astype(np.int32)
    pick = np.arange(n_members) if n_members <= 512 else np.random.default_rng(1).choice(n_members, 256, replace=False)
    want, wst = orc.udeb_run(b, P[:, pick].copy(), F, scen=scen[pick].copy(), threads=8)
    got, st = _gpu(ra, b, P, F, scen=scen, chunks=(1, 29))
    assert not st.any() and not wst.any()
    _assert_close({k: v[:, pick] for k, v in got.items()}, want, f"{n_layers} layers")


def _variant(ra, v):
    from rscm_amd import _lib as L
    L.check(L.load().rscm_gpu_set_udeb_variant(v))


@pytest.mark.parametrize("n_layers", [2, 3, 7, 19, 21, 25, 49, 51, 64, 65, 80, 100, 128, 129])
def test_udeb_gpu_any_layer_count(ra, orc, n_layers):
    """Every n_layers >= 2 the reference accepts (parameters/climate_udeb.rs:41; from_parameters refuses < 2, mod.rs:162-165) runs on
    the device.  Up to 64 layers a member's columns stay in registers + LDS: counts other than 20 / 30 / 40 / 50 take the next
    capacity's instance of the unrolled kernels with the count at run time (csrc/udeb_body.hpp, DYN: no branch -- the rows past the
    count are ZERO ROWS of the geometry table, exact no-ops; the statements of a live row unchanged); 65 to 128 layers a hemisphere per wavefront with the column in registers and the sweep's c' array
    in LDS; beyond 128 the columns-in-HBM kernel (csrc/udeb_any_body.hpp: plain loops over the layers, the same row arithmetic).  Same 1e-9 bar against the oracle (which takes any count; the reference's MAGICC7 files pin 50
    layers only: at every other count, and for the LDS kernel, parity is against the in-repo restatement alone -- PARITY UNPINNED); both arithmetic modes; launch boundaries (resume from the
    stored columns and scalars) change nothing; a member the reference refuses to build is flagged and NaN; more than 50 layers
    means the initial profile's last value below layer 50, as in the oracle.  Up to 64 layers the three kernels -- a hemisphere per
    wavefront, one thread per member, columns in HBM -- carry the same bits; from 65 to 128 the LDS kernel and the HBM kernel do."""
    years = np.arange(1850.0, 1931.0)
    b = np.append(years, 1931.0)
    n = 300   # a ragged last workgroup of the 256-thread kernel
    P = _ensemble_params(orc, n, seed=100 + n_layers, n_layers=float(n_layers))
    P[orc.UDEB_PARAM_NAMES.index("prescribed_efficacy_co2"), 7] = -1.0
    F = np.stack([np.where(years >= 1851, 3.71, 0.0), 3.71 * np.log(np.where(years > 1850, 1.01 ** (years - 1850), 1.0)) / np.log(2.0)])
    scen = (np.arange(n) % 2).astype(np.int32)
    want, wst = orc.udeb_run(b, P, F, scen=scen, threads=8)
    got, st = _gpu(ra, b, P, F, scen=scen)
    assert (st == wst).all() and st[7] == 2 and (st != 0).sum() == 1
    _assert_close(got, want, f"{n_layers} layers")
    assert all(np.isnan(got[k][1:, 7]).all() for k in NAMES)
    again, _ = _gpu(ra, b, P, F, scen=scen, chunks=(1, 29))
    for k in NAMES:
        assert np.array_equal(again[k], got[k], equal_nan=True), k
    fast, _ = _gpu(ra, b, P, F, scen=scen, mode=ra.MODE_FAST)
    _assert_close(fast, want, f"{n_layers} layers, FAST")
    if n_layers <= 128:
        try:
            for mode, ref in ((None, got), (ra.MODE_FAST, fast)):
                for v in ((0, 2, 3) if n_layers <= 64 else (3,)):   # one thread per member; a hemisphere per wavefront; columns in HBM
                    _variant(ra, v)
                    other, st_v = _gpu(ra, b, P, F, scen=scen, chunks=(17,), mode=mode)
                    assert (st_v == st).all()
                    for k in NAMES:
                        assert np.array_equal(other[k], ref[k], equal_nan=True), (v, mode, k)
        finally:
            _variant(ra, -1)


@pytest.mark.parametrize("n_layers", [21, 49, 51, 64])
def test_udeb_gpu_runtime_layer_count_one_thread_kernel(ra, orc, n_layers):
    """The one-thread-per-member kernel (the default beyond 32 768 members) with the layer count at run time, at a size that
    takes it by default: against the oracle on a sample of members, and bit for bit against the two-wavefront kernel."""
    years = np.arange(1850.0, 1911.0)
    b = np.append(years, 1911.0)
    n = 33_000
    P = _ensemble_params(orc, n, seed=200 + n_layers, n_layers=float(n_layers))
    F = np.stack([np.where(years >= 1851, 3.71, 0.0), 3.71 * np.log(np.where(years > 1850, 1.01 ** (years - 1850), 1.0)) / np.log(2.0)])
    scen = (np.arange(n) % 2).astype(np.int32)
    pick = np.random.default_rng(2).choice(n, 192, replace=False)
    want, wst = orc.udeb_run(b, P[:, pick].copy(), F, scen=scen[pick].copy(), threads=8)
    got, st = _gpu(ra, b, P, F, scen=scen, chunks=(1, 29))
    assert not st.any() and not wst.any()
    _assert_close({k: v[:, pick] for k, v in got.items()}, want, f"{n_layers} layers")
    try:
        _variant(ra, 2)
        two, _ = _gpu(ra, b, P, F, scen=scen)
    finally:
        _variant(ra, -1)
    for k in NAMES:
        assert np.array_equal(two[k], got[k], equal_nan=True), k


def test_udeb_gpu_layer_count_errors(ra, orc):
    """n_layers < 2 is refused with the reference's message (mod.rs:162-165), a fractional count too; the count is structural."""
    years = np.arange(1850.0, 1861.0)
    b = np.append(years, 1861.0)
    F = np.zeros((1, len(years)))
    for bad in (1.0, 0.0, 2.5):
        with pytest.raises(ra.RscmGpuError, match="n_layers"):
            _gpu(ra, b, _ensemble_params(orc, 8, n_layers=bad), F)
    Q = _ensemble_params(orc, 8, n_layers=25.0)
    Q[orc.UDEB_PARAM_NAMES.index("n_layers"), 3] = 26.0
    with pytest.raises(ra.RscmGpuError, match="same for every member"):
        _gpu(ra, b, Q, F)


@pytest.mark.parametrize("n_members", [257, 40000])   # the two-wavefront kernel and the one-thread kernel
def test_udeb_gpu_fast_mode(ra, orc, n_members):
    """RSCM_MODE_FAST: one refinement term of the row reciprocals instead of two (relative error 2^-46 instead of 2^-69
    per row of the column solve).  The same 1e-9 bar against the oracle as the default mode (measured: 1e-13, printed);
    the two kernels carry the same bits in this mode too; launch boundaries change nothing."""
    years = np.arange(1850.0, 1981.0)
    b = np.append(years, 1981.0)
    P = _ensemble_params(orc, n_members, seed=9)
    F = np.stack([np.where(years >= 1851, 3.71, 0.0), 3.71 * np.log(np.where(years > 1850, 1.01 ** (years - 1850), 1.0)) / np.log(2.0)])
    scen = (np.arange(n_members) % 2).astype(np.int32)
    pick = np.arange(n_members) if n_members <= 512 else np.arange(256)
    want, wst = orc.udeb_run(b, P[:, pick].copy(), F, scen=scen[pick].copy(), threads=8)
    got, st = _gpu(ra, b, P, F, scen=scen, mode=ra.MODE_FAST)
    exact, _ = _gpu(ra, b, P, F, scen=scen)
    assert not st.any() and not wst.any()
    _assert_close({k: v[:, pick] for k, v in got.items()}, want, "FAST")
    worst = max(float(np.nanmax(np.abs(got[k] - exact[k]) / np.maximum(1.0, np.abs(exact[k])))) for k in NAMES)
    print(f"ClimateUDEB FAST vs default mode, {n_members} members x {len(years) - 1} years: max relative deviation {worst:.2e}")
    assert 0.0 < worst < 1e-11
    again, _ = _gpu(ra, b, P, F, scen=scen, chunks=(1, 50), mode=ra.MODE_FAST)
    for k in NAMES:
        assert np.array_equal(again[k], got[k], equal_nan=True), k
    if n_members > 32768:   # the first members through the other kernel: the same bits
        small, _ = _gpu(ra, b, P[:, :256].copy(), F, scen=scen[:256].copy(), mode=ra.MODE_FAST)
        for k in NAMES:
            assert np.array_equal(small[k], got[k][:, :256], equal_nan=True), k


def test_udeb_gpu_ensemble_vs_oracle(ra, orc):
    years = np.arange(1850.0, 1951.0)
    b = np.append(years, 1951.0)
    n = 257
    P = _ensemble_params(orc, n)
    F = np.stack([np.where(years >= 1851, 3.71, 0.0),
                  3.71 * np.log(np.where(years > 1850, 1.01 ** (years - 1850), 1.0)) / np.log(2.0),
                  -1.5 * np.ones(len(years))])
    scen = (np.arange(n) % 3).astype(np.int32)
    want, wst = orc.udeb_run(b, P, F, scen=scen, threads=8)
    got, st = _gpu(ra, b, P, F, scen=scen)
    assert (st == wst).all()
    _assert_close(got, want, "ensemble")
    # resume: three launches give the same bits as one
    again, _ = _gpu(ra, b, P, F, scen=scen, chunks=(1, 37))
    for k in NAMES:
        assert np.array_equal(again[k], got[k], equal_nan=True), k


@pytest.mark.parametrize("n_members", [96, 40000])   # the two-wavefront kernel (a hemisphere per wavefront) and the one-thread kernel
def test_udeb_gpu_feedback_window_shorter_than_a_model_step(ra, orc, n_members):
    """feedback_cumt_period = 0.5 yr on an annual axis with feedback_cumt_sensitivity != 0: the look-back window of adjusted_ecs()
    ends inside the PREVIOUS model step, so the only history entry it touches is the one stored at the end of that step -- in the
    two-wavefront kernel by the other wavefront, after the year's last barrier.  Both wavefronts take it from their own register;
    the result is the oracle's (climate/udeb/mod.rs:399-470)."""
    years = np.arange(1850.0, 1931.0)
    b = np.append(years, 1931.0)
    P = _ensemble_params(orc, n_members, seed=12)
    P[orc.UDEB_PARAM_NAMES.index("feedback_cumt_period")] = 0.5
    P[orc.UDEB_PARAM_NAMES.index("feedback_cumt_sensitivity")] = 0.02
    F = np.stack([np.where(years >= 1851, 3.71, 0.0), 3.71 * np.log(np.where(years > 1850, 1.01 ** (years - 1850), 1.0)) / np.log(2.0)])
    scen = (np.arange(n_members) % 2).astype(np.int32)
    pick = np.arange(n_members) if n_members <= 512 else np.arange(256)
    want, wst = orc.udeb_run(b, P[:, pick].copy(), F, scen=scen[pick].copy(), threads=8)
    got, st = _gpu(ra, b, P, F, scen=scen, chunks=(1, 29))
    assert not st.any() and not wst.any()
    _assert_close({k: v[:, pick] for k, v in got.items()}, want, "half-year feedback window")
    base = P.copy()
    base[orc.UDEB_PARAM_NAMES.index("feedback_cumt_sensitivity")] = 0.0
    off, _ = _gpu(ra, b, base, F, scen=scen)
    assert np.nanmax(np.abs(off["sst"] - got["sst"])) > 1e-6   # the window does act in this configuration


def test_udeb_gpu_failed_construction_is_flagged(ra, orc):
    years = np.arange(1850.0, 1871.0)
    b = np.append(years, 1871.0)
    P = _ensemble_params(orc, 8, seed=3)
    idx = orc.UDEB_PARAM_NAMES.index
    P[idx("prescribed_efficacy_co2"), 2] = -1.0      # rejected by from_parameters (status 2)
    P[idx("rlo"), 5] = 50.0                           # LAMCALC cannot match this ratio (status 4)
    erf = np.where(years >= 1851, 3.71, 0.0)
    want, wst = orc.udeb_run(b, P, erf)
    got, st = _gpu(ra, b, P, erf)
    assert wst[2] == 2 and wst[5] == 4
    assert (st == wst).all()
    _assert_close(got, want, "failed members")
    assert np.isnan(got["st0"][1:, 2]).all() and np.isnan(got["st0"][1:, 5]).all()


def test_udeb_structural_parameters_must_be_uniform(ra, orc):
    from rscm_amd import RscmGpuError
    years = np.arange(1850.0, 1861.0)
    b = np.append(years, 1861.0)
    P = _ensemble_params(orc, 4)
    with ra.Ensemble(ra.KIND_UDEB, 4, b) as e:
        bad = P.copy()
        bad[orc.UDEB_PARAM_NAMES.index("land_heat_capacity_enabled"), 1] = 0.0
        with pytest.raises(RscmGpuError, match="same for every member"):
            e.set_params(bad)
        bad = P.copy()
        bad[0] = 1.0
        with pytest.raises(RscmGpuError, match="n_layers: must be >= 2"):   # the reference's message (mod.rs:162-165)
            e.set_params(bad)
        e.set_params(P)
        e.set_forcing(np.zeros(len(years)))
        with pytest.raises(RscmGpuError, match="MissingInitialValue"):
            e.run()


def test_udeb_through_the_reference_shaped_front(ra, orc):
    """The model construction of tests/regression/test_ocean_udeb.py:57-130 (ClimateUDEBBuilder,
    schema with a FourBox 'Surface Temperature', exogenous Linear ERF, scalar initial value)
    through rscm_amd's mirror, on scenario 10 (full defaults, 1pctCO2)."""
    from rscm_amd import core
    from rscm_amd.magicc import ClimateUDEBBuilder
    g = json.load(open(GOLDEN))["10_full_default"]
    kw, years, erf = scenario_inputs("10_full_default", g)
    climate = ClimateUDEBBuilder.from_parameters({"ecs": kw["ecs"], "rf_2xco2": kw["rf_2xco2"]}).build()
    axis = core.TimeAxis.from_bounds(np.concatenate([years, [years[-1] + 1.0]]))
    erf_ts = core.Timeseries(erf, axis, "W/m^2", core.InterpolationStrategy.Linear)
    schema = core.VariableSchema()
    schema.add_variable("Effective Radiative Forcing", "W/m^2")
    schema.add_variable("Surface Temperature", "K", core.GridType.FourBox)
    for n, u in (("Heat Uptake", "W/m^2"), ("Ocean Heat Content", "J/m^2"), ("Sea Surface Temperature", "K")):
        schema.add_variable(n, u)
    model = (core.ModelBuilder().with_time_axis(axis).with_schema(schema).with_rust_component(climate)
             .with_exogenous_variable("Effective Radiative Forcing", erf_ts)
             .with_initial_values({"Surface Temperature": 0.0}).build())
    model.run()
    res = model.timeseries()
    t4 = res.get_fourbox_timeseries_by_name("Surface Temperature")
    assert t4 is not None and t4.values().shape == (len(years), 4)
    actual = t4.values() @ W
    expected = np.array(g["surface_temperature"])
    m = np.abs(expected) > 1e-6
    assert np.all(np.abs(actual[m] - expected[m]) <= 0.1 * np.abs(expected[m]))  # the reference's bar
    assert np.abs((actual[5:] - expected[5:]) / expected[5:]).max() < 0.013
    sst = res.get_timeseries_by_name("Sea Surface Temperature").values()
    assert np.isnan(sst[0]) and np.all(np.isfinite(sst[1:]))
    with pytest.raises(ValueError, match="unknown field"):
        ClimateUDEBBuilder.from_parameters({"nope": 1.0})
    model.close()


def test_udeb_model_runner_and_device_likelihood(ra, orc):
    """The calibration front over the UDEB kind: ModelRunner varies ecs and kappa per member,
    every other ClimateUDEB parameter stays at the builder's value; device log-likelihood equals
    the host formula over the extracted outputs."""
    from rscm_amd import calibrate as cal
    from rscm_amd import core
    from rscm_amd.magicc import ClimateUDEBBuilder
    years = np.arange(1850.0, 1921.0)
    axis = core.TimeAxis.from_values(years)
    erf = np.where(years >= 1851.0, 3.71, 0.0)
    b = (core.ModelBuilder().with_time_axis(axis)
         .with_rust_component(ClimateUDEBBuilder.from_parameters({}).build())
         .with_exogenous_variable("Effective Radiative Forcing",
                                  core.Timeseries(erf, axis, "W/m^2", core.InterpolationStrategy.Previous))
         .with_initial_values({"Surface Temperature": 0.0}))
    runner = cal.ModelRunner(b, ["ecs", "kappa"], ["Sea Surface Temperature", "Heat Uptake"])
    sets = np.column_stack([np.linspace(2.0, 5.0, 9), np.linspace(0.5, 1.5, 9)])
    outs = runner.run_batch(sets)
    assert len(outs) == 9
    P = np.repeat(orc.udeb_default_params().reshape(-1, 1), 9, axis=1)
    P[orc.UDEB_PARAM_NAMES.index("ecs")] = sets[:, 0]
    P[orc.UDEB_PARAM_NAMES.index("kappa")] = sets[:, 1]
    want, _ = orc.udeb_run(np.append(years, years[-1] + 1.0), P, erf)
    for i, o in enumerate(outs):
        assert 1850.0 not in o["Sea Surface Temperature"]  # index 0 is NaN -> skipped
        got = np.array([o["Sea Surface Temperature"][float(y)] for y in years[1:]])
        assert np.abs(got - want["sst"][1:, i]).max() <= RTOL * max(1.0, np.abs(want["sst"][1:, i]).max())
    sst = np.array([outs[k]["Sea Surface Temperature"][1920.0] for k in range(9)])
    assert np.all(np.diff(sst) > 0)  # higher ECS (and kappa) -> warmer after 70 years here
    target = cal.Target()
    for y in (1870.0, 1900.0, 1920.0):
        target.add_observation("Sea Surface Temperature", y, outs[4]["Sea Surface Temperature"][y], 0.05)
    lik = cal.GaussianLikelihood()
    dev = runner.log_likelihood_batch(sets, target, lik)
    host = np.array([lik.ln_likelihood(o, target) for o in outs])
    assert np.allclose(dev, host, rtol=1e-12, atol=1e-12) and dev[4] == 0.0 and np.argmax(dev) == 4
    runner.close()


@pytest.mark.parametrize("seed", range(10))
def test_udeb_fuzz(ra, orc, seed):
    """Seeded random ClimateUDEB configurations: structural switches (land heat capacity,
    efficacy mode, sub-steps per year, depth-dependent area, feedback window length), an irregular
    time axis, random launch chunking -- the running feedback window, the structured LAMCALC and the
    regrouped column algebra against the oracle's plain forms."""
    rng = np.random.default_rng(500 + seed)
    T = int(rng.integers(3, 90))
    n = int(rng.choice([1, 63, 130]))
    b = np.concatenate([[1850.0], 1850.0 + np.cumsum(rng.choice([0.5, 1.0, 1.0, 2.0], T))])
    fixed = dict(land_heat_capacity_enabled=float(rng.integers(0, 2)), efficacy_apply=float(rng.integers(0, 3)),
                 steps_per_year=float(rng.choice([1, 4, 12])), depth_dependent_area=float(rng.choice([0.0, 0.5, 1.0])),
                 feedback_cumt_period=float(rng.choice([3.0, 17.5, 300.0])), prescribed_efficacy_co2=float(rng.choice([1.0, 1.1])))
    P = _ensemble_params(orc, n, seed=seed, **fixed)
    if rng.random() < 0.3:
        P[orc.UDEB_PARAM_NAMES.index("feedback_cumt_sensitivity")] = 0.0
    if rng.random() < 0.3:
        P[orc.UDEB_PARAM_NAMES.index("kappa_dkdt")] = 0.0
    S = int(rng.choice([1, 3]))
    F = np.cumsum(rng.normal(0.05, 0.3, (S, T)), axis=1)
    scen = rng.integers(0, S, n).astype(np.int32) if S > 1 else None
    want, wst = orc.udeb_run(b, P, F, scen=scen, threads=8)
    cuts = tuple(sorted(set(int(x) for x in rng.integers(1, T, int(rng.integers(0, 3))))))
    got, st = _gpu(ra, b, P, F, scen=scen, chunks=cuts)
    assert (st == wst).all()
    _assert_close(got, want, f"fuzz seed {seed} ({fixed}, T={T}, n={n}, cuts={cuts})")


@pytest.mark.parametrize("n_layers", [50, 49])   # the count compiled in; the count at run time (capacity 50)
def test_udeb_runs_cut_into_member_blocks_and_chunks_keep_the_bits(ra, orc, n_layers):
    """A whole-axis ClimateUDEB run over more than 65 536 members (one wavefront per SIMD) is issued as two halves of the members on two
    streams in chunks of model steps (rscm_ens_last_run_plan: 2 x 8 for 750 steps; each chunk reloads and stores the ocean columns and
    the scalars like any resumed run).  The same axis in five pieces of 150 steps (fewer than three chunks) takes the single-launch path: same bits,
    also the internal state at the end; members on both sides of the cut against the oracle."""
    n = 70_001
    years = np.arange(1750.0, 2501.0)
    b = np.append(years, 2501.0)
    P = _ensemble_params(orc, n, seed=21, n_layers=float(n_layers))
    F = np.stack([3.71 * np.minimum((years - 1750.0) / 200.0, 1.0), 1.5 * np.sin((years - 1750.0) / 40.0)])
    scen = (np.arange(n) % 2).astype(np.int32)

    def run(pieces):
        with ra.Ensemble(ra.KIND_UDEB, n, b) as e:
            e.set_params(P)
            e.set_forcing(F, scen)
            for k in range(1, 5):
                e.set_initial(k, 0.0)
            plans = []
            for c in pieces:
                e.run(c)
                plans.append(e.last_run_plan())
            e.run()
            plans.append(e.last_run_plan())
            rows = {k: e.get_series(v, 0, 751, 125) for k, v in NAMES.items()}
            sample = {k: e.get_series(v, 0, 751, 1, 34_960, 35_060) for k, v in NAMES.items()}   # the cut is at member 35 008
            return rows, sample, plans, e.status().copy(), e.checkpoint()

    cut_rows, cut_sample, plans, st_cut, ck_cut = run(())
    assert plans == [(2, 8)], plans
    one_rows, one_sample, plans, st_one, ck_one = run((150, 300, 450, 600))
    assert plans == [(1, 1)] * 5, plans
    assert np.array_equal(st_cut, st_one) and not st_cut.any()
    for k in NAMES:
        assert np.array_equal(cut_rows[k], one_rows[k], equal_nan=True), k
        assert np.array_equal(cut_sample[k], one_sample[k], equal_nan=True), k
    for key in ck_cut:   # ocean columns, scalars, history: the internal state
        a, c = ck_cut[key], ck_one[key]
        if isinstance(a, np.ndarray):
            assert np.array_equal(a, c, equal_nan=True), key
    pick = np.arange(34_960, 35_060)
    want, wst = orc.udeb_run(b, P[:, pick].copy(), F, scen=scen[pick].copy(), threads=8)
    assert not wst.any()
    _assert_close(cut_sample, want, "members across the cut")


@pytest.mark.parametrize("seed", range(24))
def test_udeb_fuzz_over_layer_counts(ra, orc, seed):
    """The fuzz of test_udeb_fuzz with the LAYER COUNT drawn too -- 2 to 128, i.e. every instance of the column solve: the count
    compiled in, the count at run time in each capacity, the c' array in LDS -- together with layer thickness, mixed-layer depth and the
    depth-dependent area (which shape the geometry table the rows past the end of a column read as zeros), random launch chunking
    (resume through the stored columns) and ensemble sizes on both sides of a wavefront."""
    rng = np.random.default_rng(900 + seed)
    T = int(rng.integers(3, 60))
    n = int(rng.choice([1, 64, 65, 200]))
    nl = int(rng.choice([2, 5, 11, 20, 23, 30, 37, 40, 46, 50, 57, 64, 65, 90, 128]))
    b = np.concatenate([[1850.0], 1850.0 + np.cumsum(rng.choice([0.5, 1.0, 1.0, 2.0], T))])
    fixed = dict(n_layers=float(nl), layer_thickness=float(rng.choice([40.0, 100.0, 250.0])), mixed_layer_depth=float(rng.choice([50.0, 60.0, 90.0])),
                 land_heat_capacity_enabled=float(rng.integers(0, 2)), efficacy_apply=float(rng.integers(0, 3)),
                 steps_per_year=float(rng.choice([1, 4, 12])), depth_dependent_area=float(rng.choice([0.0, 0.5, 1.0])),
                 feedback_cumt_period=float(rng.choice([3.0, 17.5, 300.0])))
    P = _ensemble_params(orc, n, seed=seed, **fixed)
    S = int(rng.choice([1, 3]))
    F = np.cumsum(rng.normal(0.05, 0.3, (S, T)), axis=1)
    scen = rng.integers(0, S, n).astype(np.int32) if S > 1 else None
    want, wst = orc.udeb_run(b, P, F, scen=scen, threads=8)
    cuts = tuple(sorted(set(int(x) for x in rng.integers(1, T, int(rng.integers(0, 3))))))
    for mode in (None, ra.MODE_FAST):
        got, st = _gpu(ra, b, P, F, scen=scen, chunks=cuts, mode=mode)
        assert (st == wst).all()
        _assert_close(got, want, f"fuzz seed {seed} ({fixed}, T={T}, n={n}, cuts={cuts}, mode={mode})")


@pytest.mark.parametrize("n_layers,n", [(49, 200), (21, 40_000), (100, 130)])
def test_udeb_runtime_layer_count_one_step_per_launch(ra, orc, n_layers, n):
    """What a lock-step graph does to ClimateUDEB -- one model step per launch, the columns and scalars out to HBM and back every
    step -- with the layer count at run time (two-wavefront and one-thread kernels, c' in LDS): the same bits as the whole axis in one
    launch."""
    years = np.arange(1850.0, 1886.0)
    b = np.append(years, 1886.0)
    P = _ensemble_params(orc, n, seed=300 + n_layers, n_layers=float(n_layers))
    F = 3.71 * np.minimum((years - 1850.0) / 20.0, 1.0)
    whole, st = _gpu(ra, b, P, F)
    assert not st.any()
    with ra.Ensemble(ra.KIND_UDEB, n, b) as e:
        e.set_params(P)
        e.set_forcing(F)
        for k in range(1, 5):
            e.set_initial(k, 0.0)
        while not e.finished():
            e.step()
        for k, v in NAMES.items():
            assert np.array_equal(e.get_series(v), whole[k], equal_nan=True), k
