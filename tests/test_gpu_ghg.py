"""GhgForcing on the GPU (csrc/ghg.hip through the C ABI) against the CPU oracle
(oracle/ghg_oracle.c) and the MAGICC7 outputs the reference's regression tests hold
(tests/golden/ghg_forcing_magicc7.json).

Tolerance: |gpu - oracle| <= 1e-12 * max(1, |oracle|).  The kernel factorises ln(C/C0) and the
(M N)^p powers into scenario and member parts (csrc/ghg.hip), which rounds differently from the
reference's expressions; the forcings are O(1) W/m^2, so this is ~4 decimal digits above f64
rounding and 7 below the reference's own MAGICC7 tolerance (rtol 1e-5)."""
import json
import os

import numpy as np
import pytest

from tests.test_oracle_ghg import ATOL, GOLD, RTOL, ghg_params_from_config, scenario_concentrations

pytestmark = pytest.mark.gpu
TOL = 1e-12
NAMES = {"co2_erf": 1, "ch4_erf": 2, "n2o_erf": 3}


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


def _gpu(ra, T, P, conc, scen=None, chunks=()):
    b = np.arange(T + 1, dtype=float) + 1750.0
    with ra.Ensemble(ra.KIND_GHG_FORCING, P.shape[1], b) as e:
        e.set_params(P)
        e.set_forcing(conc, scen)
        for c in chunks:
            e.run(c)
        e.run()
        assert not e.status().any()
        return {k: e.get_series(v) for k, v in NAMES.items()}


def _assert_close(got, want, what=""):
    for k in NAMES:
        g, w = got[k], want[k]
        assert (np.isnan(g) == np.isnan(w)).all(), f"{what} {k}: NaN placement"
        ok = ~np.isnan(w)
        err = np.abs(g[ok] - w[ok]) / np.maximum(1.0, np.abs(w[ok]))
        assert err.max() <= TOL, f"{what} {k}: max deviation {err.max():.3e}"


@pytest.mark.parametrize("name", ["01_concentration_driven", "02_ghg_forcing_olbl"])
def test_ghg_gpu_magicc7_scenarios(ra, orc, name):
    g = GOLD[name]
    co2, ch4, n2o = scenario_concentrations(g)
    conc = np.stack([co2, ch4, n2o])
    P = ghg_params_from_config(g["config"], co2, ch4, n2o).reshape(-1, 1)
    got = _gpu(ra, len(co2), P, conc)
    _assert_close(got, orc.ghg_run(len(co2), P, conc), name)
    for var, key in (("co2_erf", "CO2"), ("ch4_erf", "CH4"), ("n2o_erf", "N2O")):
        assert np.isnan(got[var][0, 0])
        np.testing.assert_allclose(got[var][1:, 0], np.array(g["Effective Radiative Forcing|" + key])[:-1],
                                   rtol=RTOL, atol=ATOL, err_msg=f"{name} {key} vs MAGICC7")


def _ensemble(orc, n, method, seed=0):
    rng = np.random.default_rng(seed)
    P = np.repeat(orc.ghg_default_params(method=method).reshape(-1, 1), n, axis=1)
    idx = orc.GHG_PARAM_NAMES.index
    for name, (lo, hi) in dict(co2_pi=(270.0, 290.0), ch4_pi=(650.0, 800.0), n2o_pi=(260.0, 280.0),
                               delq2xco2=(3.4, 4.2), ch4_radeff=(0.03, 0.04), n2o_radeff=(0.1, 0.14),
                               adjust_co2=(0.9, 1.1), adjust_ch4=(0.7, 1.0), adjust_n2o=(0.85, 1.1),
                               olbl_co2_d1=(5.0, 5.4), olbl_ch4_d3=(0.04, 0.05), olbl_n2o_d2=(0.12, 0.16),
                               olbl_co2_b1=(6e-4, 9e-4)).items():
        P[idx(name)] = rng.uniform(lo, hi, n)
    return P


@pytest.mark.parametrize("method", ["Ipcctar", "Olbl"])
@pytest.mark.parametrize("n", [1, 63, 1000])
def test_ghg_gpu_ensemble_vs_oracle(ra, orc, method, n):
    T = 351
    yr = np.arange(T)
    # scenario 0 rises through the OLBL saturation concentration (~1809 ppm) and scenario 1 dips
    # below the pre-industrial values, so all three alpha regimes of forcing/ghg.rs:210-240 occur
    conc = np.stack([np.stack([278.0 * 1.006 ** yr, 722.0 + 6.0 * yr, 270.0 + 0.4 * yr]),
                     np.stack([300.0 - 0.2 * yr, 800.0 - 0.5 * yr, 275.0 - 0.05 * yr])])
    P = _ensemble(orc, n, method, seed=n)
    scen = (np.arange(n) % 2).astype(np.int32)
    want = orc.ghg_run(T, P, conc, scen=scen, threads=8)
    got = _gpu(ra, T, P, conc, scen=scen)
    _assert_close(got, want, f"{method} n={n}")
    if method == "Olbl":  # the regimes really are exercised
        c_max = P[1] - P[8] / (2.0 * P[7])
        assert (conc[0, 0].max() > c_max).any() and (conc[1, 0].min() < P[1]).any()
    # three launches give the same bits as one
    again = _gpu(ra, T, P, conc, scen=scen, chunks=(1, 100))
    for k in NAMES:
        assert np.array_equal(again[k], got[k], equal_nan=True), k
    # without a scenario map every member reads scenario 0
    got0 = _gpu(ra, T, P, conc[:1])
    _assert_close(got0, orc.ghg_run(T, P, conc[:1]), f"{method} n={n} one scenario")


def test_ghg_through_the_reference_shaped_front(ra, orc):
    """tests/regression/test_ghg_forcing.py build_ghg_forcing_model + test_02, spelled with the
    same builder calls against rscm_amd."""
    from rscm_amd import core
    from rscm_amd.magicc import GhgForcingBuilder
    g = GOLD["02_ghg_forcing_olbl"]
    years = np.array(g["years"], dtype=float)
    co2, ch4, n2o = scenario_concentrations(g)
    comp = GhgForcingBuilder.from_parameters({
        "method": "Olbl", "delq2xco2": 3.71, "co2_pi": float(co2[0]), "ch4_pi": float(ch4[0]),
        "n2o_pi": float(n2o[0]), "adjust_co2": 1.05, "adjust_ch4": 0.86, "adjust_n2o": 0.93}).build()
    axis = core.TimeAxis.from_bounds(np.concatenate([years, [years[-1] + 1.0]]))
    b = core.ModelBuilder().with_time_axis(axis).with_rust_component(comp)
    for name, vals, unit in (("Atmospheric Concentration|CO2", co2, "ppm"), ("Atmospheric Concentration|CH4", ch4, "ppb"),
                             ("Atmospheric Concentration|N2O", n2o, "ppb")):
        b = b.with_exogenous_variable(name, core.Timeseries(vals, axis, unit, core.InterpolationStrategy.Linear))
    model = b.build()
    model.run()
    res = model.timeseries()
    for key in ("CO2", "CH4", "N2O"):
        actual = res.get_timeseries_by_name("Effective Radiative Forcing|" + key).values()
        assert np.isnan(actual[0])
        np.testing.assert_allclose(actual[1:], np.array(g["Effective Radiative Forcing|" + key])[:-1], rtol=RTOL, atol=ATOL)
    assert np.array_equal(res.get_timeseries_by_name("Atmospheric Concentration|CH4").values(), ch4)
    with pytest.raises(ValueError, match="unknown variant"):
        GhgForcingBuilder.from_parameters({"method": "Etminan"})
    with pytest.raises(ValueError, match="unknown field"):
        GhgForcingBuilder.from_parameters({"co2_preindustrial": 278.0})
    model.close()


def test_ghg_error_conventions(ra, orc):
    T, n = 20, 8
    b = np.arange(T + 1, dtype=float)
    conc = np.stack([np.full(T, 400.0), np.full(T, 1800.0), np.full(T, 330.0)])
    with ra.Ensemble(ra.KIND_GHG_FORCING, n, b) as e:
        P = np.repeat(orc.ghg_default_params().reshape(-1, 1), n, axis=1)
        P[0, 3] = 0.0  # one member on the other method
        with pytest.raises(ra.RscmGpuError, match="method"):
            e.set_params(P)
        P[0] = 2.0
        with pytest.raises(ra.RscmGpuError, match="method"):
            e.set_params(P)
        with pytest.raises(ValueError, match="input block"):
            e.set_forcing(conc[:2])
        P[0] = 1.0
        e.set_params(P)
        with pytest.raises(ra.RscmGpuError, match="input series not set"):
            e.run()
        e.set_forcing(conc)
        e.run()
        assert e.finished()
        # constant concentrations: every year carries the same forcing
        s = e.get_series(1)
        assert np.isnan(s[0]).all() and (s[1:] == s[1]).all() and (s[1] > 0).all()


def test_ghg_full_size_properties(ra, orc):
    """1e6 members x 751 years (the size BASELINE.json's configs use): the rapid adjustment is the
    last factor applied (forcing/ghg.rs:283-289), so a run with adjust = a equals the adjust = 1 run
    times a, bit for bit; members that share parameters and scenario agree exactly; a sample of
    members matches the oracle."""
    n, T = 1_000_000, 751
    yr = np.arange(T)
    conc = np.stack([np.stack([278.0 * 1.002 ** yr, 722.0 + 2.0 * yr, 270.0 + 0.1 * yr]),
                     np.stack([278.0 + 0.3 * yr, 722.0 + 1.0 * yr, 270.0 + 0.05 * yr])])
    rng = np.random.default_rng(11)
    base = orc.ghg_default_params(method="Olbl", adjust_co2=1.0, adjust_ch4=1.0, adjust_n2o=1.0)
    P = np.repeat(base.reshape(-1, 1), n, axis=1)
    P[1] = rng.uniform(275.0, 281.0, n)
    P[1, n // 2:] = P[1, : n // 2]  # the second half repeats the first half's parameters
    scen = np.tile(np.arange(2, dtype=np.int32), n // 2)
    b = np.arange(T + 1, dtype=float) + 1750.0
    with ra.Ensemble(ra.KIND_GHG_FORCING, n, b) as e:
        e.set_params(P)
        e.set_forcing(conc, scen)
        e.run()
        raw = {k: e.get_series(v, 1, T, 50) for k, v in NAMES.items()}
        assert np.isnan(e.get_series(1, 0, 1)).all()
        adj = rng.uniform(0.8, 1.2, (3, n))
        P[18:21] = adj
        e.set_params(P)
        e.rewind()
        e.run()
        for j, (k, v) in enumerate(NAMES.items()):
            assert np.array_equal(e.get_series(v, 1, T, 50), raw[k] * adj[j]), k
    for k in NAMES:
        assert np.array_equal(raw[k][:, : n // 2], raw[k][:, n // 2:]), k
    pick = rng.choice(n, 64, replace=False)
    P[18:21] = 1.0
    want = orc.ghg_run(T, P[:, pick].copy(), conc, scen=scen[pick])
    for k in NAMES:
        w = want[k][1:T:50]
        assert (np.abs(raw[k][:, pick] - w) <= TOL * np.maximum(1.0, np.abs(w))).all(), k


def test_udeb_runs_on_a_monthly_axis(ra, orc):
    """The RK4 landing check of the two-layer kinds (ode_solvers' end-time assertion) does not
    apply to kinds without an RK4 component: ClimateUDEB takes any increasing axis."""
    b = 1850.0 + np.arange(0, 121) / 12.0
    erf = np.full(120, 3.71)
    P = orc.udeb_default_params().reshape(-1, 1)
    with ra.Ensemble(ra.KIND_UDEB, 1, b) as e:
        e.set_params(P)
        e.set_forcing(erf)
        for v in (1, 2, 3, 4):
            e.set_initial(v, 0.0)
        e.run()
        got = e.get_series(7)[:, 0]
    want, st = orc.udeb_run(b, P, erf)
    assert st[0] == 0
    assert np.abs(got[1:] - want["sst"][1:, 0]).max() <= 1e-9


@pytest.mark.parametrize("method", ["Ipcctar", "Olbl"])
def test_member_constants_follow_every_way_of_writing_the_parameters(ra, orc, method):
    """GhgForcing's member constants (ln C0, sqrt M0, sqrt N0, the pre-industrial powers and overlap: formed once per parameter set by
    launch_ghg_derive and stored beside the parameter block) must be re-formed after ANY write of the block: a second
    rscm_ens_set_params on the same handle, rows that go from uniform to varied and back, rscm_ens_sample_lhs, and values written
    straight through rscm_ens_params_devptr (then before every run).  Each run is held against the oracle with the parameters the
    device block holds at that moment."""
    import ctypes as C
    from rscm_amd import _lib as L
    T, n = 120, 257
    yr = np.arange(T)
    conc = np.stack([278.0 * 1.006 ** yr, 722.0 + 6.0 * yr, 270.0 + 0.4 * yr])[None]
    b = np.arange(T + 1, dtype=float) + 1750.0
    P1 = _ensemble(orc, n, method, seed=11)
    P2 = _ensemble(orc, n, method, seed=12)
    U = np.repeat(P1[:, :1], n, axis=1).copy()   # every row uniform: the constants come through scalar loads
    with ra.Ensemble(ra.KIND_GHG_FORCING, n, b) as e:
        e.set_forcing(conc)

        def check(P, what):
            e.rewind()
            e.run()
            got = {k: e.get_series(v) for k, v in NAMES.items()}
            _assert_close(got, orc.ghg_run(T, np.ascontiguousarray(P), conc), f"{method}: {what}")

        e.set_params(P1)
        check(P1, "first parameter set")
        e.set_params(P2)
        check(P2, "second parameter set on the same handle")
        e.set_params(U)
        check(U, "uniform rows")
        e.set_params(P1)
        check(P1, "varied again")
        lo, hi = P1.min(axis=1), P1.max(axis=1)
        e.sample_lhs(5, lo, hi)
        check(e.get_params(), "device Latin hypercube")
        ptr = C.c_void_p()
        L.check(e._lib.rscm_ens_params_devptr(e._h, C.byref(ptr)))
        for P in (P2, P1):   # written behind the library's back, twice: the constants are re-formed before every run of such a handle
            Pc = np.ascontiguousarray(P)
            L.check(e._lib.rscm_gpu_copy_to_device(0, ptr, Pc.ctypes.data_as(C.c_void_p), Pc.nbytes))
            check(P, "written through the device pointer")


def test_member_constants_are_formed_once_per_call_not_once_per_step(ra, orc):
    """A handle whose parameter block the caller holds a device pointer to is re-derived before every RUN.  A lock-step run of many
    one-step launches is one run: the member constants are formed once at its start (advisor, round 4: they were formed before
    every step, and outside the timed region) -- counted through rscm_gpu_derive_launches -- and the results are the oracle's."""
    import ctypes as C
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    T, n = 120, 257
    yr = np.arange(T)
    conc = np.stack([278.0 * 1.006 ** yr, 722.0 + 6.0 * yr, 270.0 + 0.4 * yr])[None]
    b = np.arange(T + 1, dtype=float) + 1750.0
    P1 = _ensemble(orc, n, "Ipcctar", seed=21)
    P2 = np.ascontiguousarray(_ensemble(orc, n, "Ipcctar", seed=22))
    count = C.c_int64()
    with ra.Ensemble(ra.KIND_GHG_FORCING, n, b) as e:
        e.set_forcing(conc)
        e.set_params(P1)
        ptr = C.c_void_p()
        L.check(e._lib.rscm_ens_params_devptr(e._h, C.byref(ptr)))
        L.check(e._lib.rscm_gpu_copy_to_device(0, ptr, P2.ctypes.data_as(C.c_void_p), P2.nbytes))
        L.check(e._lib.rscm_gpu_derive_launches(C.byref(count)))      # reset
        run_lockstep((e,), 40)                                          # 40 one-step launches, one call
        L.check(e._lib.rscm_gpu_derive_launches(C.byref(count)))
        assert count.value == 1, count.value
        e.run()                                                         # the rest of the axis: another run, another derive
        L.check(e._lib.rscm_gpu_derive_launches(C.byref(count)))
        assert count.value == 1 and e.last_run_ms() > 0
        got = {k: e.get_series(v) for k, v in NAMES.items()}
        _assert_close(got, orc.ghg_run(T, P2, conc), "parameters written through the device pointer, stepped in lock-step")
