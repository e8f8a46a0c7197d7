"""The GhgForcing oracle (oracle/ghg_oracle.c) against what the reference holds for it: the
MAGICC7 outputs of tests/regression/test_ghg_forcing.py (fixture tests/golden/ghg_forcing_magicc7.json,
made by tests/golden/make_ghg_goldens.py) at that file's own tolerances, and the known answers of
the unit tests in crates/rscm-magicc/src/forcing/ghg.rs:368-727.  The same fixture's ECS sweep
and CO2-only scenarios pin the ClimateUDEB oracle a second time (ERF -> temperature)."""
import json
import os

import numpy as np
import pytest

from oracle import cbind as orc
from tests.test_oracle_udeb import phased_ok

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ghg_forcing_magicc7.json")))
RTOL, ATOL = 1e-5, 1e-6  # tests/regression/test_ghg_forcing.py DEFAULT_RTOL / DEFAULT_ATOL


def ghg_params_from_config(cfg, co2, ch4, n2o):
    """build_ghg_forcing_model (test_ghg_forcing.py:139-182): method and adjustments from the
    MAGICC config, pre-industrial values from the first year."""
    if cfg.get("core_co2ch4n2o_rfmethod", "IPCCTAR") == "IPCCTAR":
        method, adj = "Ipcctar", (1.0, 1.0, 1.0)
    else:
        method, adj = "Olbl", (1.05, 0.86, 1.0)
    return orc.ghg_default_params(
        method=method, delq2xco2=cfg.get("core_delq2xco2", 3.71), co2_pi=co2[0], ch4_pi=ch4[0], n2o_pi=n2o[0],
        adjust_co2=cfg.get("core_rfrapidadjust_co2", adj[0]), adjust_ch4=cfg.get("core_rfrapidadjust_ch4", adj[1]),
        adjust_n2o=cfg.get("core_rfrapidadjust_n2o", adj[2]))


def scenario_concentrations(g):
    return [np.array(g["Atmospheric Concentrations|" + s]) for s in ("CO2", "CH4", "N2O")]


@pytest.mark.parametrize("name", ["01_concentration_driven", "02_ghg_forcing_olbl"])
def test_ghg_oracle_matches_magicc7(name):
    g = GOLD[name]
    co2, ch4, n2o = scenario_concentrations(g)
    p = ghg_params_from_config(g["config"], co2, ch4, n2o)
    out = orc.ghg_run(len(co2), p, np.stack([co2, ch4, n2o]))
    for var, key in (("co2_erf", "CO2"), ("ch4_erf", "CH4"), ("n2o_erf", "N2O")):
        got = out[var][:, 0]
        assert np.isnan(got[0])  # index 0 is the initial state
        # solve results start at index 1: actual[1:] aligns with expected[:-1]
        np.testing.assert_allclose(got[1:], np.array(g["Effective Radiative Forcing|" + key])[:-1],
                                   rtol=RTOL, atol=ATOL, err_msg=f"{name} {key}")


def test_ghg_unit_test_known_answers():
    ip = orc.ghg_default_params(method="Ipcctar", adjust_co2=1.0, adjust_ch4=1.0, adjust_n2o=1.0)
    ol = orc.ghg_default_params(method="Olbl", adjust_co2=1.0, adjust_ch4=1.0, adjust_n2o=1.0)
    f = lambda p, *c: orc.ghg_forcings(p, *c)  # noqa: E731
    for p in (ip, ol):  # zero at pre-industrial, all three gases
        z = f(p, 278.0, 722.0, 270.0)
        assert all(abs(v) < 1e-10 for v in z.values())
    assert abs(f(ip, 556.0, 722.0, 270.0)["co2_erf"] - 3.71) < 0.01  # 2xCO2
    assert abs(f(ip, 1112.0, 722.0, 270.0)["co2_erf"] - 2 * f(ip, 556.0, 722.0, 270.0)["co2_erf"]) < 0.01
    assert 0.3 < f(ip, 278.0, 1900.0, 270.0)["ch4_erf"] < 0.8  # modern CH4 (AR6 ~0.54)
    assert 0.1 < f(ip, 278.0, 722.0, 332.0)["n2o_erf"] < 0.4  # modern N2O
    # overlap reduces the direct square-root terms
    assert f(ip, 278.0, 1900.0, 270.0)["ch4_erf"] < 0.036 * (np.sqrt(1900.0) - np.sqrt(722.0))
    assert f(ip, 278.0, 722.0, 332.0)["n2o_erf"] < 0.12 * (np.sqrt(332.0) - np.sqrt(270.0))
    # OLBL differs from IPCCTAR at 560 ppm, but by less than 1 W/m^2
    d = abs(f(ip, 560.0, 722.0, 270.0)["co2_erf"] - f(ol, 560.0, 722.0, 270.0)["co2_erf"])
    assert 1e-4 < d < 1.0
    for p in (ip, ol):
        assert all(v > 0 for v in f(p, 400.0, 1900.0, 332.0).values())
    # rapid adjustments are plain factors
    adj = orc.ghg_default_params(method="Olbl", adjust_co2=1.05, adjust_ch4=0.86, adjust_n2o=0.93)
    a, r = f(adj, 400.0, 1900.0, 332.0), f(ol, 400.0, 1900.0, 332.0)
    assert abs(a["co2_erf"] - r["co2_erf"] * 1.05) < 1e-10
    assert abs(a["ch4_erf"] - r["ch4_erf"] * 0.86) < 1e-10
    assert abs(a["n2o_erf"] - r["n2o_erf"] * 0.93) < 1e-10


def test_ghg_olbl_co2_regimes():
    """forcing/ghg.rs:210-240: alpha is the full quadratic between C0 and the vertex, constant
    beyond it, and drops the concentration terms below C0."""
    p = orc.ghg_default_params(method="Olbl", adjust_co2=1.0)
    a1, b1, c1, d1 = (p[orc.GHG_PARAM_NAMES.index(k)] for k in ("olbl_co2_a1", "olbl_co2_b1", "olbl_co2_c1", "olbl_co2_d1"))
    c_max = 278.0 - b1 / (2 * a1)
    n2o = 300.0
    for co2 in (c_max + 1.0, c_max + 500.0):
        alpha = -b1 * b1 / (4 * a1) + d1 + c1 * np.sqrt(n2o)
        assert orc.ghg_forcings(p, co2, 722.0, n2o)["co2_erf"] == pytest.approx(alpha * np.log(co2 / 278.0), rel=1e-14)
    co2 = 200.0
    assert orc.ghg_forcings(p, co2, 722.0, n2o)["co2_erf"] == pytest.approx((d1 + c1 * np.sqrt(n2o)) * np.log(co2 / 278.0), rel=1e-14)
    co2 = 0.5 * (278.0 + c_max)
    dc = co2 - 278.0
    alpha = a1 * dc * dc + b1 * dc + d1 + c1 * np.sqrt(n2o)
    assert orc.ghg_forcings(p, co2, 722.0, n2o)["co2_erf"] == pytest.approx(alpha * np.log(co2 / 278.0), rel=1e-14)


def test_ghg_run_scenarios_and_members():
    rng = np.random.default_rng(5)
    T, N = 40, 33
    conc = np.stack([np.stack([278 + 3 * np.arange(T) * (s + 1), 722 + 20 * np.arange(T), 270 + np.arange(T)]) for s in range(2)])
    P = np.repeat(orc.ghg_default_params().reshape(-1, 1), N, axis=1)
    P[orc.GHG_PARAM_NAMES.index("adjust_co2")] = rng.uniform(0.9, 1.1, N)
    P[orc.GHG_PARAM_NAMES.index("co2_pi")] = rng.uniform(270, 285, N)
    scen = (np.arange(N) % 2).astype(np.int32)
    out = orc.ghg_run(T, P, conc, scen=scen, threads=4)
    for i in (0, 7, 32):
        for n in (0, 11, T - 2):
            want = orc.ghg_forcings(P[:, i].copy(), *conc[scen[i], :, n])
            for k in orc.GHG_VARS:
                assert out[k][n + 1, i] == want[k]
    assert all(np.isnan(out[k][0]).all() for k in orc.GHG_VARS)


@pytest.mark.parametrize("name", [k for k in GOLD if k.startswith(("04_", "05_"))])
def test_udeb_oracle_against_ghg_suite_temperatures(name):
    """test_04_ecs_sweep / test_05_co2_only_forcing (test_ghg_forcing.py:732-830): the scenario's
    ERF drives ClimateUDEB alone; the four-box global mean must follow MAGICC7's Surface
    Temperature within the phased tolerances 5e-2 / 3e-2 / 3e-2 (atol 1e-6)."""
    g = GOLD[name]
    years = np.array(g["years"], dtype=float)
    erf = np.array(g["Effective Radiative Forcing" if name.startswith("05_") else "Effective Radiative Forcing|CO2"])
    P = orc.udeb_default_params(ecs=g["config"]["core_climatesensitivity"], rf_2xco2=g["config"]["core_delq2xco2"])
    out, st = orc.udeb_run(np.append(years, years[-1] + 1.0), P, erf)
    assert st[0] == 0
    # build_erf_to_temperature_model interpolates the ERF linearly, i.e. the component sees the
    # scenario's own values at the step boundaries -- exactly what udeb_run consumes
    idx = orc.UDEB_PARAM_NAMES.index
    fgnl, fgsl = P[idx("nh_land_fraction")] / 2.0, P[idx("sh_land_fraction")] / 2.0
    w = np.array([0.5 - fgnl, fgnl, 0.5 - fgsl, fgsl])
    temp = sum(w[k] * out[f"st{k}"][:, 0] for k in range(4))
    ok, msg = phased_ok(temp, np.array(g["Surface Temperature"]), shock_rtol=5e-2, converge_rtol=3e-2, final_rtol=3e-2, atol=1e-6)
    assert ok, f"{name}: {msg}"
