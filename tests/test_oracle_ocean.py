"""The OceanCarbon oracle (oracle/ocean_oracle.c) against the known answers of the reference's
unit tests (crates/rscm-magicc/src/carbon/ocean.rs:218-927, parameters/ocean_carbon.rs:302-848)
and its integration tests (crates/rscm-magicc/tests/conservation.rs:96-168,
tests/carbon_cycle_physics.rs:330-470).  The reference holds no golden vectors for it."""
import numpy as np
import pytest

from oracle import cbind as orc


def test_ocean_parameter_functions():
    for model in orc.OCEAN_MODELS:
        p = orc.ocean_default_params(model)
        irf = [orc.ocean_irf(p, t) for t in (0.0, 0.5, 1.0, 5.0, 10.0, 100.0, 499.0)]
        assert irf[0] == pytest.approx(1.0, abs=2e-3)  # a unit pulse starts in the mixed layer
        assert all(a > b for a, b in zip(irf, irf[1:])) and 0.0 < irf[-1] < 0.1  # and decays monotonically
    p = orc.ocean_default_params()
    # scale_irf(raw) = raw f / (raw f + 1 - raw), here on the GFDL late form at t = 2 yr
    raw = sum(a * np.exp(-2.0 / tau) for a, tau in zip((0.01481, 0.019439, 0.038344, 0.066485, 0.24966, 0.70367),
                                                        (1.0e10, 347.55, 65.359, 15.281, 2.3488, 0.70177)))
    f = 0.9492864
    assert orc.ocean_irf(p, 2.0) == pytest.approx(raw * f / (raw * f + 1 - raw), rel=1e-14)
    # Joos A25: ~3.79 %/K, multiplicative (carbon_cycle_physics.rs:339-376)
    p0, p1, p2 = (orc.ocean_pco2(p, 0.0, t) for t in (0.0, 1.0, 2.0))
    assert 3.5 < (p1 / p0 - 1.0) * 100.0 < 4.5 and p2 / p0 == pytest.approx((p1 / p0) ** 2, abs=1e-3) and p0 == 278.0
    assert orc.ocean_pco2(orc.ocean_default_params(enable_temp_feedback=0.0), 5.0, 3.0) == 283.0
    # Revelle buffering: pCO2 rises faster than DIC (carbon_cycle_physics.rs:430-458), Joos A24 at small DIC
    assert orc.ocean_delta_pco2_from_dic(p, 50.0) / orc.ocean_delta_pco2_from_dic(p, 10.0) > 5.0
    assert orc.ocean_delta_pco2_from_dic(p, 1e-6) == pytest.approx((1.5568 - 0.013993 * 17.7) * 1e-6, rel=1e-9)
    assert orc.ocean_delta_pco2_from_dic(p, 0.0) == 0.0


def test_ocean_solve_unit_test_answers():
    p = orc.ocean_default_params()
    one = orc.ocean_solve_repeated(p, 400.0, 0.0, 278.0, 0.0, 1.0, 1)[0]
    assert one[0] > 278.0 and one[1] > 0.0 and one[2] > 0.0  # uptake raises ocean pCO2
    warm = orc.ocean_solve_repeated(p, 400.0, 2.0, 278.0, 0.0, 1.0, 1)[0]
    assert warm[2] < one[2] and warm[1] < one[1]  # warming reduces the uptake
    eq = orc.ocean_solve_repeated(p, 278.0, 0.0, 278.0, 0.0, 1.0, 3)
    assert np.all(eq[:, 1:] == 0.0) and np.all(eq[:, 0] == 278.0)  # equilibrium stays put
    # cumulative uptake equals the integrated flux over 50 years (conservation.rs:96-121)
    run = orc.ocean_solve_repeated(p, 400.0, 0.0, 278.0, 0.0, 1.0, 50)
    assert abs(run[-1, 1] - run[:, 2].sum()) < 1.0 and run[-1, 1] == pytest.approx(run[:, 2].sum(), rel=1e-12)
    assert np.all(np.diff(run[:, 0]) > 0) and run[-1, 0] < 400.0
    # 10 years at +3 K take up less than at 0 K (carbon_cycle_physics.rs:379-427)
    assert orc.ocean_solve_repeated(p, 400.0, 3.0, 278.0, 0.0, 1.0, 10)[-1, 1] < orc.ocean_solve_repeated(p, 400.0, 0.0, 278.0, 0.0, 1.0, 10)[-1, 1]


def test_ocean_first_substeps_by_hand():
    """Two monthly sub-steps written out: flux from the gas-exchange rate, the convolution with
    the scaled response at lags 0 and 1/12 yr, Joos A24 and A25."""
    p = orc.ocean_default_params(steps_per_year=2)
    k = 1.833492 / (7.66 * 12.0)
    conv = 1.72e17 / (50.9 * 3.55e14)
    f0 = k * (400.0 - 278.0)
    d0 = f0 * orc.ocean_irf(p, 0.0) * conv
    pco2_1 = orc.ocean_pco2(p, orc.ocean_delta_pco2_from_dic(p, d0), 0.5)
    f1 = k * (400.0 - pco2_1)
    d1 = (f0 * orc.ocean_irf(p, 1.0 / 12.0) + f1 * orc.ocean_irf(p, 0.0)) * conv
    pco2_2 = orc.ocean_pco2(p, orc.ocean_delta_pco2_from_dic(p, d1), 0.5)
    out = orc.ocean_solve_repeated(p, 400.0, 0.5, 278.0, 10.0, 1.0, 1)[0]
    assert out[0] == pco2_2
    assert out[2] == (f0 * 12.0 * 2.124) / 2.0 + (f1 * 12.0 * 2.124) / 2.0
    assert out[1] == (10.0 + f0 * 12.0 * 2.124 * 0.5) + f1 * 12.0 * 2.124 * 0.5


def test_ocean_history_is_bounded():
    """max_history_months truncates the convolution (carbon/ocean.rs:128-131): with a 6-month
    window, year 2 onwards sees only the last six pulses."""
    p = orc.ocean_default_params(max_history_months=6)
    q = orc.ocean_default_params()
    a = orc.ocean_solve_repeated(p, 420.0, 0.0, 278.0, 0.0, 1.0, 3)
    b = orc.ocean_solve_repeated(q, 420.0, 0.0, 278.0, 0.0, 1.0, 3)
    assert a[0, 0] < b[0, 0] and a[2, 0] < b[2, 0]  # forgetting old pulses keeps ocean pCO2 lower
    assert a[2, 2] > b[2, 2]  # and the uptake higher
    # zero-length history: no DIC anomaly at all
    z = orc.ocean_solve_repeated(orc.ocean_default_params(max_history_months=0), 420.0, 0.0, 278.0, 0.0, 1.0, 2)
    assert np.all(z[:, 0] == 278.0)


def test_ocean_run_layout_and_state_persistence():
    T, N = 9, 5
    b = np.arange(T + 1, dtype=float) + 2000.0
    x = np.stack([np.stack([300.0 + 10.0 * np.arange(T), 0.1 * np.arange(T)]), np.stack([np.full(T, 350.0), np.zeros(T)])])
    P = np.repeat(orc.ocean_default_params().reshape(-1, 1), N, axis=1)
    P[orc.OCEAN_PARAM_NAMES.index("gas_exchange_tau")] = np.linspace(6.0, 9.0, N)
    scen = (np.arange(N) % 2).astype(np.int32)
    out = orc.ocean_run(b, P, x, 278.0, 0.0, scen=scen, threads=2)
    assert out.shape == (3, T, N) and np.isnan(out[2, 0]).all() and (out[0, 0] == 278.0).all()
    # member 1 sees constant inputs: identical to the repeated stand-alone calls with one history
    ref = orc.ocean_solve_repeated(P[:, 1].copy(), 350.0, 0.0, 278.0, 0.0, 1.0, T - 1)
    assert np.array_equal(out[:, 1:, 1].T, ref)
