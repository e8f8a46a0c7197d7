"""The CH4Chemistry / N2OChemistry oracle (oracle/chem_oracle.c) against the known answers of the
reference's in-file unit tests (crates/rscm-magicc/src/chemistry/ch4.rs:332-652,
chemistry/n2o.rs:262-647) and closed forms of the update rule.  The reference holds no golden
vectors for these components."""
import numpy as np
import pytest

from oracle import cbind as orc

CH4, N2O = orc.CHEM_CH4, orc.CHEM_N2O


def test_ch4_unit_test_answers():
    p = orc.chem_default_params(CH4)
    s = lambda *a, q=p: orc.ch4_solve_concentration(q, *a)  # noqa: E731  (prev, cur, E, T, NOx, CO, NMVOC)
    pi = 722.0
    assert abs(s(pi, pi, 0, 0, 0, 0, 0)[0] - pi) / pi < 0.05  # natural emissions balance the sinks
    assert s(pi, pi, 300.0, 0, 0, 0, 0)[0] > pi
    assert s(pi, pi, 400.0, 0, 0, 0, 0)[0] > s(pi, pi, 200.0, 0, 0, 0, 0)[0]
    assert s(pi, pi, 300.0, 2.0, 0, 0, 0)[1] < s(pi, pi, 300.0, 0.0, 0, 0, 0)[1]  # warming shortens the lifetime
    off = orc.chem_default_params(CH4, include_temp_feedback=0.0)
    assert abs(s(pi, pi, 300.0, 2.0, 0, 0, 0, q=off)[1] - s(pi, pi, 300.0, 0.0, 0, 0, 0, q=off)[1]) < 1e-10
    assert s(1800.0, 1800.0, 300.0, 0, 0, 0, 0)[1] > s(pi, pi, 0.0, 0, 0, 0, 0)[1]  # self-feedback
    assert s(pi, pi, 300.0, 0, 50.0, 0, 0)[1] < s(pi, pi, 300.0, 0, 0, 0, 0)[1]  # NOx makes OH
    assert s(pi, pi, 300.0, 0, 0, 1000.0, 0)[1] > s(pi, pi, 300.0, 0, 0, 0, 0)[1]  # CO consumes it
    c, tau = s(1500.0, 1500.0, 350.0, 1.0, 30.0, 500.0, 100.0)
    assert 0 < c < 5000 and 5 < tau < 20
    assert s(1800.0, 1800.0, 0, 0, 0, 0, 0, q=orc.chem_default_params(CH4, natural_emissions=0.0))[0] < 1800.0
    c, tau = s(10000.0, 10000.0, 300.0, 0, 0, 0, 0)
    assert c > 0 and np.isfinite(c) and np.isfinite(tau)
    c, tau = s(100.0, 100.0, 50.0, 0, 0, 0, 0)
    assert c > 0 and tau > 0
    # negative temperature anomalies are clamped to zero (ch4.rs:95-97)
    assert s(pi, pi, 300.0, -1.5, 0, 0, 0) == s(pi, pi, 300.0, 0.0, 0, 0, 0)


def test_ch4_closed_form_without_feedbacks():
    """No self-, emission- or temperature feedback: tau_oh is the constant base value and each
    Prather pass is B <- B_prev + E - (B + B_prev)/2 * (1/tau_oh + 1/tau_other)."""
    p = orc.chem_default_params(CH4, ch4_self_feedback=0.0, include_temp_feedback=0.0, include_emissions_feedback=0.0)
    k = 1 / 9.3 + (1 / 150.0 + 1 / 120.0 + 1 / 200.0)
    b_prev, b = 1500.0 * 2.75, 1600.0 * 2.75
    E = 400.0 + 209.0
    for _ in range(4):
        b = b_prev + (E - (b + b_prev) / 2.0 * k)
    c, tau = orc.ch4_solve_concentration(p, 1500.0, 1600.0, 400.0, 3.0, 10.0, 100.0, 50.0)
    assert c == pytest.approx(b / 2.75, rel=1e-13) and tau == pytest.approx(1 / k, rel=1e-13)


def test_n2o_unit_test_answers():
    p = orc.chem_default_params(N2O)
    c, tau = orc.n2o_solve_concentration(p, 270.0, 270.0, 270.0, 0.0, 1.0)
    assert abs(tau - 139.275) / 139.275 < 0.01 and abs(c - 270.0) / 270.0 < 0.05
    # steady state at 320 ppb: emissions that balance B/tau keep the concentration (n2o.rs:317-350)
    burden = 320.0 * 4.79
    tau320 = 139.275 * (burden / (270.0 * 4.79)) ** -0.04
    c, tau = orc.n2o_solve_concentration(p, 320.0, 320.0, 320.0, burden / tau320 - 11.0, 1.0)
    assert abs(c - 320.0) < 1e-9 and tau == pytest.approx(tau320, rel=1e-14)
    assert orc.n2o_solve_concentration(p, 300.0, 300.0, 300.0, 10.0, 1.0)[0] > orc.n2o_solve_concentration(p, 300.0, 300.0, 300.0, 5.0, 1.0)[0]
    assert orc.n2o_solve_concentration(p, 350.0, 350.0, 350.0, 5.0, 1.0)[1] < orc.n2o_solve_concentration(p, 280.0, 280.0, 280.0, 5.0, 1.0)[1]
    # the burden change scales with the step length (n2o.rs:84-89)
    d1 = orc.n2o_solve_concentration(p, 300.0, 300.0, 300.0, 8.0, 1.0)[0] - 300.0
    d12 = orc.n2o_solve_concentration(p, 300.0, 300.0, 300.0, 8.0, 1.0 / 12.0)[0] - 300.0
    assert d12 == pytest.approx(d1 / 12.0, rel=1e-3)
    # below pre-industrial the ratio is floored at 1: the base lifetime
    assert orc.n2o_solve_concentration(p, 200.0, 200.0, 200.0, 0.0, 1.0)[1] == 139.275


def test_chem_run_history_indexing():
    """previous() / at_offset(-delay) fall back exactly as chemistry/n2o.rs:196-218 and
    chemistry/ch4.rs:311-313 do when the history is shorter than the lag."""
    T = 12
    b = np.arange(T + 1, dtype=float) + 2000.0
    e = np.linspace(0.0, 11.0, T)
    for delay in (1, 3):
        p = orc.chem_default_params(N2O, strat_delay=delay)
        conc, life = orc.chem_run(N2O, b, p, e[None, None, :], 290.0)
        c = [290.0]
        for n in range(T - 1):
            prev = c[n - 1] if n > 0 else c[n]
            t_d = c[n - delay] if n - delay >= 0 else prev
            t_d1 = c[n - delay - 1] if n - delay - 1 >= 0 else t_d
            new, tau = orc.n2o_solve_concentration(p, prev, c[n], (t_d + t_d1) / 2.0, e[n], 1.0)
            c.append(new)
            assert life[n + 1, 0] == tau
        assert np.array_equal(conc[:, 0], np.array(c)) and np.isnan(life[0, 0])
    p = orc.chem_default_params(CH4)
    x = np.stack([np.linspace(100, 400, T), np.linspace(0, 1.5, T), np.full(T, 20.0), np.full(T, 300.0), np.full(T, 80.0)])
    conc, life = orc.chem_run(CH4, b, p, x[None], 800.0)
    c = [800.0]
    for n in range(T - 1):
        new, tau = orc.ch4_solve_concentration(p, c[n - 1] if n > 0 else c[n], c[n], *x[:, n])
        c.append(new)
        assert life[n + 1, 0] == tau
    assert np.array_equal(conc[:, 0], np.array(c))
