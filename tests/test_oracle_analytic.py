"""Build-supplied pins for the oracle where the reference holds no numeric vector
(SURVEY.md section 8c): closed-form linear solution, O(h^4) convergence, energy identity,
equilibrium, and C restatement == generic Python stepper, bit for bit."""
import math

import numpy as np

from oracle import cbind
from oracle import reference_model as rm
from tests.helpers import (assert_bit_equal, axis_values, coupled_params, emissions_syn, f_syn,
                           two_layer_params)

P_DEFAULT = [1.0, 0.0, 1.0, 0.7, 8.0, 100.0]  # configs/two-layer/defaults.toml:21-42


def _exact_linear(p, F, ts0, td0, dt):
    """a = 0: y' = A y + b with constant F -> y(dt) = y_eq + expm(A dt) (y0 - y_eq)."""
    lam, _, eps, eta, cs, cd = p
    A = np.array([[-(lam + eps * eta) / cs, eps * eta / cs], [eta / cd, -eta / cd]])
    yeq = np.array([F / lam, F / lam])
    w, V = np.linalg.eig(A)
    c = np.linalg.solve(V, np.array([ts0, td0]) - yeq)
    return yeq + (V * np.exp(w * dt)) @ c


def test_linear_closed_form_and_order4():
    ts = td = 0.0
    errs = []
    for h in (0.2, 0.1, 0.05):
        a, b, _ = cbind.two_layer_solve(P_DEFAULT, 4.0, 0.0, 10.0, h, ts, td)
        ex = _exact_linear(P_DEFAULT, 4.0, ts, td, 10.0)
        errs.append(max(abs(a - ex[0]), abs(b - ex[1])))
    assert errs[1] < 1e-7
    assert 12.0 < errs[0] / errs[1] < 20.0 and 12.0 < errs[1] / errs[2] < 20.0


def test_piecewise_constant_forcing_750yr_vs_closed_form():
    t = axis_values()
    b = cbind.bounds_from_values(t)
    F = f_syn(t)
    params = np.array(P_DEFAULT).reshape(6, 1)
    ts, td = cbind.two_layer_run(b, params, F, 0.0, 0.0)
    y = np.zeros(2)
    worst = 0.0
    for n in range(len(t) - 1):
        y = _exact_linear(P_DEFAULT, F[n], y[0], y[1], 1.0)
        worst = max(worst, abs(ts[n + 1, 0] - y[0]), abs(td[n + 1, 0] - y[1]))
    assert worst < 1e-6, worst  # RK4 h=0.1 global error, measured ~1e-8


def test_energy_identity_heat_integral():
    """The discarded y[2] integrates Cs*dTs + Cd*dTd, so it equals Cs*dTs + Cd*dTd exactly
    up to rounding."""
    p = [1.2, 0.03, 1.4, 0.8, 9.0, 120.0]
    a, b, heat = cbind.two_layer_solve(p, 3.0, 2000.0, 2001.0, 0.1, 0.5, 0.2)
    assert math.isclose(heat, p[4] * (a - 0.5) + p[5] * (b - 0.2), rel_tol=1e-12)


def test_equilibrium():
    ts, td = 0.0, 0.0
    for _ in range(200):  # slow mode e-folds in ~240 yr
        ts, td, _ = cbind.two_layer_solve(P_DEFAULT, 3.7, 0.0, 100.0, 0.1, ts, td)
    assert abs(ts - 3.7) < 1e-9 and abs(td - 3.7) < 1e-9


def test_zero_forcing_stays_exactly_zero():
    t = axis_values(1750, 1800)
    ts, td = cbind.two_layer_run(cbind.bounds_from_values(t), two_layer_params(7),
                                 np.zeros_like(t), 0.0, 0.0)
    assert not ts.any() and not td.any()


def test_c_oracle_equals_generic_python_stepper_two_layer():
    t = axis_values(1750, 1790)
    F = f_syn(t)
    P = two_layer_params(5)
    for source in (0, 1):
        ts, td = cbind.two_layer_run(cbind.bounds_from_values(t), P, F, 0.3, -0.1, source=source)
        for i in range(P.shape[1]):
            comps = [rm.TwoLayer(*P[:, i])]
            if source == 1:  # an upstream producer of ERF registered BEFORE TwoLayer
                class Fprod(rm.Component):
                    type_name = "ForcingProducer"
                    defs = [rm.Req("Effective Radiative Forcing", rm.OUTPUT)]

                    def solve(self, t0, t1, w, _F=F, _t=t):
                        return {"Effective Radiative Forcing": _F[int(round(t1 - _t[0]))]}
                comps = [Fprod()] + comps
                exo = {}
            else:
                exo = {"Effective Radiative Forcing":
                       rm.ExoSeries(list(F), rm.TimeAxis.from_values(t), "Linear")}
            m = rm.ModelBuilder(axis=rm.TimeAxis.from_values(t), components=comps,
                                initial_values={"Surface Temperature": 0.3,
                                                "Deep Ocean Temperature": -0.1},
                                exogenous=exo).build()
            want = rm.UPSTREAM_OUTPUT if source else rm.EXOGENOUS
            assert m.sources[("Effective Radiative Forcing", "TwoLayer")] == want
            m.run()
            assert_bit_equal(ts[:, i], m.data["Surface Temperature"], f"Ts src={source} i={i}")
            assert_bit_equal(td[:, i], m.data["Deep Ocean Temperature"], f"Td src={source} i={i}")


def test_exogenous_linear_resample_is_identity_except_last_point():
    """A.6 quirk: resampling onto the same axis is exact except that the LAST point goes through
    the forward-extrapolation formula (<= 1 ulp off)."""
    t = axis_values(1750, 1800)
    F = f_syn(t)
    r = rm.interpolate_into("Linear", rm.TimeAxis.from_values(t), list(F), rm.TimeAxis.from_values(t))
    assert r[:-1] == list(F[:-1])
    assert abs(r[-1] - F[-1]) <= 2 * np.spacing(abs(F[-1]))


def test_c_oracle_equals_generic_python_stepper_coupled():
    t = axis_values(1750, 1780)
    E = emissions_syn(t)
    P = coupled_params(3)
    init = dict(ts=0.0, td=0.0, conc=278.0, cum_uptake=0.0, cum_emis=0.0)
    out = cbind.coupled_run(cbind.bounds_from_values(t), P, E, init)
    names = {"ts": "Surface Temperature", "td": "Deep Ocean Temperature",
             "conc": "Atmospheric Concentration|CO2", "cum_uptake": "Cumulative Land Uptake",
             "cum_emis": "Cumulative Emissions|CO2", "erf_co2": "Effective Radiative Forcing|CO2",
             "erf_total": "Effective Radiative Forcing"}
    for i in range(P.shape[1]):
        p = P[:, i]
        m = rm.ModelBuilder(
            axis=rm.TimeAxis.from_values(t),
            components=[rm.CarbonCycle(p[6], p[7], p[8]), rm.CO2ERF(p[9], p[7]),
                        rm.TwoLayer(*p[:6])],
            aggregates=[("Effective Radiative Forcing", "Sum", ["Effective Radiative Forcing|CO2"])],
            exogenous={"Emissions|CO2|Anthropogenic":
                       rm.ExoSeries(list(E), rm.TimeAxis.from_values(t), "Previous")},
            initial_values={"Cumulative Land Uptake": 0.0, "Cumulative Emissions|CO2": 0.0,
                            "Atmospheric Concentration|CO2": 278.0, "Surface Temperature": 0.0,
                            "Deep Ocean Temperature": 0.0}).build()
        # registration-order classification (builder.rs:470-482)
        assert m.sources[("Surface Temperature", "CarbonCycle")] == rm.EXOGENOUS
        assert m.sources[("Atmospheric Concentration|CO2", "CO2ERF")] == rm.UPSTREAM_OUTPUT
        assert m.sources[("Effective Radiative Forcing", "TwoLayer")] == rm.UPSTREAM_OUTPUT
        m.run()
        for k, name in names.items():
            assert_bit_equal(out[k][:, i], m.data[name], f"{k} member {i}")
        assert math.isnan(m.data["Effective Radiative Forcing"][0])


def test_resume_is_bit_identical():
    t = axis_values(1750, 1850)
    b = cbind.bounds_from_values(t)
    P, F = two_layer_params(9), f_syn(t)
    full = cbind.two_layer_run(b, P, F, 0.0, 0.0)
    ts, td = cbind.two_layer_run(b, P, F, 0.0, 0.0, step_end=37)
    assert np.isnan(ts[38:]).all()
    ts, td = cbind.two_layer_run(b, P, F, 0.0, 0.0, step_begin=37, ts=ts, td=td)
    assert_bit_equal(ts, full[0])
    assert_bit_equal(td, full[1])


def test_threaded_split_matches_single_thread():
    t = axis_values(1750, 1800)
    b = cbind.bounds_from_values(t)
    P, F = two_layer_params(1001), f_syn(t)
    one = cbind.two_layer_run(b, P, F, 0.0, 0.0, threads=1)
    many = cbind.two_layer_run(b, P, F, 0.0, 0.0, threads=5)
    assert_bit_equal(one[0], many[0])
    assert_bit_equal(one[1], many[1])


def test_oracle_is_clean_under_asan_ubsan():
    """Sanitizers run on the CPU build only (GPU ASan is unavailable on the pool)."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-C", here, "asan-check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "oracle selftest ok" in r.stdout
