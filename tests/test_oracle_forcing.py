"""The oracle of the three stateless rscm-magicc forcing components (oracle/forcing_oracle.c)
against the known answers of the reference's in-file unit tests:
crates/rscm-magicc/src/forcing/ozone.rs:240-560, aerosol_direct.rs:241-500,
aerosol_indirect.rs:172-420.  The reference holds no golden vectors for these components."""
import numpy as np
import pytest

from oracle import cbind as orc

OZ, AD, AI = orc.PW_OZONE, orc.PW_AEROSOL_DIRECT, orc.PW_AEROSOL_INDIRECT


def test_ozone_unit_test_answers():
    p = orc.pointwise_default_params(OZ)
    f = lambda *x: orc.pointwise_eval(OZ, p, x)  # noqa: E731  (EESC, CH4, NOx, CO, NMVOC, T)
    pi = (1420.0, 700.0, 0.0, 0.0, 0.0, 0.0)
    assert np.all(np.abs(f(*pi)) < 1e-10)  # zero at the reference state
    assert abs(f(1420.0 - 500.0, *pi[1:])[0]) < 1e-10  # below reference EESC: no depletion
    lo, hi = f(1620.0, *pi[1:])[0], f(1820.0, *pi[1:])[0]
    assert hi < lo < 0.0 and abs(hi / lo - 2.0 ** 1.7) < 0.01  # power law in EESC - reference
    assert -0.15 < f(2000.0, *pi[1:])[0] < 0.0  # realistic magnitude at peak EESC
    assert f(1420.0, 1800.0, 40.0, 500.0, 100.0, 0.0)[1] > 0.0
    f2, f4 = f(1420.0, 1400.0, 0, 0, 0, 0)[1], f(1420.0, 2800.0, 0, 0, 0, 0)[1]
    assert abs((f4 - f2) - f2) < 1e-10  # logarithmic in CH4
    assert abs(f(1420.0, 700.0, 20.0, 0, 0, 0)[1] - 2 * f(1420.0, 700.0, 10.0, 0, 0, 0)[1]) < 1e-10  # linear in NOx
    assert 0.2 < f(1420.0, 1900.0, 42.0, 550.0, 110.0, 0.0)[1] < 0.8  # circa 2020
    assert abs(f(*pi[:5], 2.0)[2] - (-0.037 * 2.0)) < 1e-10  # temperature feedback
    assert abs(f(*pi[:5], 2.0)[2] - 2 * f(*pi[:5], 1.0)[2]) < 1e-10
    s, t, fb = f(1800.0, 1900.0, 40.0, 500.0, 100.0, 1.2)
    assert s < 0 < t and fb < 0
    # non-positive CH4 (or CH4_pi) switches the logarithmic term off (ozone.rs:123-127)
    assert f(1420.0, 0.0, 5.0, 0, 0, 0)[1] == 0.032 * (0.168 * 5.0)
    assert orc.pointwise_eval(OZ, orc.pointwise_default_params(OZ, ch4_pi=0.0), (1420.0, 1900.0, 5.0, 0, 0, 0))[1] == 0.032 * (0.168 * 5.0)


def test_aerosol_direct_unit_test_answers():
    p = orc.pointwise_default_params(AD)
    f = lambda *x: orc.pointwise_eval(AD, p, x)  # noqa: E731  (SOx, BC, OC, NOx) -> NO, NL, SO, SL
    pi = (1.0, 2.5, 10.0, 10.0)
    assert np.all(f(*pi) == 0.0)  # |total| < 1e-15: uniform zeros
    sox = f(51.0, *pi[1:])
    assert sox.sum() == pytest.approx(-0.0035 * 50.0, abs=1e-12)  # SOx cools; regions sum to the global value
    assert abs(sox[1]) > abs(sox[2])  # NH land carries more than SH ocean for a pure SOx signal
    assert np.allclose(sox, -0.0035 * 50.0 * np.array([0.15, 0.55, 0.10, 0.20]), rtol=1e-14)
    assert f(1.0, 7.5, 10.0, 10.0).sum() == pytest.approx(0.0077 * 5.0, abs=1e-12)  # BC warms
    assert f(21.0, *pi[1:]).sum() == pytest.approx(2 * f(11.0, *pi[1:]).sum(), abs=1e-12)  # linear
    mixed = f(60.0, 8.0, 35.0, 40.0)
    species = np.array([-0.0035 * 59.0, 0.0077 * 5.5, -0.002 * 25.0, -0.001 * 30.0])
    assert mixed.sum() == pytest.approx(species.sum(), abs=1e-12) and np.all(np.abs(mixed) > 1e-15)
    assert -1.0 < mixed.sum() < 0.5
    # opposing species that cancel exactly: the |total| < 1e-15 branch, not a 0/0
    q = orc.pointwise_default_params(AD, sox_coefficient=-0.01, bc_coefficient=0.01)
    assert np.all(orc.pointwise_eval(AD, q, (2.0, 3.5, 10.0, 10.0)) == 0.0)


def test_aerosol_indirect_unit_test_answers():
    p = orc.pointwise_default_params(AI)
    f = lambda sox, oc: orc.pointwise_eval(AI, p, (sox, oc))[0]  # noqa: E731
    assert abs(f(1.0, 10.0)) < 1e-10  # zero at pre-industrial
    assert f(51.0, 30.0) < 0.0  # cooling above
    assert f(0.5, 5.0) == 0.0  # and none below
    burden = lambda sox, oc: 1.0 * sox + 0.3 * oc  # noqa: E731
    assert f(51.0, 30.0) == pytest.approx(-1.0 * np.log(1.0 + (burden(51.0, 30.0) - burden(1.0, 10.0)) / 50.0), rel=1e-15)
    # logarithmic saturation: each doubling of the excess burden adds less
    a, b, c = f(11.0, 10.0), f(21.0, 10.0), f(41.0, 10.0)  # excess burden 10, 20, 40
    assert c < b < a < 0 and b / a < 2.0 and c / b < b / a


@pytest.mark.parametrize("kind", [OZ, AD, AI, orc.PW_FOURBOX_OHU, orc.PW_OSPP])
def test_pointwise_run_layout(kind):
    rng = np.random.default_rng(kind)
    L = orc.lib()
    ni, no = L.orc_pointwise_n_inputs(kind), L.orc_pointwise_n_outputs(kind)
    T, N = 30, 17
    inputs = rng.uniform(1.0, 2000.0, (2, ni, T))
    P = np.repeat(orc.pointwise_default_params(kind).reshape(-1, 1), N, axis=1)
    P[0] *= rng.uniform(0.9, 1.1, N)
    scen = (np.arange(N) % 2).astype(np.int32)
    out = orc.pointwise_run(kind, T, P, inputs, scen=scen, threads=3)
    assert out.shape == (no, T, N) and np.isnan(out[:, 0]).all() and np.isfinite(out[:, 1:]).all()
    for i in (0, 8, 16):
        for n in (0, 13, T - 2):
            assert np.array_equal(out[:, n + 1, i], orc.pointwise_eval(kind, P[:, i].copy(), inputs[scen[i], :, n]))


def test_ospp_reference_known_answers():
    """ocean_surface_partial_pressure.rs:196-259: delta SST = 4 K, delta DIC = 5 umol/kg give
    339.089 and 381.003 ppm (max_relative 10e-5) for the two parameter sets of the test."""
    K = orc.PW_OSPP
    p1 = orc.pointwise_default_params(K)
    assert orc.pointwise_eval(K, p1, (4.0, 5.0))[0] == pytest.approx(339.089, rel=10e-5)
    names = orc.PW_PARAM_NAMES[K]
    over = dict(ospp_preindustrial=315.0, sensitivity_ospp_to_temperature=0.0423)
    over.update({f"delta_ospp_offsets_{i}": v for i, v in enumerate((1.5, 7.5, 1.3, 2.5, 1.6))})
    over.update({f"delta_ospp_coefficients_{i}": v for i, v in enumerate((-0.02, -0.2, -0.1, -0.14, -0.2))})
    assert set(over) <= set(names)
    assert orc.pointwise_eval(K, orc.pointwise_default_params(K, **over), (4.0, 5.0))[0] == pytest.approx(381.003, rel=10e-5)
    # no anomaly: the pre-industrial pressure
    assert orc.pointwise_eval(K, p1, (0.0, 0.0))[0] == 278.0


def test_fourbox_ocean_heat_uptake_answers():
    """four_box_ocean_heat_uptake.rs:115-260: regional = ERF x ratio; the default ratios average to one."""
    K = orc.PW_FOURBOX_OHU
    p = orc.pointwise_default_params(K)
    assert abs(p.mean() - 1.0) < 0.01
    out = orc.pointwise_eval(K, p, (2.0,))
    assert np.array_equal(out, 2.0 * np.array([1.2, 0.6, 1.6, 0.6])) and out.mean() == pytest.approx(2.0)
