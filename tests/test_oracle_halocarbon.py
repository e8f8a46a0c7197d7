"""The HalocarbonChemistry oracle (oracle/halocarbon_oracle.c) against the known answers of the
reference's in-file unit tests (crates/rscm-magicc/src/chemistry/halocarbon.rs:352-741,
parameters/halocarbon.rs:162-391).  The reference holds no golden vectors for this component."""
import numpy as np
import pytest

from oracle import cbind as orc


def test_halocarbon_parameter_table():
    p = orc.halo_default_params()
    L = orc.lib()
    assert L.orc_halo_n_fgases() == 23 and L.orc_halo_n_species() == 41 and len(p) == 6 + 41 * 7
    assert p[orc.halo_index("CF4", "lifetime")] == 50000.0 and p[orc.halo_index("HFC-152a", "lifetime")] == 1.6
    assert p[orc.halo_index("CFC-11", "n_cl")] == 3 and p[orc.halo_index("CFC-11", "fractional_release")] == 0.47
    assert p[orc.halo_index("CH3Cl", "concentration_pi")] == 500.0 and p[orc.halo_index("CH3Br", "concentration_pi")] == 5.0
    assert p[orc.halo_index("Halon-1202", "n_br")] == 2 and p[orc.halo_index("Halon-1202", "molecular_weight")] == 209.8
    # only Montreal gases release halogens
    assert all(p[orc.halo_index(s, "fractional_release")] == 0.0 for s in orc.HALO_SPECIES[:23])


def test_halocarbon_unit_test_answers():
    p = orc.halo_default_params()
    assert orc.halo_decay_species(p, "CF4", 100.0, 0.0, 1.0) == pytest.approx(100.0 * np.exp(-1.0 / 50000.0), abs=1e-10)
    assert orc.halo_decay_species(p, "HFC-152a", 100.0, 0.0, 1.0) == pytest.approx(100.0 * np.exp(-1.0 / 1.6), abs=1e-8)
    # equilibrium C = E conv tau after 500 years (HFC-134a, halocarbon.rs:426-456)
    conv = (28.97 / 102.0) * (1e9 / (5.133e9 * 1e12)) * 1e12 / 0.949
    c = 0.0
    for _ in range(500):
        c = orc.halo_decay_species(p, "HFC-134a", c, 100.0, 1.0)
    assert abs(c - 100.0 * conv * 14.0) / (100.0 * conv * 14.0) < 0.01
    c = 250.0
    for _ in range(520):
        c = orc.halo_decay_species(p, "CFC-11", c, 0.0, 1.0)
    assert c < 0.02  # ten lifetimes
    # forcing: zero at pre-industrial, linear in the excess, total = F-gases + Montreal
    assert orc.halo_aggregates(p, {})[:3] == (0.0, 0.0, 0.0)
    t1 = orc.halo_aggregates(p, {"CFC-12": 100.0})[0]
    assert t1 == pytest.approx(100.0 * 0.364 / 1000.0, rel=1e-14) and orc.halo_aggregates(p, {"CFC-12": 200.0})[0] == pytest.approx(2 * t1, rel=1e-14)
    tot, fg, mo, _ = orc.halo_aggregates(p, {"CFC-11": 230.0, "CFC-12": 520.0, "HFC-134a": 100.0, "SF6": 10.0})
    assert abs(tot - (fg + mo)) < 1e-10 and tot > 0 and fg == pytest.approx((100.0 * 0.16 + 10.0 * 0.57) / 1000.0, rel=1e-14)
    # EESC: chlorine count x normalised release; bromine weighted 60x (halocarbon.rs:534-600)
    assert orc.halo_aggregates(p, {"CFC-11": 200.0, "CH3Cl": 0.0, "CH3Br": 0.0})[3] == pytest.approx(600.0, abs=1e-6)
    assert orc.halo_aggregates(p, {"Halon-1301": 3.0, "CH3Cl": 0.0, "CH3Br": 0.0})[3] == pytest.approx(3.0 * 60.0 * 0.28 / 0.47, abs=1e-6)
    assert orc.halo_aggregates(p, {"SF6": 50.0, "HFC-23": 30.0, "CH3Cl": 0.0, "CH3Br": 0.0})[3] == 0.0  # F-gases carry no Cl/Br
    # the natural background: CH3Cl and CH3Br at their pre-industrial levels
    assert orc.halo_aggregates(p, {})[3] == pytest.approx(500.0 * 1 * 0.44 / 0.47 + 5.0 * 60.0 * 0.60 / 0.47, rel=1e-14)


def test_halocarbon_run_layout():
    T, N = 15, 6
    rng = np.random.default_rng(4)
    b = np.concatenate([[2000.0], 2000.0 + np.cumsum(rng.uniform(0.5, 1.5, T))])
    E = rng.uniform(0.0, 50.0, (2, 41, T))
    P = np.repeat(orc.halo_default_params().reshape(-1, 1), N, axis=1)
    P[orc.halo_index("CFC-11", "lifetime")] = rng.uniform(45.0, 60.0, N)
    c0 = rng.uniform(0.0, 100.0, 41)
    scen = (np.arange(N) % 2).astype(np.int32)
    out = orc.halo_run(b, P, E, c0, scen=scen, threads=2)
    assert out.shape == (45, T, N) and np.isnan(out[41:, 0]).all() and (out[:41, 0] == c0[:, None]).all()
    for i in (0, 3, 5):
        p = P[:, i].copy()
        c = c0.copy()
        for n in range(T - 1):
            c = np.array([orc.halo_decay_species(p, s, c[k], E[scen[i], k, n], b[n + 1] - b[n]) for k, s in enumerate(orc.HALO_SPECIES)])
            assert np.array_equal(out[:41, n + 1, i], c)
            assert tuple(out[41:, n + 1, i]) == orc.halo_aggregates(p, dict(zip(orc.HALO_SPECIES, c)))
