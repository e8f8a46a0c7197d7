"""HalocarbonChemistry on the GPU (csrc/halocarbon.hip through the C ABI) against the CPU oracle
(oracle/halocarbon_oracle.c).

Each species step multiplies by exp(-dt/tau) from the device math library (<= 1-2 ulp from
glibc's); the recurrence is a contraction, the aggregates are sums of 41 terms in the reference's
order: |gpu - oracle| <= 1e-12 * max(1, |oracle|)."""
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


def _gpu(ra, bounds, P, E, c0, scen=None, chunks=()):
    with ra.Ensemble(ra.KIND_HALOCARBON, P.shape[1], bounds) as e:
        e.set_params(P)
        e.set_forcing(E, scen)
        for s in range(41):
            e.set_initial(s + 1, c0[s])
        for c in chunks:
            e.run(c)
        e.run()
        assert not e.status().any()
        return np.stack([e.get_series(v) for v in range(1, 46)])


def test_product_species_table_matches_the_oracle(orc):
    from rscm_amd import _lib
    assert _lib.HC_SPECIES == orc.HALO_SPECIES and len(_lib.HC_PARAM_NAMES) == 293
    assert np.array_equal(np.array(_lib.HC_DEFAULTS), orc.halo_default_params())


@pytest.mark.parametrize("n", [1, 63, 1000])
def test_halocarbon_gpu_vs_oracle(ra, orc, n):
    rng = np.random.default_rng(n)
    T = 121
    b = np.concatenate([[1900.0], 1900.0 + np.cumsum(np.where(np.arange(T) % 9 == 4, 0.5, 1.0))])  # uneven steps
    yr = np.arange(T, dtype=float)
    E = np.stack([rng.uniform(0.0, 1.0, (41, 1)) * np.maximum(80.0 - np.abs(yr - 70.0), 0.0),
                  rng.uniform(0.0, 40.0, (41, T))])
    P = np.repeat(orc.halo_default_params().reshape(-1, 1), n, axis=1)
    for sp in ("CFC-11", "CFC-12", "HFC-134a", "SF6", "CH3Br", "Halon-1301"):
        P[orc.halo_index(sp, "lifetime")] *= rng.uniform(0.8, 1.25, n)
        P[orc.halo_index(sp, "radiative_efficiency")] *= rng.uniform(0.9, 1.1, n)
    P[orc.HALO_GLOBALS.index("br_multiplier")] = rng.uniform(45.0, 75.0, n)
    P[orc.HALO_GLOBALS.index("mixing_box_fraction")] = rng.uniform(0.9, 1.0, n)
    c0 = np.array([P[orc.halo_index(s, "concentration_pi"), 0] for s in orc.HALO_SPECIES]) + rng.uniform(0.0, 5.0, 41)
    scen = (np.arange(n) % 2).astype(np.int32)
    want = orc.halo_run(b, P, E, c0, scen=scen, threads=8)
    got = _gpu(ra, b, P, E, c0, scen=scen)
    assert (np.isnan(got) == np.isnan(want)).all() and np.isnan(got[41:, 0]).all()
    ok = ~np.isnan(want)
    err = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
    assert err.max() <= TOL, f"max deviation {err.max():.3e}"
    # the aggregates are exactly what the oracle makes of the GPU's own concentrations (sum order)
    for i in (0, n // 2, n - 1):
        for row in (1, 37, T - 1):
            assert tuple(got[41:, row, i]) == orc.halo_aggregates(P[:, i].copy(), dict(zip(orc.HALO_SPECIES, got[:41, row, i])))
    # launches that split inside a 16-row aggregate chunk give the same bits
    assert np.array_equal(_gpu(ra, b, P, E, c0, scen=scen, chunks=(1, 21, 40)), got, equal_nan=True)


def test_halocarbon_through_the_reference_shaped_front(ra, orc):
    from rscm_amd import _lib, core
    from rscm_amd.magicc import HalocarbonChemistryBuilder
    years = np.arange(1980.0, 2011.0)
    axis = core.TimeAxis.from_bounds(np.append(years, 2011.0))
    T = len(years)
    rng = np.random.default_rng(8)
    E = rng.uniform(0.0, 60.0, (41, T))
    mont = [dict(zip(("name",) + _lib.HC_FIELDS, s)) for s in _lib.HC_MONTREAL]
    mont[0]["lifetime"] = 45.0  # CFC-11
    b = core.ModelBuilder().with_time_axis(axis).with_rust_component(
        HalocarbonChemistryBuilder.from_parameters({"br_multiplier": 65.0, "montreal_gases": mont}).build())
    for k, name in enumerate(_lib.HC_INPUTS):
        b = b.with_exogenous_variable(name, core.Timeseries(E[k], axis, "kt/yr", core.InterpolationStrategy.Previous))
    c0 = {f"Atmospheric Concentration|{s}": float(10 + k) for k, s in enumerate(_lib.HC_SPECIES)}
    m = b.with_initial_values(c0).build()
    m.run()
    res = m.timeseries()
    m.close()
    P = orc.halo_default_params(br_multiplier=65.0, species={"CFC-11.lifetime": 45.0})
    want = orc.halo_run(np.append(years, 2011.0), P, E, np.arange(10.0, 51.0))
    for k, name in enumerate(list(c0) + ["Forcing|Halocarbons", "Forcing|F-gases", "Forcing|Montreal Gases", "EESC"]):
        got, w = res.get_timeseries_by_name(name).values(), want[k, :, 0]
        assert (np.isnan(got) == np.isnan(w)).all() and np.nanmax(np.abs(got - w) / np.maximum(1.0, np.abs(w))) <= TOL, name
    with pytest.raises(NotImplementedError, match="default species set"):
        HalocarbonChemistryBuilder.from_parameters({"fgases": [{"name": "CF4", "lifetime": 1.0}]})


def test_halocarbon_full_size_properties(ra, orc):
    """1e5 members x 751 years (27 GB of series): zero emissions leave pure exponential decay,
    C(t) = C0 exp(-t/tau), for every member and species to rounding; total forcing equals F-gas plus
    Montreal forcing; sampled members match the oracle."""
    n, T = 100_000, 751
    rng = np.random.default_rng(6)
    b = np.arange(T + 1, dtype=float) + 1750.0
    P = np.repeat(orc.halo_default_params().reshape(-1, 1), n, axis=1)
    tau_row = orc.halo_index("HCFC-22", "lifetime")
    P[tau_row] = rng.uniform(8.0, 16.0, n)
    c0 = np.full(41, 100.0)
    with ra.Ensemble(ra.KIND_HALOCARBON, n, b) as e:
        e.set_params(P)
        e.set_forcing(np.zeros((41, T)))
        for s in range(41):
            e.set_initial(s + 1, c0[s])
        e.run()
        ms = e.last_run_ms()
        v = orc.HALO_SPECIES.index("HCFC-22") + 1
        c40 = e.get_series(v, 40, 41)[0]
        tot, fg, mo = (e.get_series(k, 300, 301)[0] for k in (42, 43, 44))
        cf4 = e.get_series(1, T - 1, T)[0]
    assert np.abs(c40 - 100.0 * np.exp(-40.0 / P[tau_row])).max() < 1e-10
    assert np.abs(tot - (fg + mo)).max() < 1e-13
    assert np.abs(cf4 - 100.0 * np.exp(-750.0 / 50000.0)).max() < 1e-10
    print(f"halocarbon 1e5 x 750 yr: {ms:.1f} ms")
