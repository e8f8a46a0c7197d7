"""Shared synthetic inputs: F_syn forcing, the notebook's emissions
scenario, Latin-hypercube two-layer parameter draws.  Pure numpy; no oracle, no product code."""
import numpy as np

TL_RANGES = [(0.8, 1.5), (0.0, 0.1), (1.0, 1.8), (0.5, 1.0), (5.0, 15.0), (50.0, 200.0)]
CC_RANGES = [(15.0, 40.0), (0.0, 0.1)]  # tau, alpha_temperature
SEED = 20260327


def axis_values(start=1750, end=2500):
    """python/rscm/config/builder.py:95: np.arange(start, end + 1) as f64."""
    return np.arange(start, end + 1, dtype=np.float64)


def f_syn(t):
    return 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0)) + 0.3 * np.sin(2.0 * np.pi * (t - 1750.0) / 11.0)


def emissions_syn(t):
    """docs/notebooks/coupled_model.py:397-407 knots, linear between, 1.0 after 2100."""
    years = np.array([1750.0, 1850.0, 1950.0, 2000.0, 2020.0, 2050.0, 2100.0])
    vals = np.array([0.0, 0.5, 3.0, 7.0, 10.0, 5.0, 1.0])
    return np.interp(t, years, vals)


def lhs(n, ranges, seed=SEED):
    """One sample per stratum per dimension, shuffled per dimension -> [P][n] SoA."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((len(ranges), n))
    for j, (lo, hi) in enumerate(ranges):
        u = (np.arange(n) + rng.random(n)) / n
        rng.shuffle(u)
        out[j] = lo + u * (hi - lo)
    return out


def two_layer_params(n, seed=SEED):
    return lhs(n, TL_RANGES, seed)


def coupled_params(n, seed=SEED):
    """[10][n]: two-layer 6, tau, conc_pi, alpha_temperature, erf_2xco2."""
    p = lhs(n, TL_RANGES + CC_RANGES, seed)
    out = np.empty((10, n))
    out[:6] = p[:6]
    out[6] = p[6]
    out[7] = 278.0
    out[8] = p[7]
    out[9] = 3.7
    return out


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    bad = bits(a) != bits(b)
    # any NaN == any NaN for parity purposes (payload bits are not part of the contract)
    bad &= ~(np.isnan(a) & np.isnan(b))
    if bad.any():
        idx = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} differ; first at {tuple(idx)}: "
                             f"{a[tuple(idx)]!r} vs {b[tuple(idx)]!r}")
