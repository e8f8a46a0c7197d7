"""GPU tier (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties.

Parity bars (stated here, as the task requires):
  * two-layer kind, RSCM_MODE_EXACT: BIT-EXACT vs the oracle (same f64 expression order, no FMA
    contraction, IEEE division), including non-finite members.
  * two-layer kind, RSCM_MODE_FAST: |gpu - oracle| <= 1e-11 * max(1, |oracle|) on members whose
    trajectory stays bounded (finite to the end and |Ts| < 50 K; measured worst case 8e-13 on
    20000 members, scripts/fast_mode_error.py).  Members in runaway feedback (lambda0 - a*Ts < 0)
    approach a finite-time singularity where any rounding difference is amplified without
    bound: up to the year |Ts| passes 50 K they are held to 1e-9, the year itself to +-1, and the status flag
    (test_fast_mode_on_the_headline_draw_including_its_runaway_members: all 1e5 members of bench.py's draw).
  * coupled kind: exp/log come from the device math library (<= 1 ulp from glibc), so
    |gpu - oracle| <= 1e-11 * max(1, |oracle|) on bounded members in RSCM_MODE_EXACT (the reference's
    expression order) and in RSCM_MODE_FAST (closed-form RK4 step of the linear carbon box, folded heat
    capacities, FMAs; measured 3e-13) alike -- every coupled test below runs in both.  Cumulative
    emissions stay bit-exact in either mode.
  * which kinds have a FAST arithmetic at all: two-layer, coupled, CarbonCycle, ClimateUDEB, OceanCarbon
    (include/rscm_gpu.h, RSCM_MODE_FAST); for every other kind the mode is accepted and changes nothing.
  * integer/index work (time indexing, scenario selection, status flags, LHS strata): exact.
"""
import os

import numpy as np
import pytest

from tests.helpers import (CC_RANGES, SEED, TL_RANGES, assert_bit_equal, axis_values,
                           coupled_params, emissions_syn, f_syn, two_layer_params)

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FAST_RTOL = 1e-11


@pytest.fixture(scope="module")
def ra():
    import rscm_amd
    from rscm_amd import _lib
    _lib.load()  # fails loudly when the HIP extension is missing
    assert _lib.device_count() >= 1, "no HIP device visible"
    return rscm_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import cbind
    return cbind


def _tl_gpu(ra, t, P, F, ts0, td0, *, scen=None, source=0, mode=0, h=None, chunks=None):
    b = np.append(t, t[-1] + (t[-1] - t[-2]))
    with ra.Ensemble(ra.KIND_TWO_LAYER, P.shape[1], b) as e:
        e.set_mode(mode)
        if h is not None:
            e.set_step_size(0, h)
        e.set_params(P)
        e.set_forcing(F, scen, source)
        e.set_initial("Surface Temperature", ts0)
        e.set_initial("Deep Ocean Temperature", td0)
        if chunks:
            for c in chunks:
                e.run(c)
        e.run()
        assert e.finished()
        return (e.get_series("Surface Temperature"), e.get_series("Deep Ocean Temperature"),
                e.status())


def _bounded(ts):
    """Members that stay finite to the end and below 50 K (see the module docstring)."""
    with np.errstate(all="ignore"):
        return np.isfinite(ts[-1]) & (np.nanmax(np.abs(ts), axis=0) < 50.0)


def _close(a, b, rtol):
    return np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b))


def _held_until_runaway(ts_got, ts_want, pairs, rtol=1e-9, min_rows=100):
    """Members in runaway feedback ([T][members], none of them bounded): up to the row where |Ts| passes 50 K -- on either side; the two
    rows may differ by one -- every (got, want) pair agrees within rtol; the prefix is most of the run (these members blow up late)."""
    with np.errstate(all="ignore"):
        big_want = ~(np.abs(ts_want) < 50.0)          # (NaN / inf count as past the bar)
        big_got = ~(np.abs(ts_got) < 50.0)
    T = ts_want.shape[0]
    first_want = np.where(big_want.any(axis=0), big_want.argmax(axis=0), T)
    first_got = np.where(big_got.any(axis=0), big_got.argmax(axis=0), T)
    assert (np.abs(first_want - first_got) <= 1).all()
    before = np.arange(T)[:, None] < np.minimum(first_want, first_got)[None, :]
    with np.errstate(all="ignore"):
        for got, want in pairs:
            assert (_close(got, want, rtol) | ~before).all()
    assert before.sum() > min_rows * ts_want.shape[1]


# ------------------------------------------------------------------------------ division
def test_hoisted_reciprocal_division_is_ieee(ra):
    from rscm_amd.ensemble import selftest_div
    rng = np.random.default_rng(7)
    n = 1 << 21
    num = rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))
    den = rng.uniform(0.5, 300.0, n) * rng.choice([-1.0, 1.0], n)
    # edge cases: zeros, denormals, huge, inf, nan, divisors outside the window
    edge_n = np.array([0.0, -0.0, 5e-324, 1e-310, 1e-300, 2.0 ** -767, 2.0 ** -768, 2.0 ** 511,
                       2.0 ** 512, 1e300, 1.7e308, np.inf, -np.inf, np.nan, 1.0, 3.0])
    edge_d = np.array([8.0, 100.0, 1e-200, 1e200, 0.0, -0.0, np.inf, np.nan, 5e-324, 2.0 ** 128,
                       2.0 ** 129, 2.0 ** -128, 2.0 ** -129, 3.0, 2.13, 1.0])
    en, ed = np.meshgrid(edge_n, edge_d)
    num = np.concatenate([num, en.ravel()])
    den = np.concatenate([den, ed.ravel()])
    # wide-exponent numerators against typical heat capacities
    wide = np.ldexp(rng.uniform(1, 2, 1 << 16), rng.integers(-1074, 1023, 1 << 16))
    num = np.concatenate([num, wide])
    den = np.concatenate([den, rng.uniform(5.0, 200.0, wide.size)])
    # operands hugging the window edges: |n| near 2^-511 and 2^513, |d| near 2^-128 and 2^129
    edge = np.ldexp(rng.uniform(1, 2, 1 << 16), rng.choice([-512, -511, -510, 511, 512, 513], 1 << 16))
    num = np.concatenate([num, edge])
    den = np.concatenate([den, np.ldexp(rng.uniform(1, 2, edge.size),
                                        rng.choice([-129, -128, -127, 127, 128, 129], edge.size))])
    ref, fast, used = selftest_div(num, den)
    u = used.astype(bool)
    with np.errstate(all="ignore"):
        assert_bit_equal(ref, num / den, "device IEEE division vs host IEEE division")
    # inside the windows the three-instruction quotient IS the IEEE quotient
    assert_bit_equal(fast[u], ref[u], "hoisted-reciprocal quotient vs IEEE division")
    assert u[: n].mean() > 0.99  # typical operands are inside
    assert u[n: n + en.size].mean() < 0.5  # the edge-case grid mostly is not
    assert 0.2 < u[-edge.size:].mean() < 0.8  # both sides of the window edges are exercised
    # the windows are what the kernels assume: |n| in [2^-511, 2^513), |d| in [2^-128, 2^129)
    with np.errstate(all="ignore"):
        en_, ed_ = np.frexp(np.abs(num))[1] - 1, np.frexp(np.abs(den))[1] - 1
    inside = (np.isfinite(num) & np.isfinite(den) & (np.abs(num) >= 2.0 ** -511) &
              (np.abs(num) < 2.0 ** 513) & (np.abs(den) >= 2.0 ** -128) & (np.abs(den) < 2.0 ** 129))
    assert (u == inside).all()


# ------------------------------------------------------------------------------ two-layer exact
@pytest.mark.parametrize("n_members", [1, 63, 257, 1000])
@pytest.mark.parametrize("source", [0, 1])
def test_two_layer_exact_bitwise_vs_oracle(ra, orc, n_members, source):
    t = axis_values()
    P = two_layer_params(n_members, seed=SEED + n_members)
    F = f_syn(t)
    want = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.0, 0.0, source=source, threads=8)
    ts, td, st = _tl_gpu(ra, t, P, F, 0.0, 0.0, source=source)
    assert_bit_equal(ts, want[0], "Ts")
    assert_bit_equal(td, want[1], "Td")
    bad = ~(np.isfinite(want[0][-1]) & np.isfinite(want[1][-1]))
    assert (st.astype(bool) == bad).all()


@pytest.mark.parametrize("n_points", [2, 3])
@pytest.mark.parametrize("n_members", [1, 65])
def test_smallest_shapes(ra, orc, n_points, n_members):
    """The shortest axis TimeAxis::from_values accepts (two points: ONE step, runtime.rs:524 runs len-1 of them)
    and three points, one member and one member more than a wavefront: bit-exact against the oracle; a run over
    an empty range is a no-op; a likelihood over no observations is the empty sum."""
    t = axis_values(1750, 1750 + n_points - 1)
    P = two_layer_params(n_members, seed=SEED + 7 * n_points + n_members)
    F = f_syn(t)
    b = orc.bounds_from_values(t)
    for source in (0, 1):
        want = orc.two_layer_run(b, P, F, 0.25, -0.5, source=source, threads=1)
        with ra.Ensemble(ra.KIND_TWO_LAYER, n_members, b) as e:
            e.set_params(P)
            e.set_forcing(F, None, source)
            e.set_initial("Surface Temperature", 0.25)
            e.set_initial("Deep Ocean Temperature", -0.5)
            e.run(0)                       # [0, 0): nothing to do
            assert e.time_index == 0
            e.run()
            assert e.finished() and e.time_index == n_points - 1
            assert_bit_equal(e.get_series("Surface Temperature"), want[0], f"Ts, {n_points} points, source {source}")
            assert_bit_equal(e.get_series("Deep Ocean Temperature"), want[1], f"Td, {n_points} points, source {source}")
            assert not e.status().any()
            ll = e.loglik(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), np.zeros(0))
            assert ll.shape == (n_members,) and (ll == 0.0).all()


def test_two_layer_golden_fixture(ra):
    g = np.load(os.path.join(GOLDEN, "two_layer_golden.npz"))
    for source in (0, 1):
        ts, td, _ = _tl_gpu(ra, g["time_values"], g["params"], g["forcing"], g["ts0"], g["td0"],
                            scen=g["scen"], source=source)
        assert_bit_equal(ts, g[f"ts_src{source}"], f"golden Ts src{source}")
        assert_bit_equal(td, g[f"td_src{source}"], f"golden Td src{source}")


def test_two_layer_scenarios_and_member_initials(ra, orc):
    t = axis_values(1750, 2100)  # config 1 axis: 351 points
    n = 513
    rng = np.random.default_rng(3)
    P = two_layer_params(n)
    F = np.stack([f_syn(t) * s for s in (1.0, 0.5, -0.25, 2.0, 0.0)])
    scen = rng.integers(0, 5, n).astype(np.int32)
    ts0, td0 = rng.normal(0, 0.5, n), rng.normal(0, 0.2, n)
    want = orc.two_layer_run(orc.bounds_from_values(t), P, F, ts0, td0, scen=scen, threads=8)
    ts, td, _ = _tl_gpu(ra, t, P, F, ts0, td0, scen=scen)
    assert_bit_equal(ts, want[0])
    assert_bit_equal(td, want[1])


def test_two_layer_many_scenarios_bypass_lds(ra, orc):
    """S*T*8 B beyond the LDS budget: forcing is read through L2 instead; same bits."""
    t = axis_values()
    n, S = 300, 40  # 40 * 750 * 8 = 240 KB > 160 KB
    rng = np.random.default_rng(5)
    P = two_layer_params(n)
    F = np.stack([f_syn(t) * rng.uniform(0.2, 1.2) for _ in range(S)])
    scen = rng.integers(0, S, n).astype(np.int32)
    want = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.0, 0.0, scen=scen, threads=8)
    ts, td, _ = _tl_gpu(ra, t, P, F, 0.0, 0.0, scen=scen)
    assert_bit_equal(ts, want[0])
    assert_bit_equal(td, want[1])


def test_two_layer_nonfinite_and_extreme_members(ra, orc):
    """Runaway feedback (a large) overflows to inf/NaN; tiny and huge forcings leave the fast
    division window; zero forcing keeps exact zeros.  All bit-identical, failures flagged."""
    t = axis_values(1750, 2000)
    base = np.array([1.0, 0.0, 1.0, 0.7, 8.0, 100.0])
    rows = []
    for a in (0.0, 0.05, 0.3, 1.0, 5.0):
        p = base.copy()
        p[1] = a
        rows.append(p)
    for cs, cd in ((1e-3, 1e3), (2.0 ** -130, 1.0), (1.0, 2.0 ** 130), (0.0, 100.0), (8.0, np.inf),
                   (-8.0, 100.0), (np.nan, 100.0)):
        p = base.copy()
        p[4], p[5] = cs, cd
        rows.append(p)
    P = np.array(rows).T.copy()
    n = P.shape[1]
    F = np.stack([f_syn(t), np.zeros_like(t), f_syn(t) * 1e-300, f_syn(t) * 1e250,
                  f_syn(t) * 5e-324])
    with np.errstate(all="ignore"):
        for s in range(F.shape[0]):
            scen = np.full(n, s, np.int32)
            want = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.0, 0.0, scen=scen)
            ts, td, st = _tl_gpu(ra, t, P, F, 0.0, 0.0, scen=scen)
            assert_bit_equal(ts, want[0], f"Ts scenario {s}")
            assert_bit_equal(td, want[1], f"Td scenario {s}")
            bad = ~(np.isfinite(want[0][-1]) & np.isfinite(want[1][-1]))
            assert (st.astype(bool) == bad).all()


def test_two_layer_irregular_axis_and_step_size(ra, orc):
    """Sub-step count ceil((t1-t0)/h) varies per model step; h = 1/120 as in coupled_models.rs."""
    t = np.concatenate([np.arange(1750.0, 1760.0, 0.5), np.arange(1760.0, 1800.0, 1.0),
                        np.arange(1800.0, 1900.0, 5.0)])
    P = two_layer_params(130)
    F = f_syn(t)
    for h in (0.1, 1.0 / 120.0, 0.25):
        want = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.1, 0.0, h=h, threads=8)
        ts, td, _ = _tl_gpu(ra, t, P, F, 0.1, 0.0, h=h)
        assert_bit_equal(ts, want[0], f"h={h}")
        assert_bit_equal(td, want[1], f"h={h}")


def test_step_and_resume_equal_one_run(ra):
    t = axis_values(1750, 1850)
    P, F = two_layer_params(200), f_syn(t)
    one = _tl_gpu(ra, t, P, F, 0.0, 0.0)
    chunked = _tl_gpu(ra, t, P, F, 0.0, 0.0, chunks=[1, 2, 37, 38, 99])
    assert_bit_equal(one[0], chunked[0])
    assert_bit_equal(one[1], chunked[1])
    b = np.append(t, t[-1] + 1.0)
    with ra.Ensemble(ra.KIND_TWO_LAYER, 200, b) as e:
        e.set_params(P)
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.step()
        e.step()
        assert e.time_index == 2
        part = e.get_series(1)
        assert_bit_equal(part[:3], one[0][:3])
        assert np.isnan(part[3:]).all()  # not yet computed by this model instance
        e.run()
        assert_bit_equal(e.get_series(1), one[0])
        # strided / member-window extraction
        sub = e.get_series(2, 10, 90, 7, 13, 101)
        assert_bit_equal(sub, one[1][10:90:7, 13:101])


# ------------------------------------------------------------------------------ two-layer fast
def test_two_layer_fast_mode_tolerance(ra, orc):
    t = axis_values()
    P = two_layer_params(2000)
    F = f_syn(t)
    want = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.0, 0.0, threads=8)
    ts, td, st = _tl_gpu(ra, t, P, F, 0.0, 0.0, mode=1)
    bounded = _bounded(want[0])
    assert bounded.mean() > 0.5
    assert _close(ts[:, bounded], want[0][:, bounded], FAST_RTOL).all()
    assert _close(td[:, bounded], want[1][:, bounded], FAST_RTOL).all()
    failed = ~(np.isfinite(want[0][-1]) & np.isfinite(want[1][-1]))
    assert (st.astype(bool)[failed]).all()


def test_fast_mode_on_the_headline_draw_including_its_runaway_members(ra, orc):
    """bench.py's own workload -- 1e5 members of the seeded Latin hypercube, 750 years -- in RSCM_MODE_FAST, EVERY member checked:
    3.4 % of this draw run away (lambda0 - a Ts turns negative and Ts heads for a finite-time singularity).  Bounded members: the
    stated 1e-11.  Runaway members are not left to their status flag: up to the year the oracle's |Ts| passes 50 K the two
    trajectories agree within 1e-9 (the feedback amplifies a rounding difference e-fold per ~0.5 years by then, hence the wider
    bar), the year they pass 50 K is the same give or take one, and both sides flag the member failed at the end."""
    t = axis_values()
    b = np.append(t, t[-1] + 1.0)
    F = f_syn(t)
    n = 100_000
    lo = np.array([r[0] for r in TL_RANGES])
    hi = np.array([r[1] for r in TL_RANGES])
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_mode(1)
        e.sample_lhs(SEED, lo, hi)
        P = e.get_params()
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run()
        ts, td, st = e.get_series(1), e.get_series(2), e.status().astype(bool)
    want_ts, want_td = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.0, 0.0, threads=8)
    bounded = _bounded(want_ts)
    assert 0.95 < bounded.mean() < 0.98
    assert _close(ts[:, bounded], want_ts[:, bounded], FAST_RTOL).all() and _close(td[:, bounded], want_td[:, bounded], FAST_RTOL).all()
    assert not st[bounded].any()
    away = np.flatnonzero(~bounded)
    _held_until_runaway(ts[:, away], want_ts[:, away], [(ts[:, away], want_ts[:, away]), (td[:, away], want_td[:, away])])
    failed = ~(np.isfinite(want_ts[-1]) & np.isfinite(want_td[-1]))
    # (a member that overflows in the very last year could do so a year apart on the two sides: none does in this draw, one in a thousand is allowed)
    assert (st[away] == failed[away]).mean() >= 0.999 and failed[away].mean() > 0.9


# ------------------------------------------------------------------------------ coupled chain
def _cp_gpu(ra, t, P, E, init, scen=None, mode=0):
    b = np.append(t, t[-1] + (t[-1] - t[-2]))
    with ra.Ensemble(ra.KIND_COUPLED, P.shape[1], b) as e:
        e.set_mode(mode)
        e.set_params(P)
        e.set_forcing(E, scen)
        for k, v in init.items():
            e.set_initial(k, v)
        e.run()
        return {k: e.get_series(k) for k in e.var_ids if e.var_ids[k] > 0}, e.status()


CP_NAMES = {"ts": "Surface Temperature", "td": "Deep Ocean Temperature",
            "conc": "Atmospheric Concentration|CO2", "cum_uptake": "Cumulative Land Uptake",
            "cum_emis": "Cumulative Emissions|CO2", "erf_co2": "Effective Radiative Forcing|CO2",
            "erf_total": "Effective Radiative Forcing"}
CP_INIT = {"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0,
           "Atmospheric Concentration|CO2": 278.0, "Cumulative Land Uptake": 0.0,
           "Cumulative Emissions|CO2": 0.0}


@pytest.mark.parametrize("mode", [0, 1])
def test_coupled_vs_oracle(ra, orc, mode):
    t = axis_values()
    P = coupled_params(777)
    E = emissions_syn(t)
    want = orc.coupled_run(orc.bounds_from_values(t), P, E,
                           dict(ts=0.0, td=0.0, conc=278.0, cum_uptake=0.0, cum_emis=0.0),
                           threads=8)
    got, st = _cp_gpu(ra, t, P, E, CP_INIT, mode=mode)
    bounded = _bounded(want["ts"])
    assert bounded.mean() > 0.5
    for k, name in CP_NAMES.items():
        g, w = got[name], want[k]
        # index-0 NaN of pure outputs (no initial value) is part of the contract
        assert (np.isnan(g[0]) == np.isnan(w[0])).all(), name
        assert _close(g[1:, bounded], w[1:, bounded], FAST_RTOL).all(), name
    # cumulative emissions involve no transcendental: bit-exact
    assert_bit_equal(got["Cumulative Emissions|CO2"], want["cum_emis"])


@pytest.mark.parametrize("mode", [0, 1])
def test_coupled_golden_fixture(ra, mode):
    g = np.load(os.path.join(GOLDEN, "coupled_golden.npz"))
    got, _ = _cp_gpu(ra, g["time_values"], g["params"], g["emissions"], CP_INIT, mode=mode)
    for k, name in CP_NAMES.items():
        assert _close(got[name][1:], g[k][1:], FAST_RTOL).all(), name


# ------------------------------------------------------------------------------ ensemble ops
def test_loglik_summary_status(ra, orc):
    t = axis_values(1750, 2100)
    n = 1500
    P = two_layer_params(n)
    F = f_syn(t)
    b = np.append(t, t[-1] + 1.0)
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_params(P)
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run()
        ts, td = e.get_series(1), e.get_series(2)
        tidx = np.arange(100, 271, 10, dtype=np.int32)  # 1850..2020 step 10
        obs = ts[tidx, 0] + 0.05
        sig = np.full(len(tidx), 0.1)
        for normalize in (False, True):
            got = e.loglik(np.ones(len(tidx), int), tidx, obs, sig, normalize)
            want = orc.gaussian_loglik([ts], np.zeros(len(tidx), np.int32), tidx, obs, sig,
                                       normalize, threads=4)
            fin = np.isfinite(want)
            assert (np.isfinite(got) == fin).all()
            assert np.allclose(got[fin], want[fin], rtol=1e-13, atol=0)
            assert (got[~fin] == -np.inf).all()
        # two variables: per-variable partial sums, then the total
        ov = np.r_[np.ones(3, int), np.full(2, 2)]
        ot = np.array([10, 20, 30, 40, 50], np.int32)
        val = np.r_[ts[[10, 20, 30], 1], td[[40, 50], 1]] + 0.01
        got = e.loglik(ov, ot, val, np.full(5, 0.2))
        want = orc.gaussian_loglik([ts, td], (ov - 1).astype(np.int32), ot, val, np.full(5, 0.2))
        fin = np.isfinite(want)
        assert np.allclose(got[fin], want[fin], rtol=1e-13, atol=0)
        s = e.summary(1, 200)
        row = ts[200]
        ok = np.isfinite(row)
        assert s["count"] == ok.sum()
        assert np.isclose(s["mean"], row[ok].mean(), rtol=1e-12)
        assert s["min"] == row[ok].min() and s["max"] == row[ok].max()
        assert (e.status().astype(bool) == ~(np.isfinite(ts[-1]) & np.isfinite(td[-1]))).all()


def test_device_lhs_is_a_latin_hypercube(ra):
    """One sample per stratum per dimension (the property parameter_set.rs:207-233 guarantees),
    and rank-sharded generation equals single-device generation."""
    t = axis_values(1750, 1760)
    b = np.append(t, t[-1] + 1.0)
    n = 4099
    lo = np.array([r[0] for r in TL_RANGES])
    hi = np.array([r[1] for r in TL_RANGES])
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.sample_lhs(SEED, lo, hi)
        P = e.get_params()
    u = (P - lo[:, None]) / (hi - lo)[:, None]
    assert ((u >= 0) & (u < 1)).all()
    strata = np.floor(u * n).astype(np.int64)
    for j in range(6):
        assert np.array_equal(np.sort(strata[j]), np.arange(n)), f"dimension {j}"
    # dimensions are shuffled independently
    assert abs(np.corrcoef(u[0], u[1])[0, 1]) < 0.06
    assert not np.array_equal(strata[0], strata[1])
    # two ranks owning [0, k) and [k, n) reproduce the same global matrix, no communication
    k = 1500
    with ra.Ensemble(ra.KIND_TWO_LAYER, k, b) as e0, ra.Ensemble(ra.KIND_TWO_LAYER, n - k, b) as e1:
        e0.sample_lhs(SEED, lo, hi, 0, n)
        e1.sample_lhs(SEED, lo, hi, k, n)
        assert_bit_equal(np.concatenate([e0.get_params(), e1.get_params()], axis=1), P)
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.sample_lhs(SEED + 1, lo, hi)
        assert not np.array_equal(e.get_params(), P)


# ------------------------------------------------------------------------------ error behaviour
def test_error_conventions(ra):
    from rscm_amd import RscmGpuError
    t = axis_values(1750, 1760)
    b = np.append(t, t[-1] + 1.0)
    with ra.Ensemble(ra.KIND_TWO_LAYER, 10, b) as e:
        with pytest.raises(RscmGpuError) as ei:  # nothing configured
            e.run()
        assert ei.value.code == 2
        e.set_params(two_layer_params(10))
        e.set_forcing(f_syn(t))
        e.set_initial(1, 0.0)
        with pytest.raises(RscmGpuError, match="MissingInitialValue"):
            e.run()
        e.set_initial(2, 0.0)
        with pytest.raises(ValueError):
            e.set_params(np.zeros((5, 10)))
        with pytest.raises(ValueError, match="Expected 6 parameters"):
            e.set_params_aos(np.zeros((10, 5)))
        with pytest.raises(RscmGpuError):  # scenario index out of range
            e.set_forcing(f_syn(t), np.full(10, 3, np.int32))
        e.run()
        with pytest.raises(RscmGpuError) as ei:  # Model::step asserts time_index < len-1
            e.step()
        assert ei.value.code == 2
    # sub-annual axis (1/16 yr) with the hard-coded h = 0.1: one RK4 step overshoots the end of
    # the model step by 0.0375 yr >= 5e-3 -- the reference panics in get_last_step (ivp/mod.rs:97)
    tm = 1750.0 + np.arange(25) / 16.0
    with ra.Ensemble(ra.KIND_TWO_LAYER, 4, np.append(tm, tm[-1] + 1 / 16.0)) as e:
        e.set_params(two_layer_params(4))
        e.set_forcing(np.ones(25))
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        with pytest.raises(RscmGpuError) as ei:
            e.run()
        assert ei.value.code == 3
        assert e.time_index == 0  # nothing was launched
        e.set_step_size(0, 1.0 / 64.0)
        e.run()
        assert e.finished()
    with pytest.raises(RscmGpuError):
        ra.Ensemble(ra.KIND_TWO_LAYER, 4, [3.0, 2.0, 1.0])


# ------------------------------------------------------------------------------ full size
@pytest.mark.parametrize("n_members", [100_000, 1_000_000])
def test_full_size_properties(ra, orc, n_members):
    """BASELINE.json sizes (1e5 and 1e6 members x 751 points), checked through properties that do
    not need the oracle on every member:
      * determinism: two runs give identical bits;
      * a = 0 members are linear and doubling F is exact in binary64, so Ts(2F) == 2*Ts(F) bitwise;
      * permutation equivariance: reversing the member order reverses the outputs;
      * oracle spot check on 512 random members, bit-exact.
    """
    t = axis_values()
    b = np.append(t, t[-1] + 1.0)
    F = f_syn(t)
    lo = np.array([r[0] for r in TL_RANGES])
    hi = np.array([r[1] for r in TL_RANGES])
    with ra.Ensemble(ra.KIND_TWO_LAYER, n_members, b) as e:
        e.sample_lhs(SEED, lo, hi)
        P = e.get_params()
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run()
        rows = [1, 100, 375, 750]
        first = {r: e.get_series(1, r, r + 1)[0] for r in rows}
        last_td = e.get_series(2, 750, 751)[0]
        rng = np.random.default_rng(11)
        pick = np.sort(rng.choice(n_members, 512, replace=False))
        want = orc.two_layer_run(orc.bounds_from_values(t), np.ascontiguousarray(P[:, pick]), F,
                                 0.0, 0.0, threads=8)
        for i, m in enumerate(pick[:64]):
            got = e.get_series(1, 0, 751, 1, int(m), int(m) + 1)[:, 0]
            assert_bit_equal(got, want[0][:, i], f"member {m}")
        for r in rows:
            assert_bit_equal(first[r][pick], want[0][r], f"row {r}")
        assert_bit_equal(last_td[pick], want[1][750])
        # determinism
        e.rewind()
        e.run()
        for r in rows:
            assert_bit_equal(e.get_series(1, r, r + 1)[0], first[r], "second run")
        # permutation equivariance
        e.set_params(np.ascontiguousarray(P[:, ::-1]))
        e.rewind()
        e.run()
        assert_bit_equal(e.get_series(1, 750, 751)[0][::-1], first[750], "reversed members")
        # exact linearity in F for a = 0
        P0 = P.copy()
        P0[1] = 0.0
        e.set_params(P0)
        e.rewind()
        e.run()
        base = e.get_series(1, 750, 751)[0]
        e.set_forcing(2.0 * F)
        e.rewind()
        e.run()
        assert_bit_equal(e.get_series(1, 750, 751)[0], 2.0 * base, "Ts(2F) == 2 Ts(F)")
        assert np.isfinite(base).all()


@pytest.mark.parametrize("mode", [0, 1])
def test_coupled_full_size_properties(ra, orc, mode):
    """BASELINE.json configs[2]: the coupled chain CarbonCycle -> CO2ERF -> Sum -> TwoLayer at
    1e6 members x 751 points (42 GB of series on the device), through properties plus an oracle
    spot check:
      * determinism: a second run gives identical bits;
      * sub-ensemble equality: members 0..776 equal a 777-member run with the same parameters, bit for
        bit (nothing depends on the ensemble size or on the position inside a wavefront / workgroup);
      * 256 random members against the CPU oracle at 1e-11 on all seven series (bounded members);
      * `Cumulative Emissions|CO2` involves no transcendental: bit-exact against the oracle."""
    n = 1_000_000
    t = axis_values()
    b = np.append(t, t[-1] + 1.0)
    P = coupled_params(n)
    E = emissions_syn(t)
    rows = [1, 250, 500, 750]
    rng = np.random.default_rng(23)
    pick = np.sort(rng.choice(n, 256, replace=False))
    with ra.Ensemble(ra.KIND_COUPLED, n, b) as e:
        e.set_mode(mode)
        e.set_params(P)
        e.set_forcing(E)
        for k, v in CP_INIT.items():
            e.set_initial(k, v)
        e.run()
        first = {(name, r): e.get_series(name, r, r + 1)[0] for name in CP_NAMES.values() for r in rows}
        picked = {name: np.stack([e.get_series(name, 0, 751, 1, int(m), int(m) + 1)[:, 0] for m in pick[:32]], axis=1)
                  for name in CP_NAMES.values()}
        head = {name: e.get_series(name, 0, 751, 1, 0, 777) for name in CP_NAMES.values()}
        status = e.status()
        e.rewind()
        e.run()
        for (name, r), want_row in first.items():
            assert_bit_equal(e.get_series(name, r, r + 1)[0], want_row, f"second run {name} row {r}")
    assert status.mean() < 0.5
    small, _ = _cp_gpu(ra, t, np.ascontiguousarray(P[:, :777]), E, CP_INIT, mode=mode)
    for name in CP_NAMES.values():
        assert_bit_equal(head[name], small[name], f"first 777 members of 1e6 vs a 777-member run: {name}")
    want = orc.coupled_run(orc.bounds_from_values(t), np.ascontiguousarray(P[:, pick]), E,
                           dict(ts=0.0, td=0.0, conc=278.0, cum_uptake=0.0, cum_emis=0.0), threads=8)
    bounded = _bounded(want["ts"])
    assert bounded.mean() > 0.5
    for k, name in CP_NAMES.items():
        for r in rows:
            g, w = first[(name, r)][pick], want[k][r]
            assert _close(g[bounded], w[bounded], FAST_RTOL).all(), f"{name} row {r}"
        g, w = picked[name], want[k][:, :32]
        assert (np.isnan(g[0]) == np.isnan(w[0])).all(), name
        assert _close(g[1:, bounded[:32]], w[1:, bounded[:32]], FAST_RTOL).all(), f"{name}, whole series of 32 members"
    for r in rows:
        assert_bit_equal(first[("Cumulative Emissions|CO2", r)][pick], want["cum_emis"][r], f"cumulative emissions row {r}")
    assert_bit_equal(picked["Cumulative Emissions|CO2"], want["cum_emis"][:, :32], "cumulative emissions series")


def test_checkpoint_restore_resumes_bit_identically(ra):
    """Aux subsystem: checkpoint/resume (reference: Model::checkpoint / from_checkpoint,
    crates/rscm-core/src/model/runtime.rs:270-282).  A fresh handle restored from
    (time index, parameters, state rows) continues to the same bits."""
    t = axis_values(1750, 1900)
    b = np.append(t, t[-1] + 1.0)
    P, F = two_layer_params(300), f_syn(t)
    full = _tl_gpu(ra, t, P, F, 0.0, 0.0)
    with ra.Ensemble(ra.KIND_TWO_LAYER, 300, b) as e:
        e.set_params(P)
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run(60)
        ck = e.checkpoint()
    assert ck["time_index"] == 60
    with ra.Ensemble(ra.KIND_TWO_LAYER, 300, b) as e2:
        e2.set_forcing(F)
        e2.restore(ck)
        assert e2.time_index == 60
        e2.run()
        assert_bit_equal(e2.get_series(1)[60:], full[0][60:])
        assert_bit_equal(e2.get_series(2)[60:], full[1][60:])
    with ra.Ensemble(ra.KIND_TWO_LAYER, 299, b) as e3:
        with pytest.raises(ValueError, match="does not match"):
            e3.restore(ck)
    # coupled kind: five state rows
    P10 = coupled_params(64)
    E = emissions_syn(t)
    with ra.Ensemble(ra.KIND_COUPLED, 64, b) as e:
        e.set_params(P10)
        e.set_forcing(E)
        for k, v in CP_INIT.items():
            e.set_initial(k, v)
        e.run()
        want = {k: e.get_series(k) for k in e.var_ids if e.var_ids[k] > 0}
        e.rewind()
        e.run(40)
        ck = e.checkpoint()
    with ra.Ensemble(ra.KIND_COUPLED, 64, b) as e2:
        e2.set_forcing(E)
        e2.restore(ck)
        e2.run()
        for k, w in want.items():
            assert_bit_equal(e2.get_series(k)[41:], w[41:], k)


def test_fused_run_loglik_equals_stored_path(ra):
    """rscm_ens_run_loglik (no series written) == rscm_ens_run + rscm_ens_loglik, bit for bit,
    in both arithmetic modes, with observations on both variables incl. the initial row."""
    from rscm_amd import RscmGpuError
    t = axis_values()
    b = np.append(t, t[-1] + 1.0)
    n = 3000
    P, F = two_layer_params(n), f_syn(t)
    ov = np.r_[np.full(5, 2), np.full(19, 1)]            # Deep group first, then Surface
    ot = np.r_[[0, 40, 40, 300, 750], np.arange(100, 271, 10), [750]].astype(np.int32)
    val = np.linspace(0.0, 3.0, len(ot))
    sig = np.linspace(0.05, 0.5, len(ot))
    for mode in (0, 1):
        for normalize in (False, True):
            with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
                e.set_mode(mode)
                e.set_params(P)
                e.set_forcing(F)
                e.set_initial(1, 0.1)
                e.set_initial(2, -0.05)
                e.run()
                want = e.loglik(ov, ot, val, sig, normalize)
                st = e.status()
            with ra.Ensemble(ra.KIND_TWO_LAYER, n, b, store_series=False) as e:
                e.set_mode(mode)
                e.set_params(P)
                e.set_forcing(F)
                e.set_initial(1, 0.1)
                e.set_initial(2, -0.05)
                got = e.run_loglik(ov, ot, val, sig, normalize)
                assert_bit_equal(got, want, f"mode {mode} normalize {normalize}")
                assert (e.status() == st).all()
                assert e.time_index == 0
                assert (np.isinf(got) & (got < 0)).sum() == (st != 0).sum() > 0
                with pytest.raises(RscmGpuError):  # nothing is stored on this handle
                    e.run()
                with pytest.raises(RscmGpuError):
                    e.get_series(1)
                with pytest.raises(RscmGpuError, match="ascending"):
                    e.run_loglik([1, 1], [20, 10], [0.0, 0.0], [0.1, 0.1])
                with pytest.raises(RscmGpuError, match="grouped"):
                    e.run_loglik([1, 2, 1], [1, 2, 3], [0.0] * 3, [0.1] * 3)
                assert_bit_equal(e.get_series(1, 0, 1), np.full((1, n), 0.1))


def test_likelihood_only_ensemble_of_ten_million_members(ra):
    """No series buffers: 1e7 members need 480 MB of parameters + 160 MB of state instead of
    120 GB of series."""
    t = axis_values(1750, 1850)
    b = np.append(t, t[-1] + 1.0)
    n = 10_000_000
    lo = np.array([r[0] for r in TL_RANGES])
    hi = np.array([r[1] for r in TL_RANGES])
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b, store_series=False) as e:
        e.sample_lhs(SEED, lo, hi)
        e.set_forcing(f_syn(t))
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        ll = e.run_loglik([1, 1], [50, 100], [0.5, 1.0], [0.2, 0.2])
        assert ll.shape == (n,) and np.isfinite(ll).all() and (ll <= 0).all()
        pick = np.arange(0, n, n // 257)
        P = e.get_params()[:, pick]
    with ra.Ensemble(ra.KIND_TWO_LAYER, len(pick), b) as e:
        e.set_params(np.ascontiguousarray(P))
        e.set_forcing(f_syn(t))
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run()
        assert_bit_equal(e.loglik([1, 1], [50, 100], [0.5, 1.0], [0.2, 0.2]), ll[pick])


def test_long_axis_uses_large_dynamic_lds(ra, orc):
    """9001-point axis (1750-2500 in 1/12-like steps of 1/8 yr... here 1/16 yr): the forcing slice
    is 72-144 KB of LDS, above the 64 KB default dynamic limit, so the launcher raises
    hipFuncAttributeMaxDynamicSharedMemorySize; two scenarios (144 KB) still fit 160 KB, three do
    not and go through L2.  All three paths must give the oracle's bits."""
    t = 1750.0 + np.arange(9001) / 16.0
    n = 200
    P = two_layer_params(n)
    base = f_syn(t)
    for S in (1, 2, 3):
        F = np.stack([base * (1.0 + 0.1 * s) for s in range(S)])
        scen = (np.arange(n) % S).astype(np.int32)
        want = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.0, 0.0, scen=scen, h=1.0 / 64.0,
                                 threads=8)
        ts, td, _ = _tl_gpu(ra, t, P, F, 0.0, 0.0, scen=scen, h=1.0 / 64.0)
        assert_bit_equal(ts, want[0], f"S={S}")
        assert_bit_equal(td, want[1], f"S={S}")


def test_minimal_axis_single_step(ra, orc):
    """T = 2: exactly one step (the shape TwoLayer.solve uses), N = 1 and N = 65."""
    t = np.array([2000.0, 2010.0])
    for n in (1, 65):
        P = two_layer_params(n)
        F = np.array([3.0, 3.0])
        want = orc.two_layer_run(orc.bounds_from_values(t), P, F, 0.0, 0.0)
        ts, td, _ = _tl_gpu(ra, t, P, F, 0.0, 0.0)
        assert_bit_equal(ts, want[0])
        assert_bit_equal(td, want[1])
        assert ts.shape == (2, n)


def test_external_stream_and_async_run(ra, orc, tmp_path):
    """rscm_ens_set_stream with a caller-owned HIP stream (torch's), rscm_ens_run_async + sync,
    HIP-event timing of the launch; results identical to the default path.  Runs in a fresh
    process that imports torch FIRST: torch bundles its own HIP runtime, and a process must not
    initialise two of them (bench.py imports torch first for the same reason)."""
    import subprocess
    import sys
    t = axis_values(1750, 1900)
    P, F = two_layer_params(5000), f_syn(t)
    want = _tl_gpu(ra, t, P, F, 0.0, 0.0)
    np.savez(tmp_path / "in.npz", t=t, P=P, F=F)
    code = f"""
import sys, numpy as np, torch
assert torch.cuda.is_available()
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import rscm_amd
d = np.load({str(tmp_path / 'in.npz')!r})
t, P, F = d['t'], d['P'], d['F']
b = np.append(t, t[-1] + 1.0)
stream = torch.cuda.Stream()
with rscm_amd.Ensemble(rscm_amd.KIND_TWO_LAYER, 5000, b) as e:
    e.set_stream(stream.cuda_stream)
    e.set_params(P); e.set_forcing(F); e.set_initial(1, 0.0); e.set_initial(2, 0.0)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream); e.run(sync=False); ev1.record(stream)
    e.sync(); torch.cuda.synchronize()
    assert e.finished()
    assert 0.0 < e.last_run_ms() <= ev0.elapsed_time(ev1) + 0.05, (e.last_run_ms(), ev0.elapsed_time(ev1))
    ts = e.get_series(1)
    e.set_stream(None)
    e.rewind(); e.run()
    td = e.get_series(2)
np.savez({str(tmp_path / 'out.npz')!r}, ts=ts, td=td)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = np.load(tmp_path / "out.npz")
    assert_bit_equal(out["ts"], want[0])
    assert_bit_equal(out["td"], want[1])


def test_handle_churn_does_not_leak(ra):
    """300 create/configure/run/destroy cycles (all three kinds) leave device memory where it was."""
    from rscm_amd import _lib
    t = axis_values(1750, 1770)
    b = np.append(t, t[-1] + 1.0)

    def cycle(kind):
        n = 4096
        with ra.Ensemble(kind, n, b) as e:
            if kind == ra.KIND_TWO_LAYER:
                e.set_params(two_layer_params(n))
                e.set_forcing(f_syn(t))
                e.set_initial(1, 0.0)
                e.set_initial(2, 0.0)
            elif kind == ra.KIND_COUPLED:
                e.set_params(coupled_params(n))
                e.set_forcing(emissions_syn(t))
                for k, v in CP_INIT.items():
                    e.set_initial(k, v)
            else:
                from rscm_amd import _lib
                e.set_params(np.repeat(np.array(_lib.UD_DEFAULTS, dtype=float)[:, None], n, axis=1))
                e.set_forcing(f_syn(t))
                for v in (1, 2, 3, 4):
                    e.set_initial(v, 0.0)
            e.run()
            e.loglik([1], [5], [0.0], [1.0])
            e.summary(1, 5)

    for kind in (ra.KIND_TWO_LAYER, ra.KIND_COUPLED, ra.KIND_UDEB):
        cycle(kind)  # warm any lazy allocations
    free0, total = _lib.mem_info()
    assert total > 200 << 30  # an MI355X carries 288 GB
    for _ in range(100):
        for kind in (ra.KIND_TWO_LAYER, ra.KIND_COUPLED, ra.KIND_UDEB):
            cycle(kind)
    free1, _ = _lib.mem_info()
    assert free0 - free1 < 64 << 20, (free0, free1)


def test_two_handles_from_two_threads(ra, orc):
    """One handle per thread (include/rscm_gpu.h: a handle is not thread-safe, the library is): two
    threads drive their own ensembles on their own streams at the same time and get the results of
    a serial run, bit for bit; error text stays with the thread that caused it."""
    import threading
    t = axis_values(1750, 1900)
    b = np.append(t, t[-1] + 1.0)
    F = f_syn(t)
    P = [two_layer_params(20000, seed=s) for s in (1, 2)]

    def serial(p):
        with ra.Ensemble(ra.KIND_TWO_LAYER, p.shape[1], b) as e:
            e.set_params(p)
            e.set_forcing(F)
            e.set_initial(1, 0.0)
            e.set_initial(2, 0.0)
            e.run()
            return e.get_series(1)

    want = [serial(p) for p in P]
    got, errors = [None, None], [None, None]

    def worker(k):
        try:
            for _ in range(5):
                got[k] = serial(P[k])
            if k == 1:  # provoke an error on this thread only
                with ra.Ensemble(ra.KIND_TWO_LAYER, 4, b) as e:
                    try:
                        e.run()
                    except ra.RscmGpuError as exc:
                        errors[k] = str(exc)
        except Exception as exc:  # pragma: no cover
            errors[k] = f"unexpected: {exc!r}"

    th = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert errors[0] is None and errors[1] is not None and "parameters not set" in errors[1]
    for k in (0, 1):
        assert_bit_equal(got[k], want[k])


@pytest.mark.parametrize("seed", range(24))
def test_two_layer_exact_fuzz(ra, orc, seed):
    """Seeded random configurations -- axis length and irregular step lengths, RK4 step size,
    number of scenarios (LDS and L2 paths), scenario map or none, variable source, member-wise
    initial values, launch chunking -- each bit-compared with the oracle."""
    rng = np.random.default_rng(1000 + seed)
    T = int(rng.integers(2, 400))
    n = int(rng.choice([1, 2, 63, 64, 65, 300, 1025]))
    # step lengths that RK4 steps of h land on (get_last_step asserts |t_last - t1| < 5e-3):
    # binary fractions keep (t1 - t0) / h an exact integer whatever the offset
    h = float(rng.choice([0.0625, 0.125, 0.25, 0.5]))
    lengths = rng.integers(1, 9, T) * h * int(rng.choice([1, 2, 4]))
    b = np.concatenate([[1750.0], 1750.0 + np.cumsum(lengths)])
    S = int(rng.choice([1, 1, 2, 5, 40]))
    if S == 40 and T > 300:  # 40 scenarios x 300+ years exceed the LDS budget: the L2 path
        pass
    F = rng.normal(1.5, 1.5, (S, T))
    scen = rng.integers(0, S, n).astype(np.int32) if (S > 1 or rng.random() < 0.3) else None
    source = int(rng.integers(0, 2))
    P = two_layer_params(n, seed=seed)
    ts0 = rng.normal(0.0, 0.3, n) if rng.random() < 0.5 else 0.0
    td0 = rng.normal(0.0, 0.1, n) if rng.random() < 0.5 else 0.0
    want_ts, want_td = orc.two_layer_run(b, P, F, ts0, td0, scen=scen, source=source, h=h, threads=8)
    cuts = sorted(set(int(x) for x in rng.integers(0, T, int(rng.integers(0, 4)))))
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_step_size(0, h)
        e.set_params(P)
        e.set_forcing(F, scen, source)
        e.set_initial(1, ts0)
        e.set_initial(2, td0)
        for c in cuts:
            if c > e.time_index:
                e.run(c)
        e.run()
        assert_bit_equal(e.get_series(1), want_ts, f"seed {seed} Ts (T={T}, n={n}, S={S}, h={h}, source={source}, cuts={cuts})")
        assert_bit_equal(e.get_series(2), want_td, f"seed {seed} Td")


def test_summary_series_equals_row_summaries(ra, orc):
    """rscm_ens_summary_series: the plume of a variable in two launches, each row with the bits
    of the single-row summary; NaN members are left out; rows not yet computed are empty."""
    t = axis_values(1750, 1850)
    b = np.append(t, t[-1] + 1.0)
    n = 70_000
    P = two_layer_params(n)
    P[4, ::1000] = np.nan  # members whose series is NaN from the first step on
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_params(P)
        e.set_forcing(f_syn(t))
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run(60)
        s = e.summary_series("Surface Temperature")
        assert s["count"].shape == (len(t),) and s["count"][0] == n and (s["count"][61:] == 0).all()
        assert (s["count"][1:61] == n - 70).all() and np.isnan(s["mean"][61:]).all()
        for tidx in (0, 1, 30, 60, 61, 100):
            one = e.summary("Surface Temperature", tidx)
            for k in ("count", "min", "max"):
                assert s[k][tidx] == one[k], (k, tidx)
            assert (s["mean"][tidx] == one["mean"]) or (np.isnan(s["mean"][tidx]) and np.isnan(one["mean"]))
        e.run()
        part = e.summary_series("Deep Ocean Temperature", 10, 20)
        row = e.get_series(2, 10, 20)
        assert np.allclose(part["mean"], np.nanmean(row, axis=1), rtol=1e-12) and np.array_equal(part["max"], np.nanmax(row, axis=1))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("seed", range(10))
def test_coupled_chain_fuzz(ra, orc, seed, mode):
    """Seeded random configurations of the fused coupled chain -- axis length, irregular step
    lengths, the two RK4 step sizes, scenarios (LDS and L2 paths) with or without a map, member-wise
    initial values, launch chunking -- against the oracle: 1e-11 on bounded members (device exp /
    log), cumulative emissions (no transcendental) bit for bit, status flags exact."""
    rng = np.random.default_rng(5000 + seed)
    T = int(rng.integers(3, 200))
    n = int(rng.choice([1, 63, 64, 65, 300, 1025]))
    h_tl = float(rng.choice([0.0625, 0.125, 0.25]))
    h_cc = float(rng.choice([0.0625, 0.125, 0.25, 0.5]))
    lengths = rng.integers(1, 5, T) * 0.5
    b = np.concatenate([[1750.0], 1750.0 + np.cumsum(lengths)])
    S = int(rng.choice([1, 1, 3, 30]))
    E = np.abs(rng.normal(3.0, 3.0, (S, T)))
    scen = rng.integers(0, S, n).astype(np.int32) if (S > 1 or rng.random() < 0.3) else None
    P = coupled_params(n, seed=seed)
    init = dict(ts=rng.normal(0.0, 0.2, n) if rng.random() < 0.5 else 0.0, td=0.0,
                conc=rng.uniform(270.0, 300.0, n) if rng.random() < 0.5 else 278.0, cum_uptake=0.0, cum_emis=1.5)
    want = orc.coupled_run(b, P, E, init, scen=scen, h_tl=h_tl, h_cc=h_cc, threads=8)
    cuts = sorted(set(int(x) for x in rng.integers(0, T, int(rng.integers(0, 3)))))
    with ra.Ensemble(ra.KIND_COUPLED, n, b) as e:
        e.set_mode(mode)
        e.set_step_size(0, h_tl)
        e.set_step_size(1, h_cc)
        e.set_params(P)
        e.set_forcing(E, scen)
        for key, name in CP_NAMES.items():
            if key in init:
                e.set_initial(name, init[key])
        for c in cuts:
            if c > e.time_index:
                e.run(c)
        e.run()
        got = {key: e.get_series(name) for key, name in CP_NAMES.items()}
        st = e.status()
    what = f"seed {seed} mode {mode} (T={T}, n={n}, S={S}, h_tl={h_tl}, h_cc={h_cc}, cuts={cuts})"
    bounded = _bounded(want["ts"])
    for key in CP_NAMES:
        g, w = got[key], want[key]
        assert (np.isnan(g[:, bounded]) == np.isnan(w[:, bounded])).all(), f"{what} {key}"
        assert _close(g[1:, bounded], w[1:, bounded], FAST_RTOL).all(), f"{what} {key}"
    assert_bit_equal(got["cum_emis"], want["cum_emis"], f"{what} cumulative emissions")
    finite = np.isfinite(want["ts"][-1]) & np.isfinite(want["td"][-1]) & np.isfinite(want["conc"][-1]) & np.isfinite(want["cum_uptake"][-1])
    assert np.array_equal(st[bounded] == 0, finite[bounded]), what


def test_quantile_series_is_numpy_nanquantile(ra, orc):
    """rscm_ens_quantile_series: per time index numpy.nanquantile(row, q, method='linear') over the
    members -- the same order statistics and numpy's interpolation, so the same bits; NaN members left
    out, all-NaN rows NaN with count 0, rows not computed yet empty; +-inf members are ordinary
    order statistics (interpolating next to one gives what numpy's arithmetic gives)."""
    t = axis_values(1750, 1830)
    b = np.append(t, t[-1] + 1.0)
    n = 70_001
    P = two_layer_params(n)
    P[4, ::1000] = np.nan          # members whose series is NaN from the first step on
    q = [0.0, 0.05, 0.17, 0.5, 0.83, 0.95, 1.0]
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_params(P)
        e.set_forcing(f_syn(t))
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        e.run(60)
        ts = e.get_series(1)
        got = e.quantile_series(1, q)
        with np.errstate(all="ignore"):
            want = np.nanquantile(ts[:61], q, axis=1).T
        assert_bit_equal(got["quantiles"][:61], want, "quantiles of the computed rows")
        assert np.array_equal(got["count"][:61], (~np.isnan(ts[:61])).sum(axis=1))
        assert (got["count"][61:] == 0).all() and np.isnan(got["quantiles"][61:]).all()
        sub = e.quantile_series("Deep Ocean Temperature", 0.5, 10, 12)
        assert sub["quantiles"].shape == (2, 1)
        assert_bit_equal(sub["quantiles"][:, 0], np.nanmedian(e.get_series(2)[10:12], axis=1), "median")
        # an all-NaN row, and infinities among the members
        e.set_state(1, 5, np.nan)
        row = ts[6].copy()
        row[3], row[7] = np.inf, -np.inf
        e.set_state(1, 6, row)
        g = e.quantile_series(1, q, 5, 7)
        assert g["count"][0] == 0 and np.isnan(g["quantiles"][0]).all()
        with np.errstate(all="ignore"):
            w = np.nanquantile(row, q)
        assert np.array_equal(g["quantiles"][1], w, equal_nan=True)   # numpy's own inf arithmetic at q = 0 and 1 included
        assert np.isfinite(g["quantiles"][1][1:-1]).all()
        with pytest.raises(Exception, match="Quantiles must be in the range"):
            e.quantile_series(1, [1.5])


def test_params_devptr_disarms_the_uniform_row_shortcut_for_good(ra):
    """rscm_ens_params_devptr hands out the [P][N] block for device-side writers.  A caller may keep the pointer,
    upload uniform rows with rscm_ens_set_params later and then write varied values through the pointer: the
    kernels must read every member's own value (no row is treated as uniform again on such a handle)."""
    import ctypes as C
    from rscm_amd import _lib as L
    t = axis_values(1750, 1800)
    b = np.append(t, t[-1] + 1.0)
    F = f_syn(t)
    n = 300
    P = two_layer_params(n, seed=77)
    U = np.repeat(P[:, :1], n, axis=1).copy()   # every row uniform
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as ref:
        ref.set_params(P)
        ref.set_forcing(F)
        ref.set_initial(1, 0.0)
        ref.set_initial(2, 0.0)
        ref.run()
        want = ref.get_series(1)
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        ptr = C.c_void_p()
        L.check(e._lib.rscm_ens_params_devptr(e._h, C.byref(ptr)))
        e.set_params(U)                          # host upload of uniform rows AFTER the pointer went out
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.0)
        Pc = np.ascontiguousarray(P)
        L.check(e._lib.rscm_gpu_copy_to_device(0, ptr, Pc.ctypes.data_as(C.c_void_p), Pc.nbytes))
        e.run()
        assert_bit_equal(e.get_series(1), want, "varied parameters written through a cached device pointer")


def _run_plan(mode):
    from rscm_amd import _lib as L
    L.check(L.load().rscm_gpu_set_run_plan(mode))


@pytest.mark.parametrize("kind", ["two_layer", "coupled"])
@pytest.mark.parametrize("mode", [0, 1])
def test_runs_cut_into_member_blocks_and_chunks_keep_the_bits(ra, orc, kind, mode):
    """A whole-axis run over more members than the chip holds wavefronts at one per SIMD is issued as two member blocks on two streams in
    chunks of model steps (rscm_ens_last_run_plan: 2 x 12 for 750 steps).  The yardstick is the same axis as plain launches
    (rscm_gpu_set_run_plan(0)), in five pieces: the same bits -- with a scenario map (its pointer moves with the block), a ragged
    member count, and against the oracle on a sample of members from both blocks."""
    n = 100_001
    t = axis_values()
    b = np.append(t, t[-1] + 1.0)
    rng = np.random.default_rng(3)
    scen = rng.integers(0, 2, n).astype(np.int32)
    if kind == "two_layer":
        P = two_layer_params(n)
        F = np.stack([f_syn(t), 0.5 * f_syn(t)])
        names = ["Surface Temperature", "Deep Ocean Temperature"]
        init = {"Surface Temperature": 0.0, "Deep Ocean Temperature": 0.0}
        k = ra.KIND_TWO_LAYER
    else:
        P = coupled_params(n)
        F = np.stack([emissions_syn(t), 0.7 * emissions_syn(t)])
        names = list(CP_NAMES.values())
        init = CP_INIT
        k = ra.KIND_COUPLED

    source = 1 if (kind == "two_layer" and mode == 1) else 0   # (the two-layer forcing read at n + 1 in one of the four cases)

    def run(pieces):
        with ra.Ensemble(k, n, b) as e:
            e.set_mode(mode)
            e.set_params(P)
            if kind == "two_layer":
                e.set_forcing(F, scen, source)
            else:
                e.set_forcing(F, scen)
            for name, v in init.items():
                e.set_initial(name, v)
            plans = []
            for c in pieces:
                e.run(c)
                plans.append(e.last_run_plan())
            e.run()
            plans.append(e.last_run_plan())
            assert e.finished() and e.last_run_ms() > 0
            rows = {name: e.get_series(name, 0, 751, 150) for name in names}      # rows 0, 150, ..., 750 of every member
            sample = {name: e.get_series(name, 0, 751, 1, 65_500, 65_600) for name in names}   # members on both sides of the cut
            return rows, sample, plans, e.status()

    cut_rows, cut_sample, plans, st_cut = run(())
    assert plans == [(2, 12)], plans
    try:
        _run_plan(0)      # the yardstick: plain launches, one per piece
        one_rows, one_sample, plans, st_one = run((150, 300, 450, 600))
    finally:
        _run_plan(-1)
    assert plans == [(1, 1)] * 5, plans
    assert np.array_equal(st_cut, st_one)
    for name in names:
        assert_bit_equal(cut_rows[name], one_rows[name], f"{kind} mode {mode}: {name}, every member at six rows")
        assert_bit_equal(cut_sample[name], one_sample[name], f"{kind} mode {mode}: {name}, 100 members across the cut, every row")
    pick = np.arange(65_500, 65_600)
    if kind == "two_layer":
        want = orc.two_layer_run(orc.bounds_from_values(t), np.ascontiguousarray(P[:, pick]), F, 0.0, 0.0, scen=scen[pick].copy(), source=source)
        got = (cut_sample["Surface Temperature"], cut_sample["Deep Ocean Temperature"])
        if mode == 0:
            assert_bit_equal(got[0], want[0])
            assert_bit_equal(got[1], want[1])
        else:
            ok = _bounded(want[0])
            assert _close(got[0][:, ok], want[0][:, ok], FAST_RTOL).all() and _close(got[1][:, ok], want[1][:, ok], FAST_RTOL).all()
    else:
        want = orc.coupled_run(orc.bounds_from_values(t), np.ascontiguousarray(P[:, pick]), F,
                               dict(ts=0.0, td=0.0, conc=278.0, cum_uptake=0.0, cum_emis=0.0), scen=scen[pick].copy())
        ok = _bounded(want["ts"])
        for key, name in CP_NAMES.items():
            assert _close(cut_sample[name][1:, ok], want[key][1:, ok], FAST_RTOL).all(), name


@pytest.mark.parametrize("fail_at", [1, 2, 7, 24])   # the first launch of either block, one in the middle, the very last
def test_a_failed_chunk_launch_joins_the_streams_and_leaves_the_run_undone(ra, orc, fail_at):
    """The error path of the cut runs (rscm_gpu.cpp run_member_split): when a chunk's launch fails, the caller's stream is still
    joined with the handle's helper stream (nothing issued there outlives the call unseen), rscm_ens_run returns RSCM_ERR_DEVICE, the
    time index has not moved -- and the handle is intact: the same run issued again gives the uncut path's bits.  No real launch
    can be made to fail on demand: rscm_gpu_fail_chunk_launch (include/rscm_gpu_internal.h) makes the k-th chunk launch of the
    calling thread's next cut run report a failure instead of being issued."""
    from rscm_amd import _lib as L
    n = 100_001
    t = axis_values()
    b = np.append(t, t[-1] + 1.0)
    P = two_layer_params(n)
    F = f_syn(t)
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_params(P)
        e.set_forcing(F)
        e.set_initial("Surface Temperature", 0.0)
        e.set_initial("Deep Ocean Temperature", 0.0)
        L.check(L.load().rscm_gpu_fail_chunk_launch(fail_at))
        try:
            with pytest.raises(ra.RscmGpuError) as err:
                e.run()
            assert err.value.code == L.ERR_DEVICE
            assert e.time_index == 0 and not e.finished()
            e.sync()                   # the caller's stream: everything that WAS issued, on either stream, is behind this
            e.run()                    # the hook has turned itself off: the whole run, cut as usual
            assert e.last_run_plan() == (2, 12) and e.finished()
        finally:
            L.check(L.load().rscm_gpu_fail_chunk_launch(0))
        cut = {name: e.get_series(name, 0, 751, 75) for name in ("Surface Temperature", "Deep Ocean Temperature")}
        st_cut = e.status()
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as e:
        e.set_params(P)
        e.set_forcing(F)
        e.set_initial("Surface Temperature", 0.0)
        e.set_initial("Deep Ocean Temperature", 0.0)
        try:
            _run_plan(0)               # plain launches
            for c in (150, 300, 450, 600):
                e.run(c)
            e.run()
        finally:
            _run_plan(-1)
        for name in cut:
            assert_bit_equal(cut[name], e.get_series(name, 0, 751, 75), f"{name} after a failed chunk launch {fail_at}")
        assert np.array_equal(st_cut, e.status())
