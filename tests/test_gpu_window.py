"""Windowed + strided series storage (RSCM_FLAG_WINDOWED, ModelBuilder.build(series_window=...)):
a sliding window of rows per series plus every k-th row of the outputs, instead of whole series.
The reference keeps whole collections (model/builder.rs:735-830) and reads index n or n+1 of a
producer (state/windows.rs:229-234); what is under test is that dropping the rows nobody can read any
more changes no bit of what is kept, for single ensembles, linked graphs in lock-step (both execution
orders), look-back kinds, rewinds and checkpoints -- and that BASELINE.json configs[3] (125 000 members
per GPU x 9001 monthly points x the ten-component MAGICC graph) fits one MI355X that way."""
import numpy as np
import pytest

from tests.helpers import assert_bit_equal, axis_values, coupled_params, emissions_syn, f_syn, two_layer_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rscm_amd
    from rscm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1
    return rscm_amd


def _chain():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "bench_magicc_chain", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "bench_magicc_chain.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_windowed_two_layer_keeps_the_bits(ra):
    from rscm_amd import RscmGpuError
    t = axis_values(1750, 1900)
    b = np.append(t, t[-1] + 1.0)
    T, n = len(t), 300
    P, F = two_layer_params(n), f_syn(t)
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as full:
        full.set_params(P)
        full.set_forcing(F)
        full.set_initial(1, 0.0)
        full.set_initial(2, 0.1)
        full.run()
        want = {v: full.get_series(v) for v in (1, 2)}
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b, window_rows=8, output_stride=5, output_vars=["Surface Temperature"]) as e:
        e.set_params(P)
        e.set_forcing(F)
        e.set_initial(1, 0.0)
        e.set_initial(2, 0.1)
        with pytest.raises(RscmGpuError, match="do not fit a window"):
            e.run()                       # the whole axis in one launch cannot be windowed
        for k in range(T - 1):
            e.step()
            if k == 40:                   # the rows of the window are readable while they are resident
                assert_bit_equal(e.get_series(2, 39, 42), want[2][39:42], "window rows")
                with pytest.raises(RscmGpuError, match="not resident"):
                    e.get_series(2, 20, 21)
        assert e.finished()
        assert_bit_equal(e.get_series(1, 0, T, 5), want[1][::5], "strided Ts")
        with pytest.raises(RscmGpuError, match="not resident"):
            e.get_series(2, 0, T, 5)      # Td is not an output variable
        with pytest.raises(RscmGpuError, match="not resident"):
            e.get_series(1, 1, 2)
        assert_bit_equal(e.get_series(2, T - 2, T), want[2][T - 2:], "last window rows")
        # likelihood and summary read resident rows
        tidx = np.arange(10, 150, 10, dtype=np.int32)
        ll = e.loglik(np.ones(len(tidx), dtype=np.int32), tidx, want[1][tidx, 0], np.full(len(tidx), 0.3))
        assert ll[0] == 0.0 and np.isfinite(ll[np.isfinite(want[1][-1])]).all()
        with pytest.raises(RscmGpuError, match="not resident"):
            e.loglik([1], [7], [0.0], [1.0])
        assert e.summary(1, 100)["count"] == np.isfinite(want[1][100]).sum()
        # a second run after rewind: the initial rows come back although they left the window long ago
        e.rewind()
        for _ in range(T - 1):
            e.step()
        assert_bit_equal(e.get_series(1, 0, T, 5), want[1][::5], "strided Ts, second run")
        # new initial values after a finished run
        e.set_initial(1, 0.5)
        assert e.get_series(1, 0, 1)[0, 0] == 0.5 and e.get_series(2, 0, 1)[0, 0] == 0.1   # Td's initial row came back with the rewind
        for _ in range(12):
            e.step()
        assert e.get_series(1, 0, 1)[0, 0] == 0.5
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b) as full:
        full.set_params(P)
        full.set_forcing(F)
        full.set_initial(1, 0.5)
        full.set_initial(2, 0.1)
        full.run(12)
        again = full.get_series(1, 0, 13)
    with ra.Ensemble(ra.KIND_TWO_LAYER, n, b, window_rows=8, output_stride=5) as e:
        e.set_params(P)
        e.set_forcing(F)
        e.set_initial(1, 0.5)
        e.set_initial(2, 0.1)
        for _ in range(12):
            e.step()
        assert_bit_equal(e.get_series(1, 0, 13, 5), again[::5], "fresh windowed run with the new initial value")
        e.run(15)                         # a range that fits the window
        assert e.time_index == 15


def test_windowed_n2o_looks_back_through_its_window(ra):
    """N2OChemistry reads its own concentration up to strat_delay + 1 steps back (n2o.rs:203-218): the
    window keeps those rows across every slide; too short a window is refused."""
    from oracle import cbind as orc
    from rscm_amd import RscmGpuError
    n, T = 200, 121
    b = np.arange(T + 1, dtype=float) + 1850.0
    yr = np.arange(T, dtype=float)
    P = np.repeat(orc.chem_default_params(orc.CHEM_N2O).reshape(-1, 1), n, axis=1)
    names = orc.CHEM_PARAM_NAMES[orc.CHEM_N2O]
    P[names.index("strat_delay")] = (np.arange(n) % 5).astype(float)
    P[names.index("tau_n2o")] = np.random.default_rng(1).uniform(110.0, 160.0, n)
    inputs = (8.0 + 0.05 * yr)[None, None, :]
    with ra.Ensemble(ra.KIND_N2O_CHEMISTRY, n, b) as full:
        full.set_params(P)
        full.set_forcing(inputs)
        full.set_initial(1, 270.0)
        full.run()
        want = full.get_series(1), full.get_series(2)
    with ra.Ensemble(ra.KIND_N2O_CHEMISTRY, n, b, window_rows=12, output_stride=3) as e:
        e.set_params(P)
        e.set_forcing(inputs)
        e.set_initial(1, 270.0)
        for _ in range(T - 1):
            e.step()
        assert_bit_equal(e.get_series(1, 0, T, 3), want[0][::3], "N2O concentration")
        assert_bit_equal(e.get_series(2, 0, T, 3), want[1][::3], "N2O lifetime")
    with ra.Ensemble(ra.KIND_N2O_CHEMISTRY, n, b, window_rows=8) as e:
        e.set_params(P)
        e.set_forcing(inputs)
        e.set_initial(1, 270.0)
        with pytest.raises(RscmGpuError, match="too short"):
            e.step()
    # a longer delay on a window that has already slid: the rows it looks back at were not kept
    Q = P.copy()
    Q[names.index("strat_delay")] = 1.0
    with ra.Ensemble(ra.KIND_N2O_CHEMISTRY, n, b, window_rows=16) as e:
        e.set_params(Q)
        e.set_forcing(inputs)
        e.set_initial(1, 270.0)
        for _ in range(29):     # the 16-row window slid at the end of step 28: it now starts at row 26 (delay 1: 3 rows kept)
            e.step()
        R = Q.copy()
        R[names.index("strat_delay")] = 5.0
        with pytest.raises(RscmGpuError, match="window already starts"):
            e.set_params(R)
        e.step()                # still consistent with the old delay
        e.rewind()
        e.set_params(R)         # from the start of the axis the longer look-back is fine
        for _ in range(40):
            e.step()
        got = e.get_series(1, 30, 41, 1)
    with ra.Ensemble(ra.KIND_N2O_CHEMISTRY, n, b) as full:
        full.set_params(R)
        full.set_forcing(inputs)
        full.set_initial(1, 270.0)
        full.run(40)
        assert_bit_equal(got, full.get_series(1)[30:41], "N2O with the longer delay after a rewind")


def test_windowed_linked_coupled_chain_equals_the_fused_kernel(ra):
    """CarbonCycle, CO2ERF, Sum, TwoLayer as four windowed linked ensembles in lock-step (lagged
    temperature feedback across the slides): the annual-strided rows equal the fused coupled kernel's."""
    import ctypes as C
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    t = axis_values(1750, 2050)
    b = np.append(t, t[-1] + 1.0)
    T, n = len(t), 257
    P, E = coupled_params(n), emissions_syn(t)
    with ra.Ensemble(ra.KIND_COUPLED, n, b) as e:
        e.set_params(P)
        e.set_forcing(E)
        for v, x in ((1, 0.0), (2, 0.0), (3, 278.0), (4, 0.0), (5, 0.0)):
            e.set_initial(v, x)
        e.run()
        want = {v: e.get_series(v) for v in range(1, 8)}
    stream = C.c_void_p()
    L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
    kw = dict(window_rows=6, output_stride=4)
    cc, ce, ag, tl = (ra.Ensemble(k, n, b, **kw) for k in (ra.KIND_CARBON_CYCLE, ra.KIND_CO2_ERF, ra.KIND_AGGREGATE, ra.KIND_TWO_LAYER))
    try:
        for x in (cc, ce, ag, tl):
            x.set_stream(stream.value)
        cc.set_params(P[[6, 7, 8]])
        ce.set_params(P[[9, 7]])
        ag.set_params(np.zeros((9, n)))
        tl.set_params(P[:6])
        cc.set_forcing(np.stack([E, np.full(T, np.nan)]))
        for v, x in ((1, 278.0), (2, 0.0), (3, 0.0)):
            cc.set_initial(v, x)
        tl.set_initial(1, 0.0)
        tl.set_initial(2, 0.0)
        cc.link_input(1, tl, 1, ra.SRC_EXOGENOUS)
        ce.link_input(0, cc, 1, ra.SRC_UPSTREAM)
        ag.link_input(0, ce, 1, ra.SRC_UPSTREAM)
        tl.link_input(0, ag, 1, ra.SRC_UPSTREAM)
        run_lockstep((cc, ce, ag, tl))
        got = {1: tl.get_series(1, 0, T, 4), 2: tl.get_series(2, 0, T, 4), 3: cc.get_series(1, 0, T, 4), 4: cc.get_series(2, 0, T, 4),
               5: cc.get_series(3, 0, T, 4), 6: ce.get_series(1, 0, T, 4), 7: ag.get_series(1, 0, T, 4)}
        for v in range(1, 8):
            assert_bit_equal(got[v], want[v][::4], f"variable {v}")
    finally:
        cc.unlink_input(1)
        for x in (tl, ag, ce, cc):
            x.close()
        L.check(L.load().rscm_gpu_stream_destroy(0, stream))


@pytest.mark.parametrize("execution_order", ["reference", "topological"])
def test_windowed_magicc_graph_equals_full_storage(ra, execution_order):
    """The thirteen-ensemble emissions-driven MAGICC graph with a 16-row window and every 6th row of every
    variable kept, against the same graph storing whole series: every kept row carries the same bits.  In
    the reference's breadth-first order the forcing aggregate reads rows its producers have not written
    yet and must find NaN there after every slide of their windows."""
    mod = _chain()
    years, N = 90, 96
    full = mod.build_chain(N, years, execution_order)
    full.run()
    win = mod.build_chain(N, years, execution_order, series_window=16, output_stride=6)
    assert win._windowed and all(e.window_rows == 16 for e in win.ensembles.values())
    win.run()
    T = years + 1
    for name in full._var_home:
        if name == "Surface Temperature":
            continue
        assert_bit_equal(win.get_series(name, t_stride=6), full.get_series(name)[::6], f"{execution_order}: {name}")
    ud_f, ud_w = full.ensembles["ClimateUDEB"], win.ensembles["ClimateUDEB"]
    for v in range(1, 5):
        assert_bit_equal(ud_w.get_series(v, 0, T, 6), ud_f.get_series(v)[::6], f"surface temperature box {v}")
    erf = full.get_series("Effective Radiative Forcing")
    assert np.isfinite(erf[1:]).all() and np.isfinite(full.get_series("Sea Surface Temperature")[1:]).all()
    # a second run of the same windowed model object
    win.rewind()
    win.run()
    for name in ("Atmospheric Concentration|CO2", "Effective Radiative Forcing", "Sea Surface Temperature", "Atmospheric Concentration|N2O"):
        assert_bit_equal(win.get_series(name, t_stride=6), full.get_series(name)[::6], f"second run: {name}")
    # a checkpoint in the middle of a windowed run, restored into a fresh windowed model
    win.rewind()
    for _ in range(41):
        win.step()
    ck = win.checkpoint()
    win.close()
    other = mod.build_chain(N, years, execution_order, series_window=16, output_stride=6)
    other.restore(ck)
    other.run()
    for name in ("Atmospheric Concentration|CO2", "Effective Radiative Forcing", "Sea Surface Temperature", "Carbon Flux|Ocean",
                 "Atmospheric Concentration|N2O", "Atmospheric Concentration|CH4"):
        got, ref = other.get_series(name, t_begin=42, t_stride=6), full.get_series(name)[42::6]
        assert_bit_equal(got, ref, f"resumed from a checkpoint: {name}")
    other.close()
    full.close()


def test_windowed_outputs_can_be_a_subset(ra):
    mod = _chain()
    years, N = 30, 64
    names = ["Sea Surface Temperature", "Atmospheric Concentration|CO2", "Effective Radiative Forcing", "Surface Temperature"]
    win = mod.build_chain(N, years, "topological", series_window=8, output_stride=10, outputs=names)
    win.run()
    full = mod.build_chain(N, years, "topological")
    full.run()
    for name in names[:3]:
        assert_bit_equal(win.get_series(name, t_stride=10), full.get_series(name)[::10], name)
    from rscm_amd import RscmGpuError
    with pytest.raises(RscmGpuError, match="not resident"):
        win.get_series("Carbon Pool|Soil", t_stride=10)
    with pytest.raises(NotImplementedError):
        win.timeseries()
    win.close()
    full.close()


def test_configs3_shape_fits_one_gpu(ra):
    """BASELINE.json configs[3], one GPU's share: 125 000 members x the ten-component MAGICC graph x 9001
    monthly points (1750-2500).  Stored whole that is 36 series x 9001 x 8 B = 2.6 MB per member = 324 GB;
    with a 16-row window and annual (every 12th) outputs of every variable the graph allocates well under
    250 GB -- and steps."""
    from rscm_amd import _lib as L
    mod = _chain()
    free0, total = L.mem_info(0)
    N, years, spy = 125_000, 750, 12
    model = mod.build_chain(N, years, "topological", steps_per_year=spy, series_window=16, output_stride=12)
    free1, _ = L.mem_info(0)
    used = free0 - free1
    print(f"125 000 members x 9001 monthly points: {used / 2**30:.1f} GiB allocated of {total / 2**30:.0f} GiB")
    assert used < 250e9
    for _ in range(26):
        model.step()
    sst = model.get_series("Sea Surface Temperature", t_begin=24, t_end=25)
    co2 = model.get_series("Atmospheric Concentration|CO2", t_begin=0, t_end=25, t_stride=12)
    assert np.isfinite(sst).all() and np.isfinite(co2).all() and (co2[2] != co2[0]).any()
    assert not model.ensembles["ClimateUDEB"].status().any()
    model.close()
    free2, _ = L.mem_info(0)
    assert abs(free2 - free0) < 1 << 30


def test_configs3_share_runs_at_full_size(ra):
    """BASELINE.json configs[3], one GPU's share AT FULL SIZE: 125 000 members x 9000 monthly steps (1750-2500)
    of the ten-component MAGICC graph (RSCM_MODE_FAST: OceanCarbon's O(T) recurrence; 16-row window, annual
    outputs).  Checked: no member fails, THREE launches per model step (round 6: the step's last light segment rides with the next step's first), the ensemble ends warm with CO2 above
    pre-industrial -- and the first 64 members equal a 64-member run given their parameters, bit for bit, on
    every kept row of five variables (nothing depends on the ensemble size, the position in a wavefront or the
    fused launches).  scripts/run_configs3_share.py is the same thing as a program (profiles/r2_configs3_share_*)."""
    import json
    import os
    # the script is imported, not spawned: a GPU process must not exec other programs on this pool
    import sys
    import io
    import contextlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from scripts import run_configs3_share as prog
    argv, sys.argv = sys.argv, ["run_configs3_share.py"]
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            with pytest.raises(SystemExit) as done:
                prog.main()
    finally:
        sys.argv = argv
    assert done.value.code == 0
    out = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(f"configs[3] share: {out['run_s']:.2f} s, {out['hbm_allocated_gib']:.0f} GiB, {out['launches_per_step']:.0f} launches per step, "
          f"{out['member_years_per_s']:.3g} member-years/s")
    assert out["failed_members"] == 0 and 3.0 <= out["launches_per_step"] <= 3.001 and out["hbm_allocated_gib"] < 250e9 / 2**30
    assert all(out["first_64_members_equal_a_64_member_run"].values())
    assert out["warming_end_K"]["count"] == 125_000 and 1.0 < out["warming_end_K"]["mean"] < 12.0
    assert out["co2_end_ppm"]["min"] > 278.0


def test_configs3_share_exact_mode_and_fast_against_it(ra):
    """BASELINE.json configs[3], one GPU's share in RSCM_MODE_EXACT (OceanCarbon's literal history convolution, the
    reference's summation order): 125 000 members x 600 MONTHLY steps -- 42 slides of the 16-row windows, 600 x 12
    pulses into OceanCarbon's flux-history ring, 50 annual output rows.  The first 64 members equal a 64-member EXACT run
    given their parameters, bit for bit; and the RSCM_MODE_FAST run of the same ensemble (OceanCarbon's O(T)
    recurrence over fitted modes -- an approximation of the algorithm, not only of the rounding) stays within 1e-11
    relative of EXACT on every kept row of five variables (measured 3.4e-13: printed)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from scripts import run_configs3_share as prog
    N, years = 125_000, 50
    big, rows = prog.run(N, years, True)
    assert big["failed"] == 0
    print(f"configs[3] share, EXACT, {years * 12} monthly steps: {big['run_s']:.2f} s, {big['launches'] / (years * 12):.1f} launches per step")
    small = prog.first_64(N, years, True)
    for name in prog.NAMES:
        assert rows[name].shape == (years + 1, N)
        assert_bit_equal(rows[name][:, :64], small[name], f"EXACT, first 64 of {N} members vs a 64-member run: {name}")
    fast_info, fast = prog.run(N, years, False)
    assert fast_info["failed"] == 0
    worst = 0.0
    for name in prog.NAMES:
        ok = ~np.isnan(rows[name])
        assert (np.isnan(fast[name]) == np.isnan(rows[name])).all()
        err = np.abs(fast[name][ok] - rows[name][ok]) / np.maximum(1.0, np.abs(rows[name][ok]))
        worst = max(worst, float(err.max()))
        assert err.max() <= 1e-11, (name, err.max())
    print(f"FAST vs EXACT over {years * 12} monthly steps, {N} members: max relative deviation {worst:.2e}")


def test_windowed_graph_in_fast_mode(ra):
    """RSCM_MODE_FAST (OceanCarbon's O(T) recurrence with running mode sums as extra internal state) under
    windowed storage: the kept rows equal the full-storage FAST run bit for bit; a checkpoint restored into a
    fresh model re-forms the mode sums from the flux history (Horner instead of the incremental recurrence), so
    the continuation agrees to the FAST tolerance, not to the bit."""
    from rscm_amd import _lib as L
    mod = _chain()
    years, N = 80, 96
    full = mod.build_chain(N, years, "topological")
    full.set_mode(L.MODE_FAST)
    full.run()
    win = mod.build_chain(N, years, "topological", series_window=12, output_stride=4)
    win.set_mode(L.MODE_FAST)
    win.run()
    names = ["Atmospheric Concentration|CO2", "Carbon Flux|Ocean", "Ocean Surface pCO2", "Sea Surface Temperature", "Effective Radiative Forcing"]
    for name in names:
        assert_bit_equal(win.get_series(name, t_stride=4), full.get_series(name)[::4], f"FAST, windowed: {name}")
    win.rewind()
    for _ in range(37):
        win.step()
    ck = win.checkpoint()
    win.close()
    other = mod.build_chain(N, years, "topological", series_window=12, output_stride=4)
    other.set_mode(L.MODE_FAST)
    other.restore(ck)
    other.run()
    for name in names:
        got, ref = other.get_series(name, t_begin=40, t_stride=4), full.get_series(name)[40::4]
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        assert np.isfinite(got).all() and err.max() <= 2e-8, (name, err.max())
    other.close()
    full.close()
