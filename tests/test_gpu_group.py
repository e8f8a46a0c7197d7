"""Fused lock-step launches (csrc/group.hip, rscm_gpu_set_lockstep_fusion): consecutive light components of
a model step run in one launch, every thread executing the components' per-member bodies in graph order.
Model::step (crates/rscm-core/src/model/runtime.rs:368-527) walks the same components one after the other;
every graph edge is per member, so the fused step must carry the bits of the unfused one -- checked here for
the coupled chain (against the fused coupled KERNEL as well), the MAGICC graph in both execution orders, a
windowed graph, and graphs stepped one step at a time."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import assert_bit_equal, axis_values, coupled_params, emissions_syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rscm_amd
    from rscm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1
    return rscm_amd


def _fusion(on: bool):
    from rscm_amd import _lib as L
    L.check(L.load().rscm_gpu_set_lockstep_fusion(1 if on else 0))


def _stats():
    from rscm_amd import _lib as L
    a, b = C.c_int64(), C.c_int64()
    L.check(L.load().rscm_gpu_lockstep_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


def _chain():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "bench_magicc_chain", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "bench_magicc_chain.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(autouse=True)
def _fusion_back_on():
    yield
    _fusion(True)


@pytest.mark.parametrize("mode", [0, 1])
def test_fused_coupled_chain_is_one_launch_and_keeps_the_bits(ra, mode):
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    t = axis_values(1750, 1950)
    b = np.append(t, t[-1] + 1.0)
    T, n = len(t), 1000
    P, E = coupled_params(n), emissions_syn(t)
    with ra.Ensemble(ra.KIND_COUPLED, n, b) as e:
        e.set_mode(mode)
        e.set_params(P)
        e.set_forcing(E)
        for v, x in ((1, 0.0), (2, 0.0), (3, 278.0), (4, 0.0), (5, 0.0)):
            e.set_initial(v, x)
        e.run()
        want = {v: e.get_series(v) for v in range(1, 8)}
    stream = C.c_void_p()
    L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
    cc, ce, ag, tl = (ra.Ensemble(k, n, b) for k in (ra.KIND_CARBON_CYCLE, ra.KIND_CO2_ERF, ra.KIND_AGGREGATE, ra.KIND_TWO_LAYER))
    try:
        for x in (cc, ce, ag, tl):
            x.set_stream(stream.value)
            x.set_mode(mode)
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

        def collect():
            return {1: tl.get_series(1), 2: tl.get_series(2), 3: cc.get_series(1), 4: cc.get_series(2), 5: cc.get_series(3),
                    6: ce.get_series(1), 7: ag.get_series(1)}
        _stats()
        _fusion(True)
        run_lockstep((cc, ce, ag, tl))
        launches, steps = _stats()
        # the whole graph is light: ONE launch carries all four components through all steps
        assert (launches, steps) == (1, 4 * (T - 1))
        fused = collect()
        for v in range(1, 8):
            assert_bit_equal(fused[v], want[v], f"fused lock-step vs the coupled kernel: variable {v}")
        for x in (cc, ce, ag, tl):
            x.clear_series()
        _fusion(False)
        run_lockstep((cc, ce, ag, tl))
        assert _stats() == (4 * (T - 1), 4 * (T - 1))
        plain = collect()
        for v in range(1, 8):
            assert_bit_equal(plain[v], fused[v], f"unfused vs fused: variable {v}")
        # fused again, in pieces, with single ensembles stepped by hand in between (the cached table must follow)
        for x in (cc, ce, ag, tl):
            x.clear_series()
        _fusion(True)
        run_lockstep((cc, ce, ag, tl), 7)
        for x in (cc, ce, ag, tl):
            x.run(8)
        run_lockstep((cc, ce, ag, tl), 100)
        tl.set_mode(mode)   # (setting the mode it has already: the cached op table stays valid)
        run_lockstep((cc, ce, ag, tl))
        again = collect()
        for v in range(1, 8):
            assert_bit_equal(again[v], fused[v], f"fused in pieces: variable {v}")
    finally:
        cc.unlink_input(1)
        for x in (tl, ag, ce, cc):
            x.close()
        L.check(L.load().rscm_gpu_stream_destroy(0, stream))


def test_light_graph_with_lds_slots_matches_the_unfused_run(ra):
    """A graph of light components only runs all its steps in one launch and keeps parameters, states and
    linked values in LDS slots between the steps (rscm_gpu_set_lockstep_fusion: 1 with the slots, 2 without,
    0 one launch per component and step).  Seven components with every kind of edge: a producer earlier in the
    order read at n+1 (served from its slot), a producer later in the order read at n (its slot of the previous
    step: not at a launch's first step), an earlier producer read at n (from HBM: its slot already holds n+1), more
    series and parameter rows than the slot budget holds, varying and uniform rows.  Same bits all ways,
    also when the run is cut into launches."""
    from oracle import cbind as orc
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    t = axis_values(1750, 1870)
    b = np.append(t, t[-1] + 1.0)
    T, n = len(t), 777
    rng = np.random.default_rng(11)
    yr = np.arange(T, dtype=float)

    def params(kind, vary):
        P = np.repeat(orc.pointwise_default_params(kind).reshape(-1, 1), n, axis=1)
        for k in vary:
            j = orc.PW_PARAM_NAMES[kind].index(k)
            P[j] = P[j] * rng.uniform(0.8, 1.25, n)
        return P

    stream = C.c_void_p()
    L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
    ai, ohu, tr, ospp, agg, tl, bud = (ra.Ensemble(k, n, b) for k in (
        ra.KIND_AEROSOL_INDIRECT, ra.KIND_FOURBOX_OHU, ra.KIND_AGGREGATE, ra.KIND_OSPP, ra.KIND_AGGREGATE, ra.KIND_TWO_LAYER,
        ra.KIND_CO2_BUDGET))
    graph = (ai, ohu, tr, ospp, agg, tl, bud)
    try:
        for x in graph:
            x.set_stream(stream.value)
        ai.set_params(params(orc.PW_AEROSOL_INDIRECT, ("cloud_albedo_coefficient", "reference_burden", "sox_weight")))
        ai.set_forcing(np.stack([1.0 + 0.3 * yr, 10.0 + 0.1 * yr]))
        ohu.set_params(params(orc.PW_FOURBOX_OHU, ("northern_ocean_ratio",)))
        ohu.link_input(0, agg, 1, ra.SRC_EXOGENOUS)          # a later producer, read at n
        tr.set_params(np.repeat(np.array([2.0, 0.2, 0.3, 0.1, 0.4, 0, 0, 0, 0])[:, None], n, axis=1))
        for k in range(4):
            tr.link_input(k, ohu, 1 + k, ra.SRC_UPSTREAM)     # an earlier producer, read at n+1
        tr.set_initial(1, 0.0)
        ospp.set_params(params(orc.PW_OSPP, ()))              # every row uniform
        ospp.set_forcing(np.stack([np.zeros(T), 0.2 * yr]))
        ospp.link_input(0, tl, 1, ra.SRC_EXOGENOUS)           # feedback: Surface Temperature at n
        w = np.zeros((9, n))
        w[0], w[1], w[2], w[3], w[4] = 2.0, 1.0, rng.uniform(0.05, 0.15, n), 1e-3, 0.5
        agg.set_params(w)
        F = np.full((8, T), np.nan)
        F[3] = 4.0 * (1.0 - np.exp(-yr / 60.0))
        agg.set_forcing(F)
        agg.link_input(0, ai, 1, ra.SRC_UPSTREAM)
        agg.link_input(1, tr, 1, ra.SRC_UPSTREAM)
        agg.link_input(2, ospp, 1, ra.SRC_UPSTREAM)
        agg.set_initial(1, 0.0)
        P = coupled_params(n)
        tl.set_params(P[:6])
        tl.set_initial(1, 0.0)
        tl.set_initial(2, 0.0)
        tl.link_input(0, agg, 1, ra.SRC_UPSTREAM)
        bud.set_params(np.repeat(np.array([[2.123], [278.0]]), n, axis=1))
        bud.set_forcing(np.stack([8.0 + 0.02 * yr, 1.0 + 0.0 * yr, 2.0 + 0.0 * yr, 1.5 + 0.0 * yr]))
        bud.link_input(2, tr, 1, ra.SRC_EXOGENOUS)            # an earlier producer read at n: not from its slot
        bud.set_initial(1, 278.0)
        for x in graph:
            L.check(L.load().rscm_ens_set_link_order_check(x._h, 0))

        def collect():
            return [x.get_series(v) for x in graph for v in sorted(v for v in x.var_ids.values() if v > 0)]

        def run(mode, pieces=()):
            for x in graph:
                x.rewind()
                x.clear_series()
            L.check(L.load().rscm_gpu_set_lockstep_fusion(mode))
            for stop in pieces:
                run_lockstep(graph, stop)
            run_lockstep(graph)
            return collect()

        plain = run(0)
        assert np.isfinite(plain[8][1:]).all() and np.isfinite(plain[-3][2:]).all()  # Surface Temperature, the budget's CO2
        _stats()
        slots = run(1)
        assert _stats() == (1, 7 * (T - 1))
        for k, (a, w_) in enumerate(zip(slots, plain)):
            assert_bit_equal(a, w_, f"one launch with LDS slots vs one launch per component and step: series {k}")
        for k, (a, w_) in enumerate(zip(run(2), plain)):
            assert_bit_equal(a, w_, f"one launch without the slots: series {k}")
        for k, (a, w_) in enumerate(zip(run(1, pieces=(1, 2, 40)), plain)):
            assert_bit_equal(a, w_, f"with slots, in four launches: series {k}")
        # One model step per call: every step is a one-step launch of the seven ops.  Whether the scheduler cuts it into two sets of
        # independent ops on two wavefronts is its cost model's decision (here the two-layer model in the tail outweighs what the
        # cut would save; tests/test_gpu_group.py::test_independent_ops_of_a_step_on_two_wavefronts has a graph where it cuts):
        # the bits are the same with the cut allowed (mode 1), forbidden (mode 4) and with one launch per component.
        for mode in (1, 4):
            for k, (a, w_) in enumerate(zip(run(mode, pieces=tuple(range(1, T - 1))), plain)):
                assert_bit_equal(a, w_, f"one step per launch, fusion mode {mode}: series {k}")
    finally:
        for x in graph:
            for k in range(8):
                try:
                    x.unlink_input(k)
                except Exception:
                    pass
        for x in reversed(graph):
            x.close()
        L.check(L.load().rscm_gpu_stream_destroy(0, stream))


@pytest.mark.parametrize("execution_order", ["reference", "topological"])
def test_fused_magicc_graph_keeps_the_bits(ra, execution_order):
    mod = _chain()
    years, N = 60, 200
    _fusion(False)
    plain = mod.build_chain(N, years, execution_order)
    _stats()
    plain.run()
    unfused_launches, comp_steps = _stats()
    assert unfused_launches == comp_steps == len(plain._order) * years
    _fusion(True)
    fused = mod.build_chain(N, years, execution_order)
    fused.run()
    launches, steps = _stats()
    assert steps == comp_steps
    per_step = launches / years
    print(f"{execution_order}: {len(plain._order)} components per step in {per_step:.0f} launches; order {fused._order}")
    assert per_step <= 6
    for name in plain._var_home:
        if name == "Surface Temperature":
            continue
        assert_bit_equal(fused.get_series(name), plain.get_series(name), f"{execution_order}: {name}")
    for v in range(1, 5):
        assert_bit_equal(fused.ensembles["ClimateUDEB"].get_series(v), plain.ensembles["ClimateUDEB"].get_series(v), f"box {v}")
    # step by step (Model::step) goes through the same fused launches
    fused.rewind()
    for _ in range(years):
        fused.step()
    for name in ("Atmospheric Concentration|CO2", "Effective Radiative Forcing", "Sea Surface Temperature", "Atmospheric Concentration|CH4"):
        assert_bit_equal(fused.get_series(name), plain.get_series(name), f"stepwise {name}")
    plain.close()
    fused.close()


@pytest.mark.parametrize("execution_order", ["reference", "topological"])
def test_independent_ops_of_a_step_on_two_wavefronts(ra, execution_order):
    """The fused one-step launches cut a segment's ops into two sets with no edge between them plus a tail (csrc/lockstep.cpp plan_split,
    csrc/group.hip group_split_kernel: two wavefronts per 64 members run the sets at the same time, a workgroup barrier, then the tail).
    The MAGICC graph's first segment has such a cut (the aerosol forcings beside the chemistry -> greenhouse-gas branch, the Sum of the
    forcings as the tail): the launches with the split are counted, and every series equals the run without it (fusion mode 4) and the
    unfused run (mode 0), bit for bit -- in the reference's breadth-first order too, where a consumer may run before its producer
    (they are tied to one wavefront).  A ragged last workgroup (N = 203) and windowed series are part of it."""
    import ctypes as C
    from rscm_amd import _lib as L
    mod = _chain()
    lib = L.load()
    years, N = 40, 203

    def run(mode, **kw):
        L.check(lib.rscm_gpu_set_lockstep_fusion(mode))
        L.check(lib.rscm_gpu_lockstep_split_launches(None))
        m = mod.build_chain(N, years, execution_order, **kw)
        m.set_mode(L.MODE_FAST)
        m.run()
        n_split = C.c_int64()
        L.check(lib.rscm_gpu_lockstep_split_launches(C.byref(n_split)))
        stride = kw.get("output_stride", 1)
        rows = {v: m.get_series(v, t_stride=stride) for v in sorted(m._var_home) if v != "Surface Temperature"}
        for v in range(1, 5):
            rows[f"box {v}"] = m.ensembles["ClimateUDEB"].get_series(v, t_stride=stride)
        m.close()
        return rows, n_split.value

    try:
        for kw in (dict(), dict(series_window=12, output_stride=4)):
            with_split, n_split = run(1, **kw)
            without, n_none = run(4, **kw)
            unfused, _ = run(0, **kw)
            assert n_none == 0 and n_split >= years, (n_split, n_none)   # at least the first segment of every step
            for name in with_split:
                assert_bit_equal(with_split[name], without[name], f"{execution_order} {kw}: split vs unsplit: {name}")
                assert_bit_equal(with_split[name], unfused[name], f"{execution_order} {kw}: split vs unfused: {name}")
    finally:
        L.check(lib.rscm_gpu_set_lockstep_fusion(1))


@pytest.mark.parametrize("execution_order", ["reference", "topological"])
def test_merged_launches_keep_the_bits(ra, execution_order):
    """Round 6: in graph order a step's LAST fused segment and the next step's FIRST one are consecutive launches; where both fit one
    by-value table (twelve ops) they go out as ONE launch -- the three light components behind OceanCarbon with the eight in front
    of ClimateUDEB: three launches per model step instead of four; in topological order that launch's sequence of kinds and its cut
    over two wavefronts have a kernel of their own (csrc/group.hip, group_split_seq_kernel; fusion mode 6 sends it through the op
    interpreter instead).  Same ops on the same operands in the same order:
    every series equals the unmerged run (fusion mode 5: round 5's launch plan) and the unfused run (mode 0), bit
    for bit; with windowed series (the window upkeep moves from between the two segments to behind the merged launch), a ragged last
    workgroup, and a run made in two calls (the prologue / epilogue of the merged schedule at a call boundary)."""
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    mod = _chain()
    lib = L.load()
    years, N = 30, 203

    def run(mode, halves=False, **kw):
        L.check(lib.rscm_gpu_set_lockstep_fusion(mode))
        L.check(lib.rscm_gpu_lockstep_merged_launches(None))
        L.check(lib.rscm_gpu_lockstep_own_cut_launches(None))
        m = mod.build_chain(N, years, execution_order, **kw)
        m.set_mode(L.MODE_FAST)
        _stats()
        if halves:   # two calls: [0, 7) and [7, end)
            order = [m.ensembles[name] for name in m._order]
            run_lockstep(order, 7, sync=False)
            run_lockstep(order, None, sync=True)
        else:
            m.run()
        launches, _ = _stats()
        n_merged = C.c_int64()
        L.check(lib.rscm_gpu_lockstep_merged_launches(C.byref(n_merged)))
        stride = kw.get("output_stride", 1)
        rows = {v: m.get_series(v, t_stride=stride) for v in sorted(m._var_home) if v != "Surface Temperature"}
        for v in range(1, 5):
            rows[f"box {v}"] = m.ensembles["ClimateUDEB"].get_series(v, t_stride=stride)
        steps = len(m._axis) - 1
        m.close()
        return rows, launches, n_merged.value, steps

    try:
        for kw in (dict(), dict(series_window=12, output_stride=4), dict(steps_per_year=12, series_window=16, output_stride=12)):
            merged, launches, n_merged, steps = run(1, **kw)
            own = C.c_int64()
            L.check(lib.rscm_gpu_lockstep_own_cut_launches(C.byref(own)))
            interpreted, launches6, n6, _ = run(6, **kw)   # the same launches through the op interpreter
            own6 = C.c_int64()
            L.check(lib.rscm_gpu_lockstep_own_cut_launches(C.byref(own6)))
            assert own6.value == 0 and launches6 == launches and n6 == n_merged
            # in both orders the merged launch's sequence of kinds and its cut have a kernel of their own
            assert own.value == n_merged == steps - 1, (own.value, n_merged, steps)
            for name in merged:
                assert_bit_equal(merged[name], interpreted[name], f"{execution_order} {kw}: the sequence's own kernel vs the interpreter: {name}")
            unmerged, launches5, n5, _ = run(5, **kw)
            unfused, _, _, _ = run(0, **kw)
            assert n5 == 0
            if execution_order == "topological":   # [8 light] ClimateUDEB OceanCarbon [3 light]: one launch fewer per step
                assert n_merged == steps - 1 and launches == launches5 - (steps - 1) == 3 * steps + 1, (n_merged, launches, launches5, steps)
            else:
                assert launches == launches5 - n_merged
            for name in merged:
                assert_bit_equal(merged[name], unmerged[name], f"{execution_order} {kw}: merged vs unmerged: {name}")
                assert_bit_equal(merged[name], unfused[name], f"{execution_order} {kw}: merged vs unfused: {name}")
        two_calls, _, n2, steps = run(1, halves=True)
        whole, _, _, _ = run(5)
        if execution_order == "topological":
            assert n2 == steps - 2     # each call has its own prologue and epilogue
        for name in whole:
            assert_bit_equal(two_calls[name], whole[name], f"{execution_order}: a run in two calls: {name}")
    finally:
        L.check(lib.rscm_gpu_set_lockstep_fusion(1))


def test_fused_windowed_graph_keeps_the_bits(ra):
    mod = _chain()
    years, N = 70, 128
    _fusion(False)
    plain = mod.build_chain(N, years, "reference")
    plain.run()
    _fusion(True)
    win = mod.build_chain(N, years, "reference", series_window=8, output_stride=5)
    win.run()
    for name in plain._var_home:
        if name != "Surface Temperature":
            assert_bit_equal(win.get_series(name, t_stride=5), plain.get_series(name)[::5], name)
    plain.close()
    win.close()


def test_long_chains_split_into_several_fused_launches(ra):
    """More light components in a row than one group launch takes (16): the step is cut into several fused
    launches.  Twenty chained aggregates (each the Weighted image of the one before) feeding a two-layer model,
    in one-step launches: fused == unfused, bit for bit, and 2 launches per step instead of 21."""
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    t = axis_values(1750, 1800)
    b = np.append(t, t[-1] + 1.0)
    T, n = len(t), 300
    rng = np.random.default_rng(8)
    stream = C.c_void_p()
    L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
    P_tl = np.stack([rng.uniform(lo, hi, n) for lo, hi in ((0.8, 1.5), (0.0, 0.1), (1.0, 1.8), (0.5, 1.0), (5.0, 15.0), (50.0, 200.0))])
    aggs = [ra.Ensemble(ra.KIND_AGGREGATE, n, b) for _ in range(20)]
    tl = ra.Ensemble(ra.KIND_TWO_LAYER, n, b)
    # a heavy component between the light ones forces per-step launches (otherwise the whole run would be one launch)
    ud = ra.Ensemble(ra.KIND_UDEB, n, b)
    try:
        for x in aggs + [tl, ud]:
            x.set_stream(stream.value)
        F = 3.0 * (1.0 - np.exp(-(t - 1750.0) / 30.0))
        for k, a in enumerate(aggs):
            w = np.zeros((9, n))
            w[0] = 2.0                       # Weighted
            w[1] = rng.uniform(0.9, 1.1, n)  # per-member weight of the single contributor
            a.set_params(w)
            if k == 0:
                tab = np.full((8, T), np.nan)
                tab[0] = F
                a.set_forcing(tab)
            else:
                a.link_input(0, aggs[k - 1], 1, ra.SRC_UPSTREAM)
        tl.set_params(P_tl)
        tl.set_initial(1, 0.0)
        tl.set_initial(2, 0.0)
        tl.link_input(0, aggs[-1], 1, ra.SRC_UPSTREAM)
        Pu = np.repeat(np.array(L.UD_DEFAULTS, dtype=np.float64)[:, None], n, axis=1)
        ud.set_params(Pu)
        for v in (1, 2, 3, 4):
            ud.set_initial(v, 0.0)
        ud.link_input(0, aggs[-1], 1, ra.SRC_EXOGENOUS)
        for a in aggs:
            a.set_initial(1, 0.0)           # ClimateUDEB reads its forcing at the start of step 0
        order = aggs + [tl, ud]
        out = {}
        for fused in (True, False):
            for x in order:
                x.clear_series()
            for a in aggs:
                a.set_initial(1, 0.0)
            tl.set_initial(1, 0.0)
            tl.set_initial(2, 0.0)
            for v in (1, 2, 3, 4):
                ud.set_initial(v, 0.0)
            _fusion(fused)
            _stats()
            run_lockstep(order)
            launches, steps = _stats()
            assert steps == 22 * (T - 1)
            assert launches == (3 if fused else 22) * (T - 1)   # 16 + 5 fused (aggregates + two-layer), ClimateUDEB
            out[fused] = (tl.get_series(1), aggs[-1].get_series(1), ud.get_series(7))
        for a_, b_ in zip(out[True], out[False]):
            assert_bit_equal(a_, b_, "fused in two groups vs unfused")
        assert np.isfinite(out[True][0][1:]).all()
    finally:
        for x in [ud, tl] + aggs[::-1]:
            for row in range(8):
                try:
                    x.unlink_input(row)
                except Exception:
                    pass
        for x in [ud, tl] + aggs[::-1]:
            x.close()
        L.check(L.load().rscm_gpu_stream_destroy(0, stream))


def test_two_threads_step_two_graphs_with_their_own_fusion_settings(ra):
    """The A/B switches and launch counters of rscm_ens_run_lockstep are per calling thread
    (include/rscm_gpu_internal.h): two threads step two linked coupled chains at the same time, one fused with
    LDS slots, one unfused, each sees its own launch count, and both get the bits of a single-thread run."""
    import threading
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    t = axis_values(1750, 1900)
    b = np.append(t, t[-1] + 1.0)
    T = len(t)
    E = emissions_syn(t)

    def chain(n, seed, mode, out):
        L.check(L.load().rscm_gpu_set_lockstep_fusion(mode))
        P = coupled_params(n, seed=seed)
        stream = C.c_void_p()
        L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
        cc, ce, ag, tl = (ra.Ensemble(k, n, b) for k in (ra.KIND_CARBON_CYCLE, ra.KIND_CO2_ERF, ra.KIND_AGGREGATE, ra.KIND_TWO_LAYER))
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
            a, bb = C.c_int64(), C.c_int64()
            L.check(L.load().rscm_gpu_lockstep_stats(None, None))
            for _ in range(3):
                for x in (cc, ce, ag, tl):
                    x.clear_series()
                run_lockstep((cc, ce, ag, tl))
            L.check(L.load().rscm_gpu_lockstep_stats(C.byref(a), C.byref(bb)))
            out["launches"] = a.value
            out["series"] = [tl.get_series(1), tl.get_series(2), cc.get_series(1), ag.get_series(1)]
        except Exception as exc:  # pragma: no cover
            out["error"] = repr(exc)
        finally:
            cc.unlink_input(1)
            for x in (tl, ag, ce, cc):
                x.close()
            L.check(L.load().rscm_gpu_stream_destroy(0, stream))

    jobs = [(3000, 11, 1), (2000, 12, 0)]
    want = [dict() for _ in jobs]
    for (n, seed, mode), w in zip(jobs, want):   # single thread first
        chain(n, seed, mode, w)
        assert "error" not in w, w.get("error")
    got = [dict() for _ in jobs]
    th = [threading.Thread(target=chain, args=(n, seed, mode, g)) for (n, seed, mode), g in zip(jobs, got)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for g, w, (n, seed, mode) in zip(got, want, jobs):
        assert "error" not in g, g.get("error")
        assert g["launches"] == w["launches"] == (3 if mode == 1 else 3 * 4 * (T - 1))
        for k, (x, y) in enumerate(zip(g["series"], w["series"])):
            assert_bit_equal(x, y, f"two threads vs one, fusion mode {mode}: series {k}")
    # the switches of the worker threads never touched this thread's: still the default (fused)
    w2 = dict()
    L.check(L.load().rscm_gpu_lockstep_stats(None, None))
    L.check(L.load().rscm_gpu_set_lockstep_fusion(1))
    chain(2000, 12, 1, w2)
    for k, (x, y) in enumerate(zip(w2["series"], want[1]["series"])):
        assert_bit_equal(x, y, f"fused vs unfused: series {k}")


@pytest.mark.parametrize("seed", range(8))
def test_random_light_graphs_with_and_without_the_cut(ra, seed):
    """Random graphs of five to eight light ops -- schema aggregates (Sum / Weighted over links and table rows: NaN contributors are
    skipped, so rows a consumer reads before its producer has written them do not poison the run) and FourBoxOceanHeatUptake ops --
    with random links: read at n or at n + 1, to ops earlier OR later in the order (the reference's breadth-first order allows a
    consumer before its producer; such pairs must stay on one wavefront).  Stepped one model step per call, so every step is a one-step
    fused launch that the scheduler may cut into two sets of independent ops on two wavefronts: with the cut allowed (mode 1),
    forbidden (mode 4) and with one launch per op (mode 0) every series carries the same bits."""
    from rscm_amd import _lib as L
    from rscm_amd.ensemble import run_lockstep
    rng = np.random.default_rng(900 + seed)
    T, n = 25, int(rng.choice([64, 130, 777]))
    b = np.arange(T + 1, dtype=float) + 1750.0
    yr = np.arange(T, dtype=float)
    n_ops = int(rng.integers(5, 9))
    kinds = [ra.KIND_FOURBOX_OHU if rng.random() < 0.3 else ra.KIND_AGGREGATE for _ in range(n_ops)]
    stream = C.c_void_p()
    L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
    graph = [ra.Ensemble(k, n, b) for k in kinds]
    linked = []
    try:
        for x, k in zip(graph, kinds):
            x.set_stream(stream.value)
            if k == ra.KIND_AGGREGATE:
                w = np.zeros((9, n))
                w[0] = float(rng.choice([0.0, 2.0]))                      # Sum or Weighted
                w[1:] = rng.uniform(0.1, 0.4, (8, 1)) * rng.uniform(0.9, 1.1, (8, n))
                x.set_params(w)
                rows = np.full((8, T), np.nan)
                for r in rng.choice(8, int(rng.integers(1, 4)), replace=False):
                    rows[r] = rng.uniform(0.5, 2.0) * np.sin(yr / rng.uniform(3.0, 9.0)) + 1.0
                x.set_forcing(rows)
                x.set_initial(1, 0.25)
            else:
                x.set_params(np.repeat(np.array([[0.3], [0.2], [0.35], [0.15]]), n, axis=1) * rng.uniform(0.9, 1.1, (4, n)))
                x.set_forcing((0.5 + 0.1 * yr)[None])
        for j, (x, k) in enumerate(zip(graph, kinds)):
            slots = list(range(8)) if k == ra.KIND_AGGREGATE else [0]
            for slot in rng.choice(slots, min(len(slots), int(rng.integers(1, 4))), replace=False):
                src = int(rng.integers(0, n_ops))
                if src == j:
                    continue
                var = 1 if kinds[src] == ra.KIND_AGGREGATE else int(rng.integers(1, 5))
                x.link_input(int(slot), graph[src], var, ra.SRC_UPSTREAM if rng.random() < 0.6 else ra.SRC_EXOGENOUS)
                linked.append((x, int(slot)))
        for x in graph:
            L.check(L.load().rscm_ens_set_link_order_check(x._h, 0))

        def run(mode):
            for x in graph:
                x.rewind()
                x.clear_series()
            L.check(L.load().rscm_gpu_set_lockstep_fusion(mode))
            L.check(L.load().rscm_gpu_lockstep_split_launches(None))
            for stop in range(1, T):
                run_lockstep(graph, stop)
            cut = C.c_int64()
            L.check(L.load().rscm_gpu_lockstep_split_launches(C.byref(cut)))
            return [x.get_series(v) for x in graph for v in sorted(v for v in x.var_ids.values() if v > 0)], cut.value

        plain, _ = run(0)
        with_cut, n_cut = run(1)
        without, n_none = run(4)
        assert n_none == 0
        print(f"seed {seed}: {n_ops} ops {['agg' if k == ra.KIND_AGGREGATE else 'ohu' for k in kinds]}, {len(linked)} links, "
              f"{n_cut} of {T - 1} steps cut")
        for k, (a, w_, p_) in enumerate(zip(with_cut, without, plain)):
            assert_bit_equal(a, p_, f"seed {seed}: with the cut vs one launch per op: series {k}")
            assert_bit_equal(w_, p_, f"seed {seed}: without the cut vs one launch per op: series {k}")
    finally:
        L.check(L.load().rscm_gpu_set_lockstep_fusion(1))
        for x, slot in linked:
            try:
                x.unlink_input(slot)
            except Exception:
                pass
        for x in reversed(graph):
            x.close()
        L.check(L.load().rscm_gpu_stream_destroy(0, stream))
