"""Linked-ensemble graphs against the fused kernel (run on the GPU box): the coupled chain
CarbonCycle -> CO2ERF -> Sum -> TwoLayer assembled from four linked ensembles and stepped in
lock-step, beside RSCM_KIND_COUPLED (one launch for the whole run).  The four components are light,
so rscm_ens_run_lockstep fuses them into one group launch (csrc/group.hip) that covers every step;
`--unfused` issues four launches per model step instead.

    python scripts/bench_graph.py [members ...] [--unfused | --interpreter]
--interpreter: the group kernel's op interpreter with the table in device memory (rscm_gpu_set_lockstep_fusion(3)) instead of
the kernel compiled for this sequence of kinds (group_seq_kernel)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd as ra  # noqa: E402
from rscm_amd import _lib as L  # noqa: E402
from rscm_amd.ensemble import run_lockstep  # noqa: E402
from tests.helpers import axis_values, coupled_params, emissions_syn  # noqa: E402

t = axis_values(1750, 2500)
b = np.append(t, t[-1] + 1.0)
E = emissions_syn(t)
INIT = (("Atmospheric Concentration|CO2", 278.0), ("Cumulative Land Uptake", 0.0), ("Cumulative Emissions|CO2", 0.0))

ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--unfused" in sys.argv:
    L.check(L.load().rscm_gpu_set_lockstep_fusion(0))
if "--interpreter" in sys.argv:
    L.check(L.load().rscm_gpu_set_lockstep_fusion(3))
for N in ([int(a) for a in ARGS] or [100_000, 1_000_000]):
    P = coupled_params(N)
    with ra.Ensemble(ra.KIND_COUPLED, N, b) as f:
        f.set_params(P)
        f.set_forcing(E)
        for var, v in INIT + (("Surface Temperature", 0.0), ("Deep Ocean Temperature", 0.0)):
            f.set_initial(var, v)
        f.run()
        f.rewind()
        t0 = time.perf_counter()
        f.run()
        fused = time.perf_counter() - t0
        ts_fused = f.get_series("Surface Temperature", 750, 751)
    stream = C.c_void_p()
    L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
    cc, ce, ag, tl = (ra.Ensemble(k, N, b) for k in (ra.KIND_CARBON_CYCLE, ra.KIND_CO2_ERF, ra.KIND_AGGREGATE, ra.KIND_TWO_LAYER))
    for e in (cc, ce, ag, tl):
        e.set_stream(stream.value)
    cc.set_params(P[[6, 7, 8]]); ce.set_params(P[[9, 7]]); ag.set_params(np.zeros((9, N))); tl.set_params(P[:6])
    cc.set_forcing(np.stack([E, np.full(len(t), np.nan)]))
    for var, v in INIT:
        cc.set_initial(var, v)
    tl.set_initial(1, 0.0); tl.set_initial(2, 0.0)
    cc.link_input(1, tl, 1, ra.SRC_EXOGENOUS)
    ce.link_input(0, cc, 1, ra.SRC_UPSTREAM)
    ag.link_input(0, ce, 1, ra.SRC_UPSTREAM)
    tl.link_input(0, ag, 1, ra.SRC_UPSTREAM)
    best = 1e9
    for rep in range(3):
        for e in (cc, ce, ag, tl):
            e.rewind()
        tl.sync()
        t0 = time.perf_counter()
        run_lockstep((cc, ce, ag, tl))
        best = min(best, time.perf_counter() - t0)
    same = np.array_equal(tl.get_series(1, 750, 751), ts_fused, equal_nan=True)
    my = N * 750
    nl, ns = C.c_int64(), C.c_int64()
    L.check(L.load().rscm_gpu_lockstep_stats(C.byref(nl), C.byref(ns)))
    print(f"N={N}: fused coupled kernel {fused*1e3:.1f} ms ({my/fused:.3g} member-years/s); "
          f"four linked ensembles in lock-step {best*1e3:.1f} ms ({my/best:.3g} member-years/s, "
          f"{nl.value // 3} launch(es) per run); same bits: {same}", flush=True)
    cc.unlink_input(1)
    for e in (tl, ag, ce, cc):
        e.close()
    L.check(L.load().rscm_gpu_stream_destroy(0, stream))
