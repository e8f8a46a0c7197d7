"""ClimateUDEB on its own against the same ensemble inside a two-component graph (a one-contributor Sum -> ClimateUDEB):
whole-axis launch, whole-graph launch (csrc/graph.hip) and one launch per component and step.
    python scripts/udeb_in_graph.py [members ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import rscm_amd as ra  # noqa: E402
from rscm_amd import _lib as L  # noqa: E402
from rscm_amd.ensemble import run_lockstep  # noqa: E402

lib = L.load()
for members in [int(a) for a in sys.argv[1:]] or [65_536, 125_000]:
    stream = C.c_void_p()
    L.check(lib.rscm_gpu_stream_create(0, C.byref(stream)))
    ud = bench.make_udeb_ensemble(members, 0, stream.value)
    ud.run()
    alone = ud.last_run_ms() / 750 * 1e3
    ref = ud.get_series("Sea Surface Temperature", 750, 751)[0]
    t = np.arange(bench.T0, bench.T1 + 1, dtype=np.float64)
    ag = ra.Ensemble(ra.KIND_AGGREGATE, members, np.append(t, t[-1] + 1.0))
    ag.set_stream(stream.value)
    ag.set_params(np.zeros((9, members)))
    F = bench.f_syn(t)
    rows = np.full((8, len(t)), np.nan)
    rows[0] = F
    ag.set_forcing(rows[None])
    ag.set_initial(1, F[0])
    ud.link_input(0, ag, 1, ra.SRC_EXOGENOUS)
    res = {}
    for label, mode in (("whole-graph launch", 4), ("one launch per component and step", 1)):
        L.check(lib.rscm_gpu_set_lockstep_fusion(mode))
        for x in (ag, ud):
            x.clear_series()
        ag.set_initial(1, F[0])
        for v in (1, 2, 3, 4):
            ud.set_initial(v, 0.0)
        run_lockstep((ag, ud), 10, sync=True)
        t0 = time.perf_counter()
        run_lockstep((ag, ud), sync=True)
        res[label] = (time.perf_counter() - t0) / 740 * 1e6
        got = ud.get_series("Sea Surface Temperature", 750, 751)[0]
        res[label + " same bits"] = bool(np.array_equal(got, ref))
    print(f"udeb {members}: alone, whole axis {alone:.1f} us per step; " + "; ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in res.items()), flush=True)
    ud.unlink_input(0)
    ag.close()
    ud.close()
    L.check(lib.rscm_gpu_stream_destroy(0, stream))
