"""ClimateUDEB stand-alone timing (run on the GPU box): N members x 750 years, best of 3 launches.
    python scripts/bench_udeb.py [--fast] [members ...]      (--fast: RSCM_MODE_FAST, one refinement term per row reciprocal)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

FAST = "--fast" in sys.argv
for members in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [65_536, 100_000]:
    e = bench.make_udeb_ensemble(members, 0)
    if FAST:
        e.set_mode(1)
    ms = []
    for _ in range(3):
        e.rewind()
        e.run()
        ms.append(e.last_run_ms())
    sst = e.get_series("Sea Surface Temperature", 750, 751)[0]
    tag = " FAST" if FAST else ""
    print(f"udeb{tag} {members} members x 750 years: {min(ms):.1f} ms ({members * 750 / min(ms) * 1e3:.3g} member-years/s); "
          f"SST[2500] mean {np.nanmean(sst):.6f}, finite {np.isfinite(sst).sum()}", flush=True)
    e.close()
