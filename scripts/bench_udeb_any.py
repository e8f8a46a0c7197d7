"""ClimateUDEB at layer counts with and without a register-resident kernel (run on the GPU box): ms per 750 years."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import rscm_amd  # noqa: E402
from rscm_amd import _lib  # noqa: E402

members = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
t = np.arange(1750.0, 2501.0)
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0))
counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (50, 49, 51, 64, 65, 80, 100, 128, 129, 200, 40, 41, 30, 25, 21, 20, 19, 2)
for nl in counts:
    lo = np.array(_lib.UD_DEFAULTS, dtype=float)
    lo[_lib.UD_PARAM_NAMES.index("n_layers")] = nl
    hi = lo.copy()
    for name, (a, b) in dict(ecs=(2.0, 5.0), kappa=(0.5, 1.5)).items():
        j = _lib.UD_PARAM_NAMES.index(name)
        lo[j], hi[j] = a, b
    with rscm_amd.Ensemble(rscm_amd.KIND_UDEB, members, np.append(t, 2501.0)) as e:
        e.sample_lhs(1, lo, hi)
        e.set_forcing(F)
        for v in (1, 2, 3, 4):
            e.set_initial(v, 0.0)
        e.run()
        e.rewind()
        e.sync()
        t0 = time.perf_counter()
        e.run()
        dt = time.perf_counter() - t0
        kind = "register-resident, count compiled in" if nl in (20, 30, 40, 50) else "register-resident, count at run time" if nl <= 64 else "column in registers, c' in LDS" if nl <= 128 else "columns in HBM"
        print(f"ClimateUDEB n_layers={nl:3d} members={members}: {dt * 1e3:8.1f} ms per 750 years ({kind} kernel), "
              f"failed members {int(e.status().astype(bool).sum())}", flush=True)
