"""Largest deviation of the ClimateUDEB kernel from the CPU oracle over a parameter ensemble
(run on the GPU box; the figure tests/test_gpu_udeb.py's 1e-9 tolerance is set against)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import rscm_amd as ra  # noqa: E402
from oracle import cbind as orc  # noqa: E402
import test_gpu_udeb as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nyears = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 1: RSCM_MODE_FAST
years = np.arange(1750.0, 1750.0 + nyears)
b = np.append(years, years[-1] + 1.0)
P = T._ensemble_params(orc, n, seed=3)
F = np.stack([np.where(years >= 1751, 3.71, 0.0),
              3.71 * np.log(1.01 ** np.minimum(years - 1750, 140)) / np.log(2.0),
              -1.5 * np.ones(len(years))])
scen = (np.arange(n) % 3).astype(np.int32)
want, wst = orc.udeb_run(b, P, F, scen=scen, threads=16)
got, st = T._gpu(ra, b, P, F, scen=scen, mode=mode)
print(f"mode {mode}, {n} members x {nyears} years")
assert (st == wst).all()
for k in T.NAMES:
    g, w = got[k], want[k]
    ok = ~np.isnan(w)
    err = np.abs(g[ok] - w[ok]) / np.maximum(1.0, np.abs(w[ok]))
    print(f"{k:>12s}: max rel deviation {err.max():.3e}  (bit-identical entries: {np.mean(g[ok] == w[ok]):.3f})")
