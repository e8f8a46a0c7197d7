"""Characterise FAST-mode deviation from the oracle (diagnostic; run on the GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rscm_amd
from oracle import cbind
from tests.helpers import axis_values, f_syn, two_layer_params

t = axis_values(); b = np.append(t, t[-1] + 1.0)
P = two_layer_params(20000); F = f_syn(t)
want = cbind.two_layer_run(cbind.bounds_from_values(t), P, F, 0.0, 0.0, threads=16)
with rscm_amd.Ensemble(0, P.shape[1], b) as e:
    e.set_mode(1); e.set_params(P); e.set_forcing(F); e.set_initial(1, 0.0); e.set_initial(2, 0.0)
    e.run(); ts = e.get_series(1)
with np.errstate(all="ignore"):
    mx = np.nanmax(np.abs(want[0]), axis=0)
    rel = np.nanmax(np.abs(ts - want[0]) / np.maximum(1.0, np.abs(want[0])), axis=0)
fin = np.isfinite(want[0][-1])
print("finite members", fin.mean())
for lo, hi in ((0, 5), (5, 8), (8, 12), (12, 20), (20, 100), (100, 1e300)):
    m = (mx >= lo) & (mx < hi) & fin
    if m.any():
        print(f"max|Ts| in [{lo},{hi}): n={m.sum()} max rel err={rel[m].max():.3e} median={np.median(rel[m]):.3e}")
m = ~fin
print("non-finite members:", m.sum())
