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

# ---- the coupled chain, RSCM_MODE_FAST and RSCM_MODE_EXACT against the oracle (all seven series)
from tests.helpers import coupled_params, emissions_syn
Pc = coupled_params(20000); E = emissions_syn(t)
init = dict(ts=0.0, td=0.0, conc=278.0, cum_uptake=0.0, cum_emis=0.0)
names = {"ts": "Surface Temperature", "td": "Deep Ocean Temperature", "conc": "Atmospheric Concentration|CO2",
         "cum_uptake": "Cumulative Land Uptake", "cum_emis": "Cumulative Emissions|CO2",
         "erf_co2": "Effective Radiative Forcing|CO2", "erf_total": "Effective Radiative Forcing"}
wantc = cbind.coupled_run(cbind.bounds_from_values(t), Pc, E, init, threads=16)
ok = np.isfinite(wantc["ts"][-1]) & (np.nanmax(np.abs(wantc["ts"]), axis=0) < 50.0)
for mode in (0, 1):
    with rscm_amd.Ensemble(rscm_amd.KIND_COUPLED, Pc.shape[1], b) as e:
        e.set_mode(mode); e.set_params(Pc); e.set_forcing(E)
        for k, v in init.items():
            e.set_initial(names[k], v)
        e.run()
        for k, name in names.items():
            g, w = e.get_series(name)[1:, ok], wantc[k][1:, ok]
            with np.errstate(all="ignore"):
                err = np.nanmax(np.abs(g - w) / np.maximum(1.0, np.abs(w)))
            print(f"coupled mode {mode} {k:10s}: bounded members {ok.sum()}, max |gpu - oracle| / max(1, |oracle|) = {err:.3e}, "
                  f"bit-equal {np.mean(g.view(np.uint64) == w.view(np.uint64)):.3f}")
