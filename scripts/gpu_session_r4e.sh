#!/bin/bash
# round 4, session e: the new member-constant / CarbonCycle-mode tests, the whole GPU tier, the any-count ClimateUDEB kernel's speed
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ghg.py tests/test_gpu_carbon.py -x -q -m gpu > gpurun_out/r4e_new_tests.log 2>&1 || { tail -40 gpurun_out/r4e_new_tests.log; exit 1; }
tail -2 gpurun_out/r4e_new_tests.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4e_tests.log 2>&1 || { tail -40 gpurun_out/r4e_tests.log; exit 1; }
tail -2 gpurun_out/r4e_tests.log
timeout -k 10 600 python - <<'PY' 2>&1 | tee gpurun_out/r4e_udeb_any.log
import time, numpy as np, sys
sys.path.insert(0, ".")
import rscm_amd
from rscm_amd import _lib
t = np.arange(1750.0, 2501.0)
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0))
for nl, members in ((50, 65536), (49, 65536), (25, 65536), (30, 65536), (100, 65536)):
    lo = np.array(_lib.UD_DEFAULTS, dtype=float); lo[_lib.UD_PARAM_NAMES.index("n_layers")] = nl
    hi = lo.copy()
    for name, (a, b) in dict(ecs=(2.0, 5.0), kappa=(0.5, 1.5)).items():
        j = _lib.UD_PARAM_NAMES.index(name); lo[j], hi[j] = a, b
    with rscm_amd.Ensemble(rscm_amd.KIND_UDEB, members, np.append(t, 2501.0)) as e:
        e.sample_lhs(1, lo, hi); e.set_forcing(F)
        for v in (1, 2, 3, 4): e.set_initial(v, 0.0)
        e.run(); e.rewind(); e.sync()
        t0 = time.perf_counter(); e.run(); dt = time.perf_counter() - t0
        print(f"ClimateUDEB n_layers={nl:3d} members={members}: {dt*1e3:8.1f} ms per 750 years ({'register-resident' if nl in (20,30,40,50) else 'any-count'} kernel), failed members {int(e.status().astype(bool).sum())}")
PY
