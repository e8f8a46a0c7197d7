"""A/B on one card: the fused one-step launches of the windowed MAGICC graph (125 000 members, monthly steps, 40 years)
with their op tables by value in the kernel arguments (rscm_gpu_set_lockstep_fusion(1)) and through device memory (3).
Measured: 0.173 vs 0.174-0.176 s."""
import os, sys, time, json, io, contextlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rscm_amd import _lib as L
from scripts.bench_magicc_chain import build_chain
lib = L.load()
def once(mode, members=125000, years=40):
    L.check(lib.rscm_gpu_set_lockstep_fusion(mode))
    m = build_chain(members, years, "topological", steps_per_year=12, series_window=16, output_stride=12)
    m.set_mode(1) if hasattr(m, "set_mode") else None
    t0 = time.perf_counter(); m.run(); dt = time.perf_counter() - t0
    m.close()
    return dt
once(1, 1000, 2)
for rep in range(3):
    for mode in (1, 3):
        print("mode", mode, "%.3f s" % once(mode), flush=True)
