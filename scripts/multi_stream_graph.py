"""configs[3]'s MAGICC graph with the members cut into K blocks, each block a graph of its own on its own HIP stream,
all stepped from one host thread in turns of a window chunk: kernels of different blocks overlap on the GPU.
    python scripts/multi_stream_graph.py [members] [years] [fusion mode] [K ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rscm_amd import _lib as L  # noqa: E402
from rscm_amd.ensemble import run_lockstep  # noqa: E402
from scripts.bench_magicc_chain import build_chain  # noqa: E402

members = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
years = int(sys.argv[2]) if len(sys.argv) > 2 else 50
fusion = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = L.load()
L.check(lib.rscm_gpu_set_lockstep_fusion(fusion))
for K in [int(a) for a in sys.argv[4:]] or [1, 2, 4, 8]:
    sizes = [members // K + (1 if b < members % K else 0) for b in range(K)]
    models = [build_chain(n, years, "topological", steps_per_year=12, series_window=16, output_stride=12) for n in sizes]
    for m in models:
        m.set_mode(L.MODE_FAST)
    lists = [[m.ensembles[name] for name in m._order] for m in models]
    last = years * 12
    chunk = 14
    for lst in lists:
        run_lockstep(lst, 28, sync=True)
    t0 = time.perf_counter()
    n = 28
    while n < last:
        n = min(n + chunk, last)
        for lst in lists:
            run_lockstep(lst, n, sync=False)
    for lst in lists:
        lst[0].sync()
    wall = time.perf_counter() - t0
    print(f"{members} members in {K} block(s) on {K} stream(s), fusion mode {fusion}: {wall / (last - 28) * 1e6:.1f} us per monthly step", flush=True)
    for m in models:
        m.time_index = last
        m.close()
