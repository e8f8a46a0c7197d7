"""Where the whole-graph launch (csrc/graph.hip) spends its time: configs[3]'s MAGICC graph for a number of monthly
steps with the in-kernel cycle stamps on (rscm_gpu_graph_stamps), then the same run timed with the stamps off and with
the whole-graph launch off (fusion mode 1, the default: four launches per step).
    python scripts/graph_stamps.py [members] [years]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rscm_amd import _lib as L  # noqa: E402
from scripts.bench_magicc_chain import build_chain  # noqa: E402

KINDS = {0: "TwoLayer", 2: "ClimateUDEB", 3: "GhgForcing", 4: "OzoneForcing", 5: "AerosolDirect", 6: "AerosolIndirect", 7: "CH4Chemistry",
         8: "N2OChemistry", 9: "CO2Budget", 10: "TerrestrialCarbon", 11: "OceanCarbon", 13: "FourBoxOHU", 14: "OSPP", 15: "CarbonCycle",
         16: "CO2ERF", 17: "Aggregate (x3: Sum of 8 + 2 grid transforms)", 28: "ClimateUDEB: step start to sub-step loop",
         29: "ClimateUDEB: 12 sub-steps", 31: "ClimateUDEB begin/end (per launch)"}
KINDS[2] = "ClimateUDEB: end of step (outputs)"

members = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
years = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = L.load()
out = {"members": members, "monthly_steps": years * 12}
for label, fusion, stamps in (("whole_graph_stamped", 4, 1), ("whole_graph", 4, 0), ("four_launches_per_step", 1, 0)):
    L.check(lib.rscm_gpu_set_lockstep_fusion(fusion))
    model = build_chain(members, years, "topological", steps_per_year=12, series_window=16, output_stride=12)
    model.set_mode(L.MODE_FAST)
    from rscm_amd.ensemble import run_lockstep
    run_lockstep([model.ensembles[name] for name in model._order], 24, sync=True)   # warm-up: first launches, allocations
    model.time_index = 24
    buf = (C.c_uint64 * 32)()
    L.check(lib.rscm_gpu_graph_stamps(0, stamps, buf))
    t0 = time.perf_counter()
    model.run()
    for e in model.ensembles.values():
        e.sync()
        break
    wall = time.perf_counter() - t0
    steps = years * 12 - 24
    out[label] = {"wall_s": wall, "us_per_step": wall / steps * 1e6}
    if stamps:
        L.check(lib.rscm_gpu_graph_stamps(0, 0, buf))
        cyc = {k: int(buf[k]) for k in range(32) if buf[k]}
        total = sum(cyc.values())
        waves = (members + 63) // 64
        out[label]["cycles_per_wavefront_step"] = {KINDS.get(k, str(k)): round(v / waves / steps) for k, v in sorted(cyc.items(), key=lambda kv: -kv[1])}
        out[label]["share"] = {KINDS.get(k, str(k)): round(v / total, 4) for k, v in sorted(cyc.items(), key=lambda kv: -kv[1])}
    model.close()
print(json.dumps(out, indent=1))
