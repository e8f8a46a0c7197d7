#!/bin/bash
# round 5, session k: the final tree -- smoke, the whole GPU tier, bench.py as the driver runs it (N = 1), two ranks on the one GPU over gloo at full size
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5k_smoke.log 2>&1 || { tail -20 gpurun_out/r5k_smoke.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5k_tests.log 2>&1 || { tail -40 gpurun_out/r5k_tests.log; exit 1; }
tail -n 2 gpurun_out/r5k_tests.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5k_bench.json 2> gpurun_out/r5k_bench.err || { tail -20 gpurun_out/r5k_bench.err; exit 1; }
RSCM_BENCH_BACKEND=gloo RSCM_BENCH_DEVICE=0 timeout -k 10 900 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r5k_bench_2ranks_gloo.json 2> gpurun_out/r5k_bench_2ranks_gloo.err || { tail -20 gpurun_out/r5k_bench_2ranks_gloo.err; exit 1; }
python3 - <<'P'
import json
for f in ("r5k_bench", "r5k_bench_2ranks_gloo"):
    rows = [x for x in open(f"gpurun_out/{f}.json").read().splitlines() if x.strip()]
    d = json.loads(rows[-1])
    print(f, len(rows), "line(s):", d["value"], d["ms_per_step"], d["roofline"]["frac"], (d["roofline_fp64_valu"].get("parallelism_bound") or {}).get("achieved_frac"))
    print("  errors:", [k for k, v in d["extra"].items() if isinstance(v, dict) and "error" in v])
    for k, v in d["extra"].items():
        if k.startswith("scale_"):
            print("  ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, str, list))})
P
