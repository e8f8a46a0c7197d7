#!/bin/bash
# round 5, session a: the scale_* extras (bench.py on all ranks) -- the new GPU test, then full-size rehearsals on the one GPU:
# two ranks over gloo (both on device 0), one rank inside a one-rank RCCL group, and the plain N = 1 line.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5a_smoke.log 2>&1 || { tail -20 gpurun_out/r5a_smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_multirank_gpu.py -x -q -m gpu > gpurun_out/r5a_tests.log 2>&1 || { tail -60 gpurun_out/r5a_tests.log; exit 1; }
tail -2 gpurun_out/r5a_tests.log
RSCM_BENCH_BACKEND=gloo RSCM_BENCH_DEVICE=0 timeout -k 10 900 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r5a_bench_2ranks_gloo.json 2> gpurun_out/r5a_bench_2ranks_gloo.err || { tail -20 gpurun_out/r5a_bench_2ranks_gloo.err; exit 1; }
RSCM_BENCH_FORCE_DIST=1 timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --scale-only --no-cpu-baseline > gpurun_out/r5a_bench_1rank_rccl.json 2> gpurun_out/r5a_bench_1rank_rccl.err || { tail -20 gpurun_out/r5a_bench_1rank_rccl.err; exit 1; }
timeout -k 10 900 python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err || { tail -20 gpurun_out/r5a_bench.err; exit 1; }
python3 - <<'P'
import json
for f in ("r5a_bench_2ranks_gloo", "r5a_bench_1rank_rccl", "r5a_bench"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["collective"].get("backend"))
    for k, v in d["extra"].items():
        if k.startswith("scale_"):
            print("  ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, str, list))})
    print("  cpu:", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("threads_used"))
P
