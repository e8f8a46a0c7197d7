#!/bin/bash
# round 6, session i: the compact line with more than one rank, at full size on the one GPU -- two ranks over gloo (bench.py as its own
# launcher), one rank inside an RCCL group -- and ten fresh processes of the headline (run-to-run spread).
set -o pipefail
mkdir -p gpurun_out
RSCM_BENCH_BACKEND=gloo RSCM_BENCH_DEVICE=0 timeout -k 10 900 python bench.py --gpus 2 --steps 20 --warmup 5 --details gpurun_out/r6i_bench_2ranks_details.json > gpurun_out/r6i_bench_2ranks_gloo.json 2> gpurun_out/r6i_bench_2ranks_gloo.err || { tail -20 gpurun_out/r6i_bench_2ranks_gloo.err; exit 1; }
RSCM_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --scale-only --no-cpu-baseline --details gpurun_out/r6i_bench_1rank_rccl_details.json > gpurun_out/r6i_bench_1rank_rccl.json 2> gpurun_out/r6i_bench_1rank_rccl.err || { tail -20 gpurun_out/r6i_bench_1rank_rccl.err; exit 1; }
wc -c gpurun_out/r6i_bench_2ranks_gloo.json gpurun_out/r6i_bench_1rank_rccl.json
python3 - <<'P'
import json
for f in ("r6i_bench_2ranks_gloo", "r6i_bench_1rank_rccl"):
    rows = [x for x in open(f"gpurun_out/{f}.json").read().splitlines() if x.strip()]
    d = json.loads(rows[-1])
    print(f, len(rows), "line(s),", len(rows[-1]), "bytes:", d["n_gpus"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["collective"], d["per_rank"])
    print("   ", {k: v for k, v in d["extra"].items()})
P
bash scripts/bench_variance.sh > gpurun_out/r6i_bench_variance.txt 2>&1 || { tail gpurun_out/r6i_bench_variance.txt; exit 1; }
cat gpurun_out/r6i_bench_variance.txt
