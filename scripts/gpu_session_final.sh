#!/bin/bash
# Round-end verification on the GPU box: multi-rank rehearsals first (their launchers must not have touched the GPU),
# then the GPU test tier, the bench line and the configs[3] share records.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
python -m pytest tests/test_multirank_gpu.py -m gpu_ranks -q > gpurun_out/final_ranks.log 2>&1 || { tail -20 gpurun_out/final_ranks.log; exit 1; }
tail -2 gpurun_out/final_ranks.log
RSCM_BENCH_BACKEND=gloo RSCM_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err || { tail -20 gpurun_out/bench_2rank.err; exit 1; }
python scripts/rehearse_rccl_one_rank.py > gpurun_out/rccl_one_rank.log 2> gpurun_out/rccl_one_rank.err || { tail -20 gpurun_out/rccl_one_rank.err; exit 1; }
tail -1 gpurun_out/rccl_one_rank.log | cut -c1-200
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu.log 2>&1 || { tail -30 gpurun_out/final_gpu.log; exit 1; }
tail -2 gpurun_out/final_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1 || { tail gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail gpurun_out/bench_final.err; exit 1; }
python scripts/run_configs3_share.py > gpurun_out/configs3_fast.json 2>/dev/null || exit 1
python scripts/run_configs3_share.py --exact > gpurun_out/configs3_exact.json 2>/dev/null || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_final.json"))
print("bench:", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k, v in d["extra"].items():
    keep = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms", "kernel_ms", "run_s", "launches", "device_ms_per_iteration", "member_years_per_s", "error")}
    print(" ", k, keep)
for f in ("configs3_fast", "configs3_exact"):
    c = json.load(open(f"gpurun_out/{f}.json"))
    print(f, c["run_s"], c["hbm_allocated_gib"], all(c["first_64_members_equal_a_64_member_run"].values()))
PY
