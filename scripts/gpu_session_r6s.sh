#!/bin/bash
# round 6, session s: the headline's evidence refreshed on the final tree -- bench.py (headline only) under the kernel trace, and the PMC
# passes of two_layer_kernel as ONE launch over the axis at 1e5 members (what profiles/traffic.json's headline entry points at).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6s_bench_trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-extra --no-cpu-baseline --details "" > "$ROOT/gpurun_out/r6s_bench_traced.json" 2> "$ROOT/gpurun_out/r6s_bench_traced.err" || { tail -5 "$ROOT/gpurun_out/r6s_bench_traced.err"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r6s_bench_trace 1000 > gpurun_out/r6_bench_kernel_trace.txt; head -8 gpurun_out/r6_bench_kernel_trace.txt
cat gpurun_out/r6s_bench_traced.json | cut -c1-400
find gpurun_out/r6s_bench_trace -name '*_kernel_trace.csv' -delete
bash scripts/gpu_profile.sh r6_exact_1e5 100000 0 0 > gpurun_out/r6s_prof.log 2>&1 || { tail -20 gpurun_out/r6s_prof.log; exit 1; }
python3 scripts/summarize_profile.py r6_exact_1e5 gpurun_out/r6_exact_1e5.txt two_layer_kernel | tail -8
find gpurun_out/prof_r6_exact_1e5 -name '*_kernel_trace.csv' -delete
