#!/bin/bash
# round 4, session b: the whole GPU tier after the round's changes so far, then the kernel trace of configs[3]'s share (50 years)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4b_tests.log 2>&1 || { tail -40 gpurun_out/r4b_tests.log; exit 1; }
tail -3 gpurun_out/r4b_tests.log
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r4b_c3trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 > "$ROOT/gpurun_out/r4b_c3trace.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r4b_c3trace.log"; exit 1; }
cd "$ROOT"
tail -1 gpurun_out/r4b_c3trace.log | cut -c1-400
python3 scripts/trace_table.py gpurun_out/r4b_c3trace > gpurun_out/r4b_c3trace_table.txt && cat gpurun_out/r4b_c3trace_table.txt
find gpurun_out/r4b_c3trace -name "*_kernel_trace.csv" -delete
