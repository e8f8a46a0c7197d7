#!/bin/bash
# round 4, session c: member constants formed once per parameter set (GhgForcing, TerrestrialCarbon), op-table lines requested up front,
# a 96-row window: the GPU tier, the kernel trace of configs[3]'s share (50 years), then the share at full length in both modes
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4c_tests.log 2>&1 || { tail -40 gpurun_out/r4c_tests.log; exit 1; }
tail -3 gpurun_out/r4c_tests.log
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r4c_c3trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 > "$ROOT/gpurun_out/r4c_c3trace.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r4c_c3trace.log"; exit 1; }
cd "$ROOT"
tail -1 gpurun_out/r4c_c3trace.log | cut -c1-300
python3 scripts/trace_table.py gpurun_out/r4c_c3trace > gpurun_out/r4c_c3trace_table.txt && cat gpurun_out/r4c_c3trace_table.txt
find gpurun_out/r4c_c3trace -name "*_kernel_trace.csv" -delete
timeout -k 10 600 python3 scripts/run_configs3_share.py > gpurun_out/r4c_configs3_fast.json 2> gpurun_out/r4c_configs3_fast.err || { tail -5 gpurun_out/r4c_configs3_fast.err; exit 1; }
cut -c1-420 gpurun_out/r4c_configs3_fast.json
timeout -k 10 600 python3 scripts/run_configs3_share.py --window 16 > gpurun_out/r4c_configs3_fast_w16.json 2> gpurun_out/r4c_configs3_fast_w16.err || { tail -5 gpurun_out/r4c_configs3_fast_w16.err; exit 1; }
cut -c1-420 gpurun_out/r4c_configs3_fast_w16.json
