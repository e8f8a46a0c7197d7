#!/bin/bash
# round 4, session h: a step's independent light ops on two wavefronts (group_split_kernel) -- its test, the GPU tier, configs[3]'s share
# with and without it, the kernel table of the share
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_group.py -x -q -m gpu -s > gpurun_out/r4h_group_tests.log 2>&1 || { tail -40 gpurun_out/r4h_group_tests.log; exit 1; }
tail -3 gpurun_out/r4h_group_tests.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4h_tests.log 2>&1 || { tail -40 gpurun_out/r4h_tests.log; exit 1; }
tail -2 gpurun_out/r4h_tests.log
timeout -k 10 600 python3 scripts/run_configs3_share.py > gpurun_out/r4h_configs3_fast.json 2> gpurun_out/r4h_configs3_fast.err || { tail -5 gpurun_out/r4h_configs3_fast.err; exit 1; }
cut -c1-330 gpurun_out/r4h_configs3_fast.json
RSCM_LOCKSTEP_SPLIT=0 timeout -k 10 600 python3 scripts/run_configs3_share.py > gpurun_out/r4h_configs3_fast_nosplit.json 2> gpurun_out/r4h_configs3_fast_nosplit.err || { tail -5 gpurun_out/r4h_configs3_fast_nosplit.err; exit 1; }
cut -c1-330 gpurun_out/r4h_configs3_fast_nosplit.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r4h_c3trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 > "$ROOT/gpurun_out/r4h_c3trace.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r4h_c3trace.log"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r4h_c3trace > gpurun_out/r4h_c3trace_table.txt && cat gpurun_out/r4h_c3trace_table.txt
find gpurun_out/r4h_c3trace -name "*_kernel_trace.csv" -delete
