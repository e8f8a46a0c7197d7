#!/bin/bash
# round 5, session w: the two-wavefront cut of a one-step segment in PHASES (levels) -- graph tests, configs[3] share, kernel table
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_links.py tests/test_gpu_window.py -x -q -m gpu > gpurun_out/r5w_tests.log 2>&1 || { tail -60 gpurun_out/r5w_tests.log; exit 1; }
tail -n 2 gpurun_out/r5w_tests.log
for f in 1 4; do
  timeout -k 10 600 python scripts/run_configs3_share.py --fusion $f > gpurun_out/r5w_share_fusion$f.json 2> gpurun_out/r5w_share_fusion$f.err || { tail -5 gpurun_out/r5w_share_fusion$f.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r5w_share_fusion$f.json')); print('fusion $f:', round(d['run_s'],4), 's', round(d['ms_per_model_step']*1e3,1), 'us per step', all(d['first_64_members_equal_a_64_member_run'].values()))"
done
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r5w_share_trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 > "$ROOT/gpurun_out/r5w_share_traced.json" 2> "$ROOT/gpurun_out/r5w_share_traced.err" || { tail -5 "$ROOT/gpurun_out/r5w_share_traced.err"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r5w_share_trace 20 > gpurun_out/r5w_share_trace_table.txt; head -12 gpurun_out/r5w_share_trace_table.txt
find gpurun_out/r5w_share_trace -name '*_kernel_trace.csv' -delete
