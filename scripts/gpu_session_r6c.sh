#!/bin/bash
# round 6, session c: merged launches (a step's last light segment with the next step's first: 3 launches per step) and the prefetch
# pass of the one-step fused launches -- the graph tests first (merged == unmerged == round 5's plan == unfused, bit for bit), then the
# whole GPU tier, then configs[3]'s share timed in the three plans (fusion 1 / 5 / 6) and under the kernel trace.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_window.py -x -q -m gpu > gpurun_out/r6c_group_tests.log 2>&1 || { tail -60 gpurun_out/r6c_group_tests.log; exit 1; }
tail -n 2 gpurun_out/r6c_group_tests.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6c_tests.log 2>&1 || { tail -40 gpurun_out/r6c_tests.log; exit 1; }
tail -n 2 gpurun_out/r6c_tests.log
for f in 1 5 6; do
  timeout -k 10 600 python3 scripts/run_configs3_share.py --fusion $f > gpurun_out/r6c_share_fusion$f.json 2> gpurun_out/r6c_share_fusion$f.err || { tail -5 gpurun_out/r6c_share_fusion$f.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/r6c_share_fusion$f.json').read().strip().splitlines()[-1]); print('fusion $f:', round(d['run_s'],4), 's', round(d['ms_per_model_step']*1e3,1), 'us/step', d['launches_per_step'], 'launches/step', all(d['first_64_members_equal_a_64_member_run'].values()))"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6c_share_trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 --no-anchor > "$ROOT/gpurun_out/r6c_share_traced.json" 2> "$ROOT/gpurun_out/r6c_share_traced.err" || { tail -5 "$ROOT/gpurun_out/r6c_share_traced.err"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r6c_share_trace 100000 > gpurun_out/r6c_share_trace_table.txt; head -12 gpurun_out/r6c_share_trace_table.txt; tail -12 gpurun_out/r6c_share_trace_table.txt
find gpurun_out/r6c_share_trace -name '*.csv' -size +2M -delete
