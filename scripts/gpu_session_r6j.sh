#!/bin/bash
# round 6, session j: six of the merged launch's ops BESIDE OceanCarbon on the helper stream -- graph tests, then the share in the three
# plans (fusion 1 beside, 6 merged on one stream, 5 round 5's) and the kernel trace of the default plan.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_links.py tests/test_gpu_window.py -x -q -m gpu > gpurun_out/r6j_group_tests.log 2>&1 || { tail -60 gpurun_out/r6j_group_tests.log; exit 1; }
tail -n 2 gpurun_out/r6j_group_tests.log
for f in 1 6 5; do
  timeout -k 10 600 python3 scripts/run_configs3_share.py --fusion $f > gpurun_out/r6j_share_fusion$f.json 2> gpurun_out/r6j_share_fusion$f.err || { tail -5 gpurun_out/r6j_share_fusion$f.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/r6j_share_fusion$f.json').read().strip().splitlines()[-1]); print('fusion $f:', round(d['run_s'],4), 's', round(d['ms_per_model_step']*1e3,1), 'us/step', d['launches_per_step'], 'launches/step', all(d['first_64_members_equal_a_64_member_run'].values()))"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6j_share_trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 20 --no-anchor > "$ROOT/gpurun_out/r6j_share_traced.json" 2> "$ROOT/gpurun_out/r6j_share_traced.err" || { tail -5 "$ROOT/gpurun_out/r6j_share_traced.err"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r6j_share_trace 100000 > gpurun_out/r6j_share_trace_table.txt; head -6 gpurun_out/r6j_share_trace_table.txt; tail -14 gpurun_out/r6j_share_trace_table.txt
find gpurun_out/r6j_share_trace -name '*.csv' -size +2M -delete
