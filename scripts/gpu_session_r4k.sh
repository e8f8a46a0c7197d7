#!/bin/bash
# round 4, session k: the tree with the member split -- GPU tier, bench.py as the driver runs it, the bench under the kernel trace
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4k_smoke.log 2>&1 || { tail -20 gpurun_out/r4k_smoke.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4k_tests.log 2>&1 || { tail -40 gpurun_out/r4k_tests.log; exit 1; }
tail -2 gpurun_out/r4k_tests.log
timeout -k 10 900 python bench.py > gpurun_out/r4k_bench.json 2> gpurun_out/r4k_bench.err || { tail -20 gpurun_out/r4k_bench.err; exit 1; }
python3 -c "
import json; d = json.load(open('gpurun_out/r4k_bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'steps')}, {k: d['roofline'][k] for k in ('frac', 'kernel_ms', 'launches_per_pass')})
print({k: round(v.get('kernel_ms', v.get('run_s', v.get('ms', v.get('device_ms_per_iteration', 0)))), 3) for k, v in d['extra'].items() if isinstance(v, dict)})"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r4k_bench_trace" -- python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline > "$ROOT/gpurun_out/r4k_bench_traced.json" 2> "$ROOT/gpurun_out/r4k_bench_traced.err" || { tail -5 "$ROOT/gpurun_out/r4k_bench_traced.err"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r4k_bench_trace 1000 > gpurun_out/r4k_bench_trace_table.txt; head -12 gpurun_out/r4k_bench_trace_table.txt
python3 -c "
import json; d = json.load(open('gpurun_out/r4k_bench_traced.json')); print('traced run:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches_per_pass'])"
find gpurun_out/r4k_bench_trace -name '*_kernel_trace.csv' -delete
