#!/bin/bash
# round 4, session f: the tree as committed -- smoke, the GPU tier, bench.py as the driver runs it (N = 1), the bench under the kernel trace
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4f_smoke.log 2>&1 || { tail -20 gpurun_out/r4f_smoke.log; exit 1; }
tail -2 gpurun_out/r4f_smoke.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4f_tests.log 2>&1 || { tail -40 gpurun_out/r4f_tests.log; exit 1; }
tail -2 gpurun_out/r4f_tests.log
timeout -k 10 900 python bench.py > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err || { tail -20 gpurun_out/r4f_bench.err; exit 1; }
python3 -c "
import json; d = json.load(open('gpurun_out/r4f_bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'steps')}, 'roofline', round(d['roofline']['frac'], 4), d['collective'], d['per_rank']['kernel_ms'])
print({k: v.get('kernel_ms', v.get('run_s', v.get('ms'))) for k, v in d['extra'].items() if isinstance(v, dict)})"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r4f_bench_trace" -- python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline > "$ROOT/gpurun_out/r4f_bench_traced.json" 2> "$ROOT/gpurun_out/r4f_bench_traced.err" || { tail -5 "$ROOT/gpurun_out/r4f_bench_traced.err"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r4f_bench_trace 1000 > gpurun_out/r4f_bench_trace_table.txt; head -8 gpurun_out/r4f_bench_trace_table.txt
python3 -c "
import json; d = json.load(open('gpurun_out/r4f_bench_traced.json')); print('traced run:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
find gpurun_out/r4f_bench_trace -name '*_kernel_trace.csv' -delete
