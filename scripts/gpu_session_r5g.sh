#!/bin/bash
# round 5, session g: the tree after the queue experiment -- smoke, the whole GPU tier, bench.py as the driver runs it, the bench under the kernel trace,
# the PMC passes of the headline kernel (exact, 1e5), and the one-rank RCCL rehearsal of the scale extras
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5g_smoke.log 2>&1 || { tail -20 gpurun_out/r5g_smoke.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5g_tests.log 2>&1 || { tail -40 gpurun_out/r5g_tests.log; exit 1; }
tail -n 2 gpurun_out/r5g_tests.log
timeout -k 10 900 python bench.py > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err || { tail -20 gpurun_out/r5g_bench.err; exit 1; }
python3 -c "
import json; d = json.load(open('gpurun_out/r5g_bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'steps')}, {k: d['roofline'][k] for k in ('frac', 'kernel_ms', 'launches_per_pass')})
print({k: round(v.get('kernel_ms', v.get('run_s', v.get('ms', v.get('device_ms_per_iteration', v.get('wall_s', 0))))), 3) for k, v in d['extra'].items() if isinstance(v, dict)})
print([k for k, v in d['extra'].items() if isinstance(v, dict) and 'error' in v])"
RSCM_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --scale-only --no-cpu-baseline > gpurun_out/r5g_bench_1rank_rccl.json 2> gpurun_out/r5g_bench_1rank_rccl.err || { tail -20 gpurun_out/r5g_bench_1rank_rccl.err; exit 1; }
python3 -c "
import json; rows=[x for x in open('gpurun_out/r5g_bench_1rank_rccl.json').read().splitlines() if x.strip()]; print('stdout lines under a one-rank RCCL group:', len(rows)); d=json.loads(rows[-1]); print(d['collective']['backend'], {k: v.get('exchange_ms_per_iteration') for k, v in d['extra'].items() if 'calibrate' in k})"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r5g_bench_trace" -- python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline > "$ROOT/gpurun_out/r5g_bench_traced.json" 2> "$ROOT/gpurun_out/r5g_bench_traced.err" || { tail -5 "$ROOT/gpurun_out/r5g_bench_traced.err"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r5g_bench_trace 1000 > gpurun_out/r5g_bench_trace_table.txt; head -12 gpurun_out/r5g_bench_trace_table.txt
find gpurun_out/r5g_bench_trace -name '*_kernel_trace.csv' -delete
bash scripts/gpu_profile.sh r5_exact_1e5 100000 0 0 > gpurun_out/r5g_prof.log 2>&1 || { tail -20 gpurun_out/r5g_prof.log; exit 1; }
python3 scripts/summarize_profile.py r5_exact_1e5 gpurun_out/r5_exact_1e5.txt two_layer_kernel | tail -8
find gpurun_out/prof_r5_exact_1e5 -name '*_kernel_trace.csv' -delete
