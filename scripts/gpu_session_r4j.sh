#!/bin/bash
# round 4, session j: whole-axis runs as two member blocks on two streams in chunks of model steps (rscm_gpu.cpp, plan_member_split):
# the GPU tier, then the headline with and without it
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4j_tests.log 2>&1 || { tail -40 gpurun_out/r4j_tests.log; exit 1; }
tail -2 gpurun_out/r4j_tests.log
for v in 1 0; do
  RSCM_SPLIT_RUNS=$v timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r4j_bench_split$v.json 2> gpurun_out/r4j_bench_split$v.err || { tail -20 gpurun_out/r4j_bench_split$v.err; exit 1; }
  python3 -c "
import json; d = json.load(open('gpurun_out/r4j_bench_split$v.json'))
print('RSCM_SPLIT_RUNS=$v', {k: d[k] for k in ('value', 'ms_per_step')}, round(d['roofline']['frac'], 4), round(d['roofline']['kernel_ms'], 3))
print({k: round(v.get('kernel_ms', v.get('run_s', v.get('ms', v.get('device_ms_per_iteration', 0)))), 3) for k, v in d['extra'].items() if isinstance(v, dict)})"
done
