#!/bin/bash
# round 4, session g: ClimateUDEB's base LAMCALC solve as member constants -- GPU tier, ClimateUDEB whole-axis and one-step timings,
# configs[3]'s share, then the bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4g_tests.log 2>&1 || { tail -40 gpurun_out/r4g_tests.log; exit 1; }
tail -2 gpurun_out/r4g_tests.log
timeout -k 10 300 python scripts/bench_udeb.py 65536 100000 125000 2>&1 | tee gpurun_out/r4g_udeb.log || exit 1
timeout -k 10 300 python scripts/bench_udeb_steps.py 65536 125000 2>&1 | tee -a gpurun_out/r4g_udeb.log || exit 1
timeout -k 10 600 python3 scripts/run_configs3_share.py > gpurun_out/r4g_configs3_fast.json 2> gpurun_out/r4g_configs3_fast.err || { tail -5 gpurun_out/r4g_configs3_fast.err; exit 1; }
cut -c1-330 gpurun_out/r4g_configs3_fast.json
timeout -k 10 900 python bench.py > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err || { tail -20 gpurun_out/r4g_bench.err; exit 1; }
python3 -c "
import json; d = json.load(open('gpurun_out/r4g_bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step')}, round(d['roofline']['frac'], 4))
print({k: v.get('kernel_ms', v.get('run_s', v.get('ms', v.get('device_ms_per_iteration')))) for k, v in d['extra'].items() if isinstance(v, dict)})"
