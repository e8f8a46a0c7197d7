#!/bin/bash
# round 6, session p: the final tree once more, as the driver will run it -- smoke, the whole GPU tier, bench.py (N = 1)
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6p_smoke.log 2>&1 || { tail -20 gpurun_out/r6p_smoke.log; exit 1; }
tail -n 1 gpurun_out/r6p_smoke.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6p_tests.log 2>&1 || { tail -40 gpurun_out/r6p_tests.log; exit 1; }
tail -n 2 gpurun_out/r6p_tests.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6p_bench_details.json > gpurun_out/r6p_bench.json 2> gpurun_out/r6p_bench.err || { tail -20 gpurun_out/r6p_bench.err; exit 1; }
wc -c gpurun_out/r6p_bench.json
python3 -c "
import json
d=json.loads(open('gpurun_out/r6p_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], {k:v for k,v in d['extra'].items() if 'error' in v})
print(d['extra']['configs3_share_125000x9000_fast'], d['extra']['magicc_chain_1e5_fast'])"
