#!/bin/bash
# round 5, session o: ClimateUDEB's southern column straight to LDS on resume -- tests, one-step launches, configs[3] share
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_udeb.py tests/test_gpu_window.py -x -q -m gpu > gpurun_out/r5o_tests.log 2>&1 || { tail -40 gpurun_out/r5o_tests.log; exit 1; }
tail -n 2 gpurun_out/r5o_tests.log
timeout -k 10 600 python scripts/bench_udeb_steps.py 65536 125000 2>&1 | tee gpurun_out/r5o_udeb_steps.log
timeout -k 10 600 python scripts/run_configs3_share.py > gpurun_out/r5o_share.json 2> gpurun_out/r5o_share.err || { tail -5 gpurun_out/r5o_share.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r5o_share.json')); print('share:', round(d['run_s'],4), 's', round(d['ms_per_model_step']*1e3,1), 'us per step', all(d['first_64_members_equal_a_64_member_run'].values()))"
timeout -k 10 300 python scripts/bench_udeb.py 100000 2>&1 | tail -3 | tee gpurun_out/r5o_udeb_1e5.log
