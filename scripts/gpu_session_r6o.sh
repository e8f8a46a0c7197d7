#!/bin/bash
# round 6, session o (experiment): the per-sequence cut kernel with a prefetch pass in front of the bodies (RSCM_EXPERIMENT_PREFETCH=1) against without
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_group.py -x -q -m gpu -k merged > gpurun_out/r6o_tests.log 2>&1 || { tail -30 gpurun_out/r6o_tests.log; exit 1; }
RSCM_EXPERIMENT_PREFETCH=1 timeout -k 10 600 python -m pytest tests/test_gpu_group.py -x -q -m gpu -k merged > gpurun_out/r6o_tests_prefetch.log 2>&1 || { tail -30 gpurun_out/r6o_tests_prefetch.log; exit 1; }
tail -n 1 gpurun_out/r6o_tests.log gpurun_out/r6o_tests_prefetch.log
for p in 0 1 0 1; do
  RSCM_EXPERIMENT_PREFETCH=$p timeout -k 10 600 python3 scripts/run_configs3_share.py --years 300 > gpurun_out/r6o_share_p$p.json 2> gpurun_out/r6o_share_p$p.err || { tail -5 gpurun_out/r6o_share_p$p.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/r6o_share_p$p.json').read().strip().splitlines()[-1]); print('prefetch $p:', round(d['run_s'],4), 's', round(d['ms_per_model_step']*1e3,1), 'us/step', all(d['first_64_members_equal_a_64_member_run'].values()))"
done
