#!/bin/bash
# round 5, session i: the whole GPU tier on the final tree, the multi-rank rehearsals, PMC passes of the headline kernel (one launch)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5i_tests.log 2>&1 || { tail -40 gpurun_out/r5i_tests.log; exit 1; }
tail -n 2 gpurun_out/r5i_tests.log
timeout -k 10 900 python -m pytest tests/test_multirank_gpu.py -x -q -m gpu_ranks > gpurun_out/r5i_ranks.log 2>&1 || { tail -40 gpurun_out/r5i_ranks.log; exit 1; }
tail -n 2 gpurun_out/r5i_ranks.log
bash scripts/gpu_profile.sh r5_exact_1e5 100000 0 0 > gpurun_out/r5i_prof.log 2>&1 || { tail -20 gpurun_out/r5i_prof.log; exit 1; }
python3 scripts/summarize_profile.py r5_exact_1e5 gpurun_out/r5_exact_1e5.txt two_layer_kernel | tail -8
bash scripts/gpu_profile.sh r5_exact_65536 65536 0 0 > gpurun_out/r5i_prof2.log 2>&1 || { tail -20 gpurun_out/r5i_prof2.log; exit 1; }
python3 scripts/summarize_profile.py r5_exact_65536 gpurun_out/r5_exact_65536.txt two_layer_kernel | tail -8
find gpurun_out/prof_r5_exact_* -name '*_kernel_trace.csv' -delete
