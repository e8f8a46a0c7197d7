#!/bin/bash
# round 4, session a: RSCM_MODE_FAST of the coupled chain -- parity (fused kernel and linked bodies, both modes), deviation from
# the oracle per series, then the kernel trace and the PMC passes of coupled_fast_kernel at 1e6 members
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_links.py tests/test_gpu_group.py -x -q -m gpu -k "coupled or fused or linked" > gpurun_out/r4a_tests.log 2>&1 || { tail -30 gpurun_out/r4a_tests.log; exit 1; }
tail -3 gpurun_out/r4a_tests.log
timeout -k 10 600 python scripts/fast_mode_error.py > gpurun_out/r4a_fast_mode_error.log 2>&1 || { tail -20 gpurun_out/r4a_fast_mode_error.log; exit 1; }
grep coupled gpurun_out/r4a_fast_mode_error.log
timeout -k 10 900 bash scripts/gpu_profile.sh r4_coupled_fast_1e6 1000000 1 1 || exit 1
python3 scripts/summarize_profile.py r4_coupled_fast_1e6 gpurun_out/r4_coupled_fast_1e6.txt coupled_fast_kernel && tail -40 gpurun_out/r4_coupled_fast_1e6.txt
