#!/bin/bash
# round 5, session d: the work-queue launch -- parity first (small timeout: a queue that does not drain must not hold the box), then the sweep
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cut_into_member_blocks or failed_chunk_launch" > gpurun_out/r5d_tests.log 2>&1 || { tail -60 gpurun_out/r5d_tests.log; exit 1; }
tail -2 gpurun_out/r5d_tests.log
timeout -k 10 300 python -m pytest tests/test_c_caller.py -x -q -m gpu > gpurun_out/r5d_c_tests.log 2>&1 || { tail -60 gpurun_out/r5d_c_tests.log; exit 1; }
tail -2 gpurun_out/r5d_c_tests.log
timeout -k 10 900 bash scripts/sweep_queue.sh gpurun_out/r5d_sweep_queue.txt
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
bash scripts/gpu_profile.sh r5_udeb_65536 65536 0 2 > gpurun_out/r5d_prof1.log 2>&1 || { tail -20 gpurun_out/r5d_prof1.log; exit 1; }
UDEB_LAYERS=49 bash scripts/gpu_profile.sh r5_udeb_49_65536 65536 0 2 > gpurun_out/r5d_prof2.log 2>&1 || { tail -20 gpurun_out/r5d_prof2.log; exit 1; }
UDEB_LAYERS=65 bash scripts/gpu_profile.sh r5_udeb_hbm65_65536 65536 0 2 > gpurun_out/r5d_prof3.log 2>&1 || { tail -20 gpurun_out/r5d_prof3.log; exit 1; }
cd "$ROOT"
python3 scripts/summarize_profile.py r5_udeb_65536 gpurun_out/r5_udeb_65536.txt udeb_kernel | tail -8
python3 scripts/summarize_profile.py r5_udeb_49_65536 gpurun_out/r5_udeb_any_49.txt udeb_kernel | tail -8
python3 scripts/summarize_profile.py r5_udeb_hbm65_65536 gpurun_out/r5_udeb_hbm_65.txt udeb_any_kernel | tail -8
find gpurun_out/prof_r5_udeb* -name '*_kernel_trace.csv' -delete
du -sh gpurun_out/prof_r5_udeb*
