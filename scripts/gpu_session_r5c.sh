#!/bin/bash
# round 5, session c: the compiled C callers, the failed-launch path, then PMC profiles of ClimateUDEB (50 layers fixed, 49 at run time, 65 in HBM)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_c_caller.py "tests/test_gpu_parity.py::test_a_failed_chunk_launch_joins_the_streams_and_leaves_the_run_undone" -x -q -m gpu > gpurun_out/r5c_tests.log 2>&1 || { tail -60 gpurun_out/r5c_tests.log; exit 1; }
tail -2 gpurun_out/r5c_tests.log
bash scripts/gpu_profile.sh r5_udeb_65536 65536 0 2 > gpurun_out/r5c_prof1.log 2>&1 || { tail -20 gpurun_out/r5c_prof1.log; exit 1; }
UDEB_LAYERS=49 bash scripts/gpu_profile.sh r5_udeb_49_65536 65536 0 2 > gpurun_out/r5c_prof2.log 2>&1 || { tail -20 gpurun_out/r5c_prof2.log; exit 1; }
UDEB_LAYERS=65 bash scripts/gpu_profile.sh r5_udeb_hbm65_65536 65536 0 2 > gpurun_out/r5c_prof3.log 2>&1 || { tail -20 gpurun_out/r5c_prof3.log; exit 1; }
cd "$ROOT"
python3 scripts/summarize_profile.py r5_udeb_65536 gpurun_out/r5_udeb_65536.txt udeb_kernel | tail -8
python3 scripts/summarize_profile.py r5_udeb_49_65536 gpurun_out/r5_udeb_any_49.txt udeb_kernel | tail -8
python3 scripts/summarize_profile.py r5_udeb_hbm65_65536 gpurun_out/r5_udeb_hbm_65.txt udeb_any_kernel | tail -8
find gpurun_out/prof_r5_udeb* -name '*_kernel_trace.csv' -delete
du -sh gpurun_out/prof_r5_udeb*
