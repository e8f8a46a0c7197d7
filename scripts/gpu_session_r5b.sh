#!/bin/bash
# round 5, session b: ClimateUDEB with the layer count at run time (every count <= 64 register-resident) -- tests, timings, then the whole GPU tier
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_udeb.py -x -q -m gpu > gpurun_out/r5b_udeb_tests.log 2>&1 || { tail -60 gpurun_out/r5b_udeb_tests.log; exit 1; }
tail -2 gpurun_out/r5b_udeb_tests.log
timeout -k 10 600 python scripts/bench_udeb_any.py 65536 > gpurun_out/r5b_udeb_any_65536.log 2>&1 || { tail -20 gpurun_out/r5b_udeb_any_65536.log; exit 1; }
cat gpurun_out/r5b_udeb_any_65536.log
timeout -k 10 600 python scripts/bench_udeb_any.py 32768 50,49,64,21 > gpurun_out/r5b_udeb_any_32768.log 2>&1 || { tail -20 gpurun_out/r5b_udeb_any_32768.log; exit 1; }
cat gpurun_out/r5b_udeb_any_32768.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5b_tests.log 2>&1 || { tail -40 gpurun_out/r5b_tests.log; exit 1; }
tail -2 gpurun_out/r5b_tests.log
