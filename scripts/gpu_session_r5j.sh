#!/bin/bash
# round 5, session j: ClimateUDEB at 65..128 layers with the c' array in LDS -- tests, timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_udeb.py -x -q -m gpu > gpurun_out/r5j_udeb_tests.log 2>&1 || { tail -60 gpurun_out/r5j_udeb_tests.log; exit 1; }
tail -n 2 gpurun_out/r5j_udeb_tests.log
timeout -k 10 600 python scripts/bench_udeb_any.py 65536 > gpurun_out/r5j_udeb_any_65536.log 2>&1 || { tail -20 gpurun_out/r5j_udeb_any_65536.log; exit 1; }
cat gpurun_out/r5j_udeb_any_65536.log
timeout -k 10 600 python scripts/bench_udeb_any.py 16384 65,128,64 > gpurun_out/r5j_udeb_any_16384.log 2>&1 || { tail -20 gpurun_out/r5j_udeb_any_16384.log; exit 1; }
cat gpurun_out/r5j_udeb_any_16384.log
