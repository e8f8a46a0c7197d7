#!/bin/bash
# round 5, session e: the work queue with device-scope hand-over accesses instead of fences -- parity, then the sweep
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cut_into_member_blocks or failed_chunk_launch" > gpurun_out/r5e_tests.log 2>&1 || { tail -60 gpurun_out/r5e_tests.log; exit 1; }
tail -n 2 gpurun_out/r5e_tests.log
timeout -k 10 900 bash scripts/sweep_queue.sh gpurun_out/r5e_sweep_queue.txt
