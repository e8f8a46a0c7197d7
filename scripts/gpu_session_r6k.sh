#!/bin/bash
# round 6, session k: the final tree -- the GPU tier once more with every device allocation poisoned (0xFF bytes: a read of memory nobody
# wrote shows as a wrong result), the four-rank rehearsal of the sharded sampler (own pytest process), bench.py as the driver runs it.
set -o pipefail
mkdir -p gpurun_out
RSCM_POISON_ALLOC=1 timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r6k_tests_poisoned.log 2>&1 || { tail -40 gpurun_out/r6k_tests_poisoned.log; exit 1; }
tail -n 2 gpurun_out/r6k_tests_poisoned.log
timeout -k 10 900 python -m pytest tests/test_multirank_gpu.py -q -m gpu_ranks > gpurun_out/r6k_ranks.log 2>&1 || { tail -40 gpurun_out/r6k_ranks.log; exit 1; }
tail -n 2 gpurun_out/r6k_ranks.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6k_bench_details.json > gpurun_out/r6k_bench.json 2> gpurun_out/r6k_bench.err || { tail -20 gpurun_out/r6k_bench.err; exit 1; }
wc -c gpurun_out/r6k_bench.json; cat gpurun_out/r6k_bench.json
