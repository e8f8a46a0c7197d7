#!/bin/bash
# Round-2 GPU session A: new / changed tests, then the two-rank rehearsal and bench.py --gpus 2 (gloo, one GPU).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export RSCM_BENCH_BACKEND=gloo RSCM_BENCH_DEVICE=0
# multi-process steps first: their launcher must not have touched the GPU
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    scripts/rehearse_two_ranks.py --out gpurun_out/two_ranks > gpurun_out/two_ranks.log 2> gpurun_out/two_ranks.err || { tail -30 gpurun_out/two_ranks.err; exit 1; }
tail -c 1500 gpurun_out/two_ranks.log; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err || { tail -30 gpurun_out/bench_2rank.err; exit 1; }
tail -c 800 gpurun_out/bench_2rank.json; echo
unset RSCM_BENCH_BACKEND RSCM_BENCH_DEVICE
timeout -k 10 900 python -m pytest tests/test_gpu_chem.py tests/test_gpu_carbon.py tests/test_gpu_sampler.py -x -q -m gpu -s > gpurun_out/pytest_a1.log 2>&1 || { tail -40 gpurun_out/pytest_a1.log; exit 1; }
tail -5 gpurun_out/pytest_a1.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_links.py -x -q -m gpu -s -k "coupled_full_size or monthly or rollback or magicc_lite or release" > gpurun_out/pytest_a2.log 2>&1 || { tail -40 gpurun_out/pytest_a2.log; exit 1; }
tail -8 gpurun_out/pytest_a2.log
