#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_chem.py tests/test_gpu_carbon.py tests/test_gpu_sampler.py -x -q -m gpu -s > gpurun_out/pytest_a1.log 2>&1 || { tail -40 gpurun_out/pytest_a1.log; exit 1; }
tail -5 gpurun_out/pytest_a1.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_links.py -x -q -m gpu -s -k "coupled_full_size or monthly or rollback or magicc_lite or release" > gpurun_out/pytest_a2.log 2>&1 || { tail -40 gpurun_out/pytest_a2.log; exit 1; }
tail -8 gpurun_out/pytest_a2.log
