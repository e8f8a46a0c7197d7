#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
bash scripts/gpu_profile.sh r2_ocean_fast_262144 262144 1 11 > gpurun_out/prof_ocean_fast.log 2>&1 || { tail -20 gpurun_out/prof_ocean_fast.log; exit 1; }
python scripts/summarize_profile.py r2_ocean_fast_262144 gpurun_out/r2_ocean_fast_262144.txt > /dev/null || exit 1
tail -12 gpurun_out/r2_ocean_fast_262144.txt
python scripts/bench_magicc_chain.py --members 100000 --years 750 > gpurun_out/chain_exact.log 2>&1 || { tail gpurun_out/chain_exact.log; exit 1; }
python scripts/bench_magicc_chain.py --members 100000 --years 750 --fast > gpurun_out/chain_fast.log 2>&1 || { tail gpurun_out/chain_fast.log; exit 1; }
grep "^run" gpurun_out/chain_exact.log gpurun_out/chain_fast.log
