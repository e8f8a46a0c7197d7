#!/bin/bash
# round 3, session d: the profiles DESIGN.md quotes (kernel trace + PMC passes per kernel, bench.py under the profiler, configs[3])
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
bash scripts/gpu_profile_bench.sh r3_bench > gpurun_out/r3d_bench.log 2>&1 || { tail -5 gpurun_out/r3d_bench.log; exit 1; }
# (here, on the box: the kernel trace of the whole bench is too large to travel back, its per-(kernel, grid) table and the timed region's launches are what is kept)
python3 scripts/summarize_bench_profile.py r3_bench gpurun_out/r3_bench_kernel_trace.txt > /dev/null || exit 1
echo "bench done"
bash scripts/gpu_profile.sh r3_coupled_1e6 1000000 0 1 > gpurun_out/r3d_coupled.log 2>&1 || { tail -5 gpurun_out/r3d_coupled.log; exit 1; }
echo "coupled done"
DRIVER=scripts/bench_graph.py bash scripts/gpu_profile.sh r3_group_seq_1e6 1000000 > gpurun_out/r3d_group.log 2>&1 || { tail -5 gpurun_out/r3d_group.log; exit 1; }
echo "group done"
bash scripts/gpu_profile.sh r3_ocean_fast_262144 262144 1 11 > gpurun_out/r3d_ocean.log 2>&1 || { tail -5 gpurun_out/r3d_ocean.log; exit 1; }
echo "ocean done"
bash scripts/gpu_profile.sh r3_udeb_65536 65536 0 2 > gpurun_out/r3d_udeb.log 2>&1 || { tail -5 gpurun_out/r3d_udeb.log; exit 1; }
echo "udeb done"
bash scripts/gpu_profile.sh r3_exact_1e5 100000 0 0 > gpurun_out/r3d_tl1e5.log 2>&1 || { tail -5 gpurun_out/r3d_tl1e5.log; exit 1; }
echo "two-layer 1e5 done"
bash scripts/gpu_trace.sh r3_configs3_fast_100yr scripts/run_configs3_share.py --years 100 > gpurun_out/r3d_c3trace.log 2>&1 || { tail -5 gpurun_out/r3d_c3trace.log; exit 1; }
echo "configs3 trace done"
python scripts/run_configs3_share.py > gpurun_out/r3_configs3_share_fast.json 2> gpurun_out/r3_configs3_share_fast.err || { tail -3 gpurun_out/r3_configs3_share_fast.err; exit 1; }
python scripts/run_configs3_share.py --exact > gpurun_out/r3_configs3_share_exact.json 2> gpurun_out/r3_configs3_share_exact.err || { tail -3 gpurun_out/r3_configs3_share_exact.err; exit 1; }
echo "configs3 done"
find gpurun_out/prof_r3_* -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
