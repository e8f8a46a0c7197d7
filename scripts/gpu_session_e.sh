#!/bin/bash
# Round 2, last batch: the fused graph launch with LDS slots -- rocprofv3 trace + counters of the coupled chain as
# linked ensembles, and the per-component counters (components left out one at a time).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
DRIVER=scripts/bench_graph.py bash scripts/gpu_profile.sh r2_group_coupled_1e6 1000000 > gpurun_out/prof_group.log 2>&1 || { tail -20 gpurun_out/prof_group.log; exit 1; }
python scripts/summarize_profile.py r2_group_coupled_1e6 gpurun_out/r2_group_coupled_1e6.txt > /dev/null || exit 1
head -12 gpurun_out/r2_group_coupled_1e6.txt | cut -c1-200
export TMPDIR=/tmp
ROOT="$(pwd)"
for MODE in 1 2; do
  (cd /tmp && RSCM_GROUP_MODE=$MODE rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv \
      -d "$ROOT/gpurun_out/pmc_group_ops_mode$MODE" -- python3 "$ROOT/scripts/profile_group_ops.py" 1000000 > "$ROOT/gpurun_out/group_ops_mode$MODE.log" 2>&1) || { tail -5 gpurun_out/group_ops_mode$MODE.log; exit 1; }
  grep " ms " gpurun_out/group_ops_mode$MODE.log
done
