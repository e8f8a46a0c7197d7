#!/bin/bash
# rocprofv3 recipe for the GPU box (run via gpurun from the repo root):
#   kernel-trace + stats in one run, PMC counters in separate runs (no trace domains mixed in).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="${1:-r1}"; MEMBERS="${2:-100000}"; MODE="${3:-0}"; KIND="${4:-0}"
# DRIVER: the program profiled (default: scripts/profile_two_layer.py <members> <mode> <passes> <kind>);
# any other script gets <members> only, e.g. DRIVER=scripts/bench_graph.py
DRIVER="${DRIVER:-scripts/profile_two_layer.py}"
OUT="$ROOT/gpurun_out/prof_${TAG}"
if [ "$DRIVER" = "scripts/profile_two_layer.py" ]; then EXTRA5="$MODE 5 $KIND"; EXTRA3="$MODE 3 $KIND"; else EXTRA5=""; EXTRA3=""; fi
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/$DRIVER" "$MEMBERS" $EXTRA5 > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/$DRIVER" "$MEMBERS" $EXTRA3 > "$OUT/pmc_sq.log" 2>&1 || { tail -5 "$OUT/pmc_sq.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/$DRIVER" "$MEMBERS" $EXTRA3 > "$OUT/pmc_fetch.log" 2>&1 || { tail -5 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/$DRIVER" "$MEMBERS" $EXTRA3 > "$OUT/pmc_write.log" 2>&1 || { tail -5 "$OUT/pmc_write.log"; exit 1; }
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
