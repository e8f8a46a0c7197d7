#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python program (the program itself after --):
#   bash scripts/gpu_trace.sh <tag> <script> [args...]   -> gpurun_out/prof_<tag>/, summary gpurun_out/<tag>_kernel_stats.txt
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="$1"; shift
OUT="$ROOT/gpurun_out/prof_${TAG}"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log" || { tail -5 "$OUT/stderr.log"; exit 1; }
F=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 $*"; echo "# program output:"; sed 's/^/#   /' "$OUT/stdout.log" | cut -c1-1200; echo; cat "$F"; } > "$ROOT/gpurun_out/${TAG}_kernel_stats.txt"
cat "$ROOT/gpurun_out/${TAG}_kernel_stats.txt" | cut -c1-220 | head -40
