#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py command (the program itself after --).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="${1:-r1_bench}"
OUT="$ROOT/gpurun_out/prof_${TAG}"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
tail -c 600 "$OUT/bench.json"; echo
find "$OUT" -name "*kernel_stats.csv" | head -3
