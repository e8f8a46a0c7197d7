#!/bin/bash
# round 6, session e: where do the fused one-step launches wait?  The op table read from DEVICE memory (fusion 3) against the by-value
# table in the kernel-argument segment, unsplit (fusion 4) -- both under the kernel trace, 10 years.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for f in 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6e_share_trace_fusion$f" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 10 --fusion $f --no-anchor > "$ROOT/gpurun_out/r6e_share_fusion$f.json" 2> "$ROOT/gpurun_out/r6e_share_fusion$f.err" || { tail -5 "$ROOT/gpurun_out/r6e_share_fusion$f.err"; exit 1; }
  (cd "$ROOT" && python3 scripts/trace_table.py gpurun_out/r6e_share_trace_fusion$f 100000 > gpurun_out/r6e_share_fusion${f}_table.txt && echo "== fusion $f" && head -6 gpurun_out/r6e_share_fusion${f}_table.txt)
done
find "$ROOT"/gpurun_out/r6e_* -name '*.csv' -size +2M -delete
