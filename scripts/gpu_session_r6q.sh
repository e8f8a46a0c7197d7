#!/bin/bash
# round 6, session q (experiment): does ClimateUDEB's one-step launch get faster when its two 50-layer columns have just been pulled through
# the Infinity Cache?  A touch kernel over the columns in front of every ClimateUDEB launch (RSCM_EXPERIMENT_TOUCH_UDEB=1 both hemispheres,
# 2 the northern one only), kernel trace of 10 years each: the udeb_kernel's own duration is what is read.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for t in 0 1 2; do
  RSCM_EXPERIMENT_TOUCH_UDEB=$t rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6q_trace_touch$t" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 10 --no-anchor > "$ROOT/gpurun_out/r6q_touch$t.json" 2> "$ROOT/gpurun_out/r6q_touch$t.err" || { tail -5 "$ROOT/gpurun_out/r6q_touch$t.err"; exit 1; }
  (cd "$ROOT" && python3 scripts/trace_table.py gpurun_out/r6q_trace_touch$t 100000 > gpurun_out/r6q_touch${t}_table.txt && echo "== touch $t" && head -6 gpurun_out/r6q_touch${t}_table.txt)
done
find "$ROOT"/gpurun_out/r6q_* -name '*.csv' -size +2M -delete
