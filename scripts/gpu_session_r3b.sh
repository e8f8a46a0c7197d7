#!/bin/bash
# (a record of round 3: the whole-graph launch -- fusion mode 4 -- and the four-wavefront ClimateUDEB kernel -- variant 4 -- it exercises were removed in round 4)
# round 3, session b: where the whole-graph launch spends its time (configs[3] share, 600 monthly steps)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r3b"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for f in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_f$f" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 --fusion $f > "$OUT/trace_f$f.log" 2>&1 || { tail -5 "$OUT/trace_f$f.log"; exit 1; }
  cat "$OUT"/trace_f$f/*/*_kernel_stats.csv | head -12
done
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQ_INST_LEVEL|SQ_WAIT" | head -40 > "$OUT/counters.txt"; cat "$OUT/counters.txt" | cut -c1-200
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 10 --fusion 1 > "$OUT/pmc_sq.log" 2>&1 || { tail -5 "$OUT/pmc_sq.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_sq/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "graph_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, len(v), sum(v) / len(v))
PY
