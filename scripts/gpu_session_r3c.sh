#!/bin/bash
# (a record of round 3: the whole-graph launch -- fusion mode 4 -- and the four-wavefront ClimateUDEB kernel -- variant 4 -- it exercises were removed in round 4)
# round 3, session c: instruction-cache behaviour of the whole-graph launch (configs[3] share, 120 monthly steps)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r3c"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for f in 1 4; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_ic_f$f" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 10 --fusion $f > "$OUT/pmc_ic_f$f.log" 2>&1 || { tail -5 "$OUT/pmc_ic_f$f.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_ic_f$f/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("graph_kernel", "udeb_kernel", "udeb2_kernel", "group_kernel", "ocean_recur"):
            if key in n:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print("fusion $f", key, {k: (len(v), round(sum(v) / len(v))) for k, v in sorted(d.items())})
PY
done
