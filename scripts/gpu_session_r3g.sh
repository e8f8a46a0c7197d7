#!/bin/bash
# (a record of round 3: the whole-graph launch -- fusion mode 4 -- and the four-wavefront ClimateUDEB kernel -- variant 4 -- it exercises were removed in round 4)
# round 3, session g: scalar- and instruction-cache behaviour of the whole-graph launch (fusion mode 4), configs[3] at 125 000 members
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r3g"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d "$OUT/pmc_a" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 5 --fusion 4 > "$OUT/pmc_a.log" 2>&1 || { tail -5 "$OUT/pmc_a.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
for sub in ("pmc_a",):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/" + sub + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if int(r["Grid_Size"]) < 100000: continue     # the 125 000-member run only
            n = r["Kernel_Name"].split("(anonymous namespace)::")[-1][:48]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[n]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, d in acc.items():
        print(sub, n, {k: round(sum(v) / len(v), 1) for k, v in sorted(d.items())}, "launches", len(d["_us"]))
PY
