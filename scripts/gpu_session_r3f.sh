#!/bin/bash
# round 3, session f: instruction- and scalar-cache behaviour of configs[3]'s per-step kernels (125 000 members, 5 years of monthly steps)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r3f"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d "$OUT/pmc_a" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 5 > "$OUT/pmc_a.log" 2>&1 || { tail -5 "$OUT/pmc_a.log"; exit 1; }
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_b" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 5 > "$OUT/pmc_b.log" 2>&1 || { tail -5 "$OUT/pmc_b.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
for sub in ("pmc_a", "pmc_b"):
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
