#!/bin/bash
# round 3, session e: what bounds the fused light-component launches of configs[3] (125 000 members, one step per launch)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r3e"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 5 > "$OUT/pmc_sq.log" 2>&1 || { tail -5 "$OUT/pmc_sq.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 5 > "$OUT/pmc_fetch.log" 2>&1 || { tail -5 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 5 > "$OUT/pmc_write.log" 2>&1 || { tail -5 "$OUT/pmc_write.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
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
