#!/bin/bash
# round 5, session n: configs[3]'s share on the final tree under the kernel trace (50 years), and the PMC counters of its per-step kernels (10 years)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r5n_share_trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 > "$ROOT/gpurun_out/r5n_share_traced.json" 2> "$ROOT/gpurun_out/r5n_share_traced.err" || { tail -5 "$ROOT/gpurun_out/r5n_share_traced.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d "$ROOT/gpurun_out/r5n_share_pmc_sq" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r5n_pmc_sq.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r5n_pmc_sq.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/gpurun_out/r5n_share_pmc_mem" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r5n_pmc_mem.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r5n_pmc_mem.log"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r5n_share_trace 20 > gpurun_out/r5n_share_trace_table.txt; head -14 gpurun_out/r5n_share_trace_table.txt
python3 - <<'P'
import csv, glob, collections
for tag in ("sq", "mem"):
    f = glob.glob(f"gpurun_out/r5n_share_pmc_{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if int(r["Grid_Size"]) < 100000: continue      # the 125 000-member run only
        name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"## PMC pass {tag}: mean per dispatch (grid >= 100000)")
    for name, cs in sorted(acc.items()):
        print(f"{name:42s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())) + f"  dispatches={len(next(iter(cs.values())))}")
P
find gpurun_out/r5n_share_trace gpurun_out/r5n_share_pmc_sq gpurun_out/r5n_share_pmc_mem -name '*.csv' -size +2M -delete
