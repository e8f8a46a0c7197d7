#!/bin/bash
# round 6, session f: what the fused one-step launches wait for -- latency counters of the instruction streams: scalar memory, vector
# memory and instruction fetch (level / count = mean cycles in flight), round 5's plan (fusion 6) and this round's (fusion 1).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for f in 6 1; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d "$ROOT/gpurun_out/r6f_lat_fusion$f" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 --fusion $f --no-anchor > "$ROOT/gpurun_out/r6f_lat_fusion$f.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6f_lat_fusion$f.log"; exit 1; }
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS --output-format csv -d "$ROOT/gpurun_out/r6f_act_fusion$f" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 --fusion $f --no-anchor > "$ROOT/gpurun_out/r6f_act_fusion$f.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6f_act_fusion$f.log"; exit 1; }
done
cd "$ROOT"
python3 - <<'P'
import csv, glob, collections, sys
sys.path.insert(0, "scripts")
from summarize_profile import short
for tag in ("lat_fusion6", "act_fusion6", "lat_fusion1", "act_fusion1"):
    f = glob.glob(f"gpurun_out/r6f_{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(f)):
        if int(r["Grid_Size"]) < 100000: continue
        n = short(r["Kernel_Name"])
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"## {tag}: mean per dispatch (grid >= 100000), per wavefront where it says so")
    for n, cs in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
        if len(dur[n]) < 10: continue
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        w = m["SQ_WAVES"]
        line = f"{n[:34]:34s} us={sum(dur[n])/len(dur[n]):6.1f} waves={w:5.0f} wave_cycles/wave(x4)={4*m['SQ_WAVE_CYCLES']/w:7.0f}"
        if "SQ_INSTS_SMEM" in m:
            line += (f" smem/wave={m['SQ_INSTS_SMEM']/w:6.1f} smem_latency={m['SQ_INST_LEVEL_SMEM']/max(1,m['SQ_INSTS_SMEM']):7.0f}"
                     f" vmem/wave={m['SQ_INSTS_VMEM']/w:6.1f} vmem_latency={m['SQ_INST_LEVEL_VMEM']/max(1,m['SQ_INSTS_VMEM']):7.0f}"
                     f" ifetch/wave={m['SQ_IFETCH']/w:7.1f} ifetch_latency={m['SQ_IFETCH_LEVEL']/max(1,m['SQ_IFETCH']):6.0f}")
        else:
            line += (f" insts/wave={m['SQ_INSTS']/w:7.0f} branch/wave={m['SQ_INSTS_BRANCH']/w:6.0f} active_any={m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}"
                     f" active_valu={m['SQ_ACTIVE_INST_VALU']/m['SQ_WAVE_CYCLES']:.2f} active_sca={m['SQ_ACTIVE_INST_SCA']/m['SQ_WAVE_CYCLES']:.2f} active_vmem={m['SQ_ACTIVE_INST_VMEM']/m['SQ_WAVE_CYCLES']:.2f}")
        print(line)
P
find gpurun_out/r6f_* -name '*.csv' -size +2M -delete
