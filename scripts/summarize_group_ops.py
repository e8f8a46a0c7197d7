"""profiles/r2_group_ops_counters.txt from the counter runs of scripts/profile_group_ops.py
(scripts/gpu_session_e.sh): instructions per wavefront and model step of the fused graph launch, by variant
of the coupled chain, with and without the LDS slots."""
import collections
import csv
import glob
import sys

NAMES = ("cc ce ag tl", "cc ce tl", "ce ag tl", "ag tl", "ag ag tl", "ag ag ag tl", "cc ce", "ce ag")
out = ["# Fused graph launch (csrc/group.hip), 1e6 members x 750 steps, per wavefront and model step",
       "# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -- python3 scripts/profile_group_ops.py 1000000",
       "# cc CarbonCycle, ce CO2ERF, ag Sum aggregate of one contributor, tl TwoLayer; every variant is ONE launch",
       "# the fused coupled_kernel for comparison: 1564 VALU + 52 SALU per wavefront and step, 37.6-38.5 ms", ""]
for label, d, log in (("mode 1: LDS slots between the steps (default)", "gpurun_out/pmc_group_ops_mode1", "gpurun_out/group_ops_mode1.log"),
                      ("mode 2: no slots (parameters, states and linked values re-read from HBM every step)", "gpurun_out/pmc_group_ops_mode2", "gpurun_out/group_ops_mode2.log"),
                      ("before this work (commit 47ecd3d): no slots, select-style aggregate, branch-per-division CarbonCycle", "gpurun_out/pmc_pgo", "gpurun_out/pgo.log.before")):
    fs = glob.glob(f"{d}/*/*counter_collection.csv")
    if not fs:
        continue
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if "group_kernel" in r["Kernel_Name"]:
            rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ms = {}
    try:
        for l in open(log):
            if " ms " in l:
                ms[l.split(":")[0].strip()] = l.split(":")[1].split("ms")[0].strip()
    except OSError:
        pass
    out.append(f"## {label}")
    cols = sorted({c for v in rows.values() for c in v if c not in ("SQ_WAVES", "SQ_BUSY_CYCLES")})
    out.append("variant".ljust(14) + "".join(c[8:].rjust(10) for c in cols) + "   ms (not under counters for the last block)")
    for k, i in enumerate(list(rows)[::3]):
        v = rows[i]
        w = v["SQ_WAVES"]
        out.append(NAMES[k].ljust(14) + "".join(f"{v[c] / w / 750:10.0f}" for c in cols) + "   " + ms.get(NAMES[k], ""))
    out.append("")
open(sys.argv[1], "w").write("\n".join(out))
print("\n".join(out))
