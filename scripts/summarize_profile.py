"""Condense a gpurun_out/prof_<tag>/ directory (scripts/gpu_profile.sh) into one text file under
profiles/: the --kernel-trace --stats table, the driver's own HIP-event timings, and the PMC
counters per dispatch, plus the derived figures DESIGN.md quotes (clock, VALU issue utilisation,
HBM traffic with the gfx950 FETCH_SIZE x2 correction)."""
import collections
import csv
import glob
import sys

tag, out = sys.argv[1], sys.argv[2]
d = f"gpurun_out/prof_{tag}"
lines = [f"# rocprofv3 summary: {tag}",
         "# produced by: bash scripts/gpu_profile.sh <tag> <members> <mode> <kind> on one MI355X (gpurun)",
         "# driver: python3 scripts/profile_two_layer.py <members> <mode> <passes> <kind>", ""]
avg_ns = None
for f in glob.glob(f"{d}/trace/runc/*_kernel_stats.csv"):
    lines.append("## rocprofv3 --kernel-trace --stats  (kernel_stats.csv)")
    for l in open(f):
        lines.append(l.rstrip())
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("two_layer_kernel", "coupled_kernel", "udeb_kernel", "ghg_kernel", "ocean_kernel", "ocean_recur_kernel", "group_kernel")):
            avg_ns = float(r["AverageNs"])
lines += ["", "## driver output under --kernel-trace (HIP events on the launch stream)"]
lines += [l for l in open(f"{d}/trace.log").read().splitlines() if l.startswith(("kind=", "member-years", "N="))]
C = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    fs = glob.glob(f"{d}/{sub}/runc/*_counter_collection.csv")
    if not fs:
        continue
    acc, meta, dur = collections.defaultdict(list), {}, []
    for r in csv.DictReader(open(fs[0])):
        if any(k in r["Kernel_Name"] for k in ("two_layer_kernel", "coupled_kernel", "udeb_kernel", "ghg_kernel", "ocean_kernel", "ocean_recur_kernel", "group_kernel")):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    lines += ["", f"## rocprofv3 --pmc pass '{sub}' (separate run), per dispatch of {meta.get('Kernel_Name', '')}",
              f"# grid={meta.get('Grid_Size')} workgroup={meta.get('Workgroup_Size')}; dispatch ms under counters: "
              + " ".join(f"{x:.3f}" for x in sorted(set(dur))[:4])]
    for k, v in sorted(acc.items()):
        lines.append(f"{k}: " + " ".join(f"{x:.6g}" for x in v))
        C[k] = v[-1]
        C[k + "_ms"] = dur[-1]
lines += ["", "## derived"]
if "GRBM_GUI_ACTIVE" in C:
    ghz = C["GRBM_GUI_ACTIVE"] / 8 / (C["GRBM_GUI_ACTIVE_ms"] * 1e-3) / 1e9
    lines.append(f"effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time = {ghz:.2f} GHz")
if "SQ_INSTS_VALU" in C and "SQ_WAVES" in C:
    lines.append(f"VALU instructions per wavefront = {C['SQ_INSTS_VALU'] / C['SQ_WAVES']:.0f}"
                 f"  (per wavefront-year over 750 years = {C['SQ_INSTS_VALU'] / C['SQ_WAVES'] / 750:.1f})")
    if "GRBM_GUI_ACTIVE" in C:
        cyc = C["SQ_INSTS_VALU_ms"] * 1e-3 * ghz * 1e9
        lines.append(f"VALU issue utilisation = 4 cycles x SQ_INSTS_VALU / 1024 SIMDs / kernel cycles = "
                     f"{4 * C['SQ_INSTS_VALU'] / 1024 / cyc:.2f}  (a wave64 f64 instruction occupies its SIMD for 4 cycles)")
if "FETCH_SIZE" in C and "WRITE_SIZE" in C:
    fetch = 2 * C["FETCH_SIZE"] * 1024  # gfx950: FETCH_SIZE reports half of a coalesced stream (MI355X_MICROARCH.md, HBM)
    write = C["WRITE_SIZE"] * 1024
    lines.append(f"HBM traffic per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB = {fetch / 1e6:.1f} MB read + {write / 1e6:.1f} MB written")
    if avg_ns:
        lines.append(f"  over the un-profiled average launch ({avg_ns / 1e6:.3f} ms) = {(fetch + write) / avg_ns:.1f} GB/s")
open(out, "w").write("\n".join(lines) + "\n")
print(open(out).read())
