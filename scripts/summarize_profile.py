"""Condense a gpurun_out/prof_<tag>/ directory (scripts/gpu_profile.sh) into one text file under
profiles/: the --kernel-trace --stats table, the driver's own HIP-event timings, and the PMC
counters per dispatch, plus the derived figures DESIGN.md quotes (clock, VALU issue utilisation,
HBM traffic with the gfx950 FETCH_SIZE x2 correction).

    python3 scripts/summarize_profile.py <tag> <out.txt> [kernel-name-substring]

One summary describes ONE kernel.  When the profiled program launches several of the hot kernels (the
fused coupled kernel beside the group kernel, say) the third argument says which one is meant; without
it the script refuses rather than mixing the dispatches of two kernels into one block of counters and
dividing one kernel's bytes by the other's duration."""
import collections
import csv
import glob
import sys

HOT = ("two_layer_kernel", "coupled_kernel", "coupled_fast_kernel", "udeb_kernel", "udeb2_kernel", "udeb2_lds_kernel", "udeb_any_kernel", "ghg_kernel", "ocean_kernel", "ocean_recur_kernel",
       "group_kernel", "group_seq_kernel")


def short(name):
    """'void rscm::(anonymous namespace)::coupled_kernel<true>(rscm::CoupledArgs)' -> 'coupled_kernel<true>'; template arguments that are
    themselves types of the anonymous namespace keep their names ('group_split_seq_kernel<OpKinds<17, 10, 9, 7>, ...>')"""
    head = name[5:] if name.startswith("void ") else name
    head = head.replace("rscm::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in head:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def main():
    tag, out = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else None
    d = f"gpurun_out/prof_{tag}"
    lines = [f"# rocprofv3 summary: {tag}" + (f" (kernel: {want})" if want else ""),
             "# produced by: bash scripts/gpu_profile.sh <tag> <members> <mode> <kind> on one MI355X (gpurun)",
             "# condensed by: python3 scripts/summarize_profile.py " + " ".join(sys.argv[1:]), ""]

    def is_hot(name):
        return any(k in name for k in HOT) and (want is None or want in name)

    # ---- kernel trace: the table as it is, and the hot kernel's un-profiled average
    avg_ns, hot_names = None, set()
    for f in glob.glob(f"{d}/trace/runc/*_kernel_stats.csv"):
        lines.append("## rocprofv3 --kernel-trace --stats  (kernel_stats.csv)")
        lines += [l.rstrip() for l in open(f)]
        for r in csv.DictReader(open(f)):
            if is_hot(r["Name"]):
                hot_names.add(short(r["Name"]))
                avg_ns = float(r["AverageNs"])
    if len(hot_names) > 1:
        sys.exit(f"summarize_profile: the trace holds {len(hot_names)} hot kernels ({', '.join(sorted(hot_names))}); "
                 "name the one this summary is about as the third argument")
    lines += ["", "## driver output under --kernel-trace (HIP events on the launch stream)"]
    lines += [l for l in open(f"{d}/trace.log").read().splitlines() if l.startswith(("kind=", "member-years", "N=", "{"))]

    # ---- counters: one kernel only
    C = {}
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        fs = glob.glob(f"{d}/{sub}/runc/*_counter_collection.csv")
        if not fs:
            continue
        acc, meta, dur, names = collections.defaultdict(list), {}, [], set()
        for r in csv.DictReader(open(fs[0])):
            if is_hot(r["Kernel_Name"]):
                names.add(short(r["Kernel_Name"]))
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = r
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        if len(names) > 1:
            sys.exit(f"summarize_profile: pass {sub} holds dispatches of {len(names)} hot kernels ({', '.join(sorted(names))}); "
                     "name the one this summary is about as the third argument")
        lines += ["", f"## rocprofv3 --pmc pass '{sub}' (separate run), per dispatch of {meta.get('Kernel_Name', '')}",
                  f"# grid={meta.get('Grid_Size')} workgroup={meta.get('Workgroup_Size')}; dispatch ms under counters: "
                  + " ".join(f"{x:.3f}" for x in sorted(set(dur))[:4])]
        for k, v in sorted(acc.items()):
            lines.append(f"{k}: " + " ".join(f"{x:.6g}" for x in v))
            C[k] = v[-1]
            C[k + "_ms"] = dur[-1]
    lines += ["", "## derived" + (f" ({sorted(hot_names)[0]})" if hot_names else "")]
    ghz = None
    if "GRBM_GUI_ACTIVE" in C:
        ghz = C["GRBM_GUI_ACTIVE"] / 8 / (C["GRBM_GUI_ACTIVE_ms"] * 1e-3) / 1e9
        lines.append(f"effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time = {ghz:.2f} GHz")
    if "SQ_INSTS_VALU" in C and "SQ_WAVES" in C:
        lines.append(f"VALU instructions per wavefront = {C['SQ_INSTS_VALU'] / C['SQ_WAVES']:.0f}"
                     f"  (per wavefront-year over 750 years = {C['SQ_INSTS_VALU'] / C['SQ_WAVES'] / 750:.1f})")
        if "SQ_INSTS_SALU" in C:
            lines.append(f"SALU instructions per wavefront-year = {C['SQ_INSTS_SALU'] / C['SQ_WAVES'] / 750:.1f}")
        if ghz:
            cyc = C["SQ_INSTS_VALU_ms"] * 1e-3 * ghz * 1e9
            lines.append(f"VALU issue utilisation = 4 cycles x SQ_INSTS_VALU / 1024 SIMDs / kernel cycles = "
                         f"{4 * C['SQ_INSTS_VALU'] / 1024 / cyc:.2f}  (a wave64 f64 instruction occupies its SIMD for 4 cycles)")
    if "FETCH_SIZE" in C and "WRITE_SIZE" in C:
        fetch = 2 * C["FETCH_SIZE"] * 1024  # gfx950: FETCH_SIZE reports half of a coalesced stream (MI355X_MICROARCH.md, HBM)
        write = C["WRITE_SIZE"] * 1024
        lines.append(f"HBM traffic per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB = {fetch / 1e6:.1f} MB read + {write / 1e6:.1f} MB written")
        if avg_ns:
            lines.append(f"  over the un-profiled average launch of the same kernel ({avg_ns / 1e6:.3f} ms) = {(fetch + write) / avg_ns:.1f} GB/s")
    open(out, "w").write("\n".join(lines) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
