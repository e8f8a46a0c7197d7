"""Condense the rocprofv3 passes of one configs[3] share run (scripts/run_configs3_share.py under the profiler) into one table per
kernel of the 125 000-member model: dispatches, duration, executed vector instructions, FP64 issue utilisation, HBM bytes fetched and
written -- and the same for a whole monthly step (the kernels of a step run strictly one after the other on one stream).

    python3 scripts/summarize_share_pmc.py --sq DIR --fetch DIR --write DIR [--trace DIR] [--min-grid 100000] --members 125000 \
        --out profiles/r6_configs3_share_pmc.txt [--traffic-key 'configs3_share|125000|fast']

The three counter directories come from three SEPARATE --pmc runs (FETCH_SIZE costs 3 of the 4 TCC slots and WRITE_SIZE 2: they
cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_* in one, FETCH_SIZE + GRBM_GUI_ACTIVE in one, WRITE_SIZE in one.
FETCH_SIZE is doubled (gfx950 tallies a coalesced 128-B request at 64 B; same guide, section HBM); both are KiB.

Issue utilisation = 4 cycles x SQ_INSTS_VALU / 1024 SIMDs / (duration x shader clock): a wave64 f64 instruction occupies its SIMD
for 4 cycles.  The clock is GRBM_GUI_ACTIVE / 8 XCDs / duration of the same dispatch in the fetch pass."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from summarize_profile import short  # noqa: E402


def read_pass(d, min_grid):
    """{kernel: {counter: [values per dispatch]}, ...}, {kernel: [dispatch us]}, {kernel: resource row}"""
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        sys.exit(f"{d}: no counter_collection.csv")
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    seen, dur, meta = set(), collections.defaultdict(list), {}
    for r in csv.DictReader(open(fs[0])):
        if int(r["Grid_Size"]) < min_grid:
            continue
        name = short(r["Kernel_Name"])
        counters[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[name] = r
    return counters, dur, meta


def read_trace(d, min_grid):
    """Un-profiled durations per kernel from a --kernel-trace run: {kernel: [us]}"""
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) < min_grid:
                continue
            out[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def mean(v):
    return sum(v) / len(v) if v else float("nan")


def main():
    if len(sys.argv) == 5 and sys.argv[1] == "--from-json" and sys.argv[3] == "--traffic-key":
        path = os.path.join(ROOT, "profiles", "traffic.json")
        table = json.load(open(path))
        table[sys.argv[4]] = json.load(open(sys.argv[2]))
        json.dump(table, open(path, "w"), indent=1)
        print(sys.argv[4], table[sys.argv[4]])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--sq", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--trace")
    ap.add_argument("--trace-table", help="a committed trace table (scripts/trace_table.py) with the un-profiled per-dispatch durations")
    ap.add_argument("--min-grid", type=int, default=100000)
    ap.add_argument("--members", type=int, default=125000)
    ap.add_argument("--step-kernels", default="udeb_kernel,ocean_recur_kernel,group_split_kernel,group_kernel_args",
                    help="the kernels launched once per model step (substrings); everything else is set-up / summary work")
    ap.add_argument("--min-dispatches", type=int, default=1,
                    help="a step kernel with fewer dispatches than this in the SQ pass is left out of the per-step sums (the prologue / epilogue "
                         "launches of the merged schedule)")
    ap.add_argument("--series", type=int, default=36, help="stored variables of the graph: algorithmic bytes per member-step = 8 x this")
    ap.add_argument("--out", required=True)
    ap.add_argument("--traffic-key")
    ap.add_argument("--title", default="configs[3] share, MAGICC graph, FAST")
    args = ap.parse_args()

    sq, sq_us, meta = read_pass(args.sq, args.min_grid)
    fe, fe_us, _ = read_pass(args.fetch, args.min_grid)
    wr, wr_us, _ = read_pass(args.write, args.min_grid)
    trace = read_trace(args.trace, args.min_grid) if args.trace else {}
    step = [s for s in args.step_kernels.split(",") if s]

    lines = [f"# rocprofv3 PMC summary: {args.title}, {args.members} members",
             "# three separate --pmc runs of scripts/run_configs3_share.py (program directly after --): "
             "SQ_* | FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE ...",
             "# condensed by: python3 scripts/summarize_share_pmc.py " + " ".join(sys.argv[1:]),
             f"# dispatches with grid >= {args.min_grid} only (the {args.members}-member model; the 64-member parity anchor's launches are left out)",
             "# fetched = 2 x FETCH_SIZE KiB (gfx950 correction), written = WRITE_SIZE KiB; issue = 4 x SQ_INSTS_VALU / 1024 SIMDs / (us x clock)",
             "# clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time, capped at the nominal 2.4 GHz (the counter includes the cycles around a short dispatch)", ""]
    head = (f"{'kernel':44s} {'disp':>5s} {'us':>8s} {'us(trace)':>9s} {'clock':>6s} {'waves':>7s} {'VALU/wave':>10s} {'SALU/wave':>10s} "
            f"{'issue':>6s} {'wait_any':>8s} {'fetched MB':>11s} {'written MB':>11s} {'GB/s':>7s} {'B/member':>9s} {'VGPR':>5s} {'AGPR':>5s} {'LDS':>6s} {'scratch':>7s}")
    lines.append(head)
    per_step = dict(us=0.0, us_trace=0.0, valu=0.0, fetched=0.0, written=0.0, cycles=0.0)
    rows = {}
    for name in sorted(sq, key=lambda n: -mean(sq_us[n]) * len(sq_us[n])):
        c = sq[name]
        us = mean(sq_us[name])
        ghz = None
        if name in fe and fe[name].get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE also counts the cycles around the dispatch (command processing): over a launch of tens of microseconds
            # that reads as a clock above the chip's 2.4 GHz.  Capped there: the issue utilisation of the short kernels is then a lower
            # bound on the cycles, i.e. the utilisation is not overstated.
            ghz = min(2.4, mean(fe[name]["GRBM_GUI_ACTIVE"]) / 8 / (mean(fe_us[name]) * 1e-6) / 1e9)
        waves = mean(c.get("SQ_WAVES", []))
        valu = mean(c.get("SQ_INSTS_VALU", []))
        salu = mean(c.get("SQ_INSTS_SALU", []))
        issue = 4 * valu / 1024 / (us * 1e-6 * ghz * 1e9) if ghz else float("nan")
        wait = mean(c["SQ_WAIT_ANY"]) / mean(c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES") else float("nan")
        fetched = 2 * mean(fe[name]["FETCH_SIZE"]) * 1024 if name in fe and fe[name].get("FETCH_SIZE") else float("nan")
        written = mean(wr[name]["WRITE_SIZE"]) * 1024 if name in wr and wr[name].get("WRITE_SIZE") else float("nan")
        t_us = mean(trace.get(name, []))
        base_us = t_us if t_us == t_us else us
        m = meta[name]
        rows[name] = dict(dispatches=len(sq_us[name]), us=us, us_trace=t_us, clock_ghz=ghz, waves=waves, valu=valu, issue=issue,
                          wait_any=wait, fetched=fetched, written=written)
        lines.append(f"{name[:44]:44s} {len(sq_us[name]):5d} {us:8.1f} {t_us:9.1f} {ghz or float('nan'):6.2f} {waves:7.0f} {valu / waves if waves else float('nan'):10.1f} "
                     f"{salu / waves if waves else float('nan'):10.1f} {issue:6.2f} {wait:8.2f} {fetched / 1e6:11.2f} {written / 1e6:11.2f} "
                     f"{(fetched + written) / (base_us * 1e-6) / 1e9:7.0f} {(fetched + written) / args.members:9.1f} "
                     f"{m['VGPR_Count']:>5s} {m['Accum_VGPR_Count']:>5s} {m['LDS_Block_Size']:>6s} {m['Scratch_Size']:>7s}")
        if any(s in name for s in step) and len(sq_us[name]) >= args.min_dispatches:
            per_step["us"] += us
            per_step["us_trace"] += base_us
            per_step["valu"] += valu
            per_step["fetched"] += fetched
            per_step["written"] += written
            per_step["cycles"] += us * 1e-6 * (ghz or 2.4) * 1e9
    lines.append("")
    alg = 8.0 * args.series * args.members
    total = per_step["fetched"] + per_step["written"]
    issue_step = 4 * per_step["valu"] / 1024 / per_step["cycles"] if per_step["cycles"] else float("nan")
    lines += ["## one model step = " + " -> ".join(n for n in rows if any(s in n for s in step) and rows[n]["dispatches"] >= args.min_dispatches) + " (one launch each, one stream, in a dependency chain)",
              f"kernel time per step under counters = {per_step['us']:.1f} us; un-profiled (kernel trace) = {per_step['us_trace']:.1f} us",
              f"HBM traffic per step = {per_step['fetched'] / 1e6:.1f} MB read + {per_step['written'] / 1e6:.1f} MB written = "
              f"{total / 1e6:.1f} MB = {total / args.members:.0f} B per member-step",
              f"algorithmic bytes per step = {args.series} series x 8 B x {args.members} members = {alg / 1e6:.1f} MB = {8 * args.series} B per member-step; "
              f"measured / algorithmic = {total / alg:.2f}",
              f"HBM rate over the step's kernel time = {total / (per_step['us_trace'] * 1e-6) / 1e9:.0f} GB/s = {total / (per_step['us_trace'] * 1e-6) / 8e12:.3f} of 8 TB/s",
              f"FP64 issue utilisation of the whole step = 4 x sum(SQ_INSTS_VALU) / 1024 SIMDs / sum(kernel cycles) = {issue_step:.2f}"]
    text = "\n".join(lines) + "\n"
    open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w").write(text)
    print(text)
    entry = {"bytes_per_member_step": total / args.members, "read_per_member_step": per_step["fetched"] / args.members,
             "written_per_member_step": per_step["written"] / args.members, "members": args.members,
             "valu_issue_utilisation": round(issue_step, 3), "kernel_us_per_step": per_step["us_trace"],
             "source": args.out.replace("gpurun_out/", "profiles/")}
    # (the passes' CSVs are too large to travel back from the GPU box: the figures go into a small file beside the summary, and
    # `--from-json FILE --traffic-key KEY` enters them into profiles/traffic.json here)
    json.dump(entry, open((os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out) + ".json", "w"), indent=1)
    if args.traffic_key:
        path = os.path.join(ROOT, "profiles", "traffic.json")
        table = json.load(open(path))
        table[args.traffic_key] = entry
        json.dump(table, open(path, "w"), indent=1)
        print(f"{args.traffic_key}: {total / args.members:.0f} B per member-step, issue {issue_step:.2f} -> profiles/traffic.json")


if __name__ == "__main__":
    main()
