"""profiles/<tag>.txt from `rocprofv3 --kernel-trace --stats -- python3 bench.py`: the stats table,
the per-(kernel, grid) averages from kernel_trace.csv (bench.py launches the same kernel at
several ensemble sizes), and bench.py's own JSON line with its HIP-event kernel time."""
import collections
import csv
import glob
import json
import sys

tag, out = sys.argv[1], sys.argv[2]
d = f"gpurun_out/prof_{tag}"
lines = [f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py   ({tag}, one MI355X via gpurun)", ""]
for f in glob.glob(f"{d}/trace/*/*_kernel_stats.csv"):
    lines.append("## kernel_stats.csv")
    lines += [l.rstrip() for l in open(f)]
groups = collections.defaultdict(list)
starts = collections.defaultdict(list)
for f in glob.glob(f"{d}/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "rscm::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("rscm::(anonymous namespace)::")[-1].split("(")[0]
            groups[(name, int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            starts[(name, int(r["Grid_Size_X"]))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
lines += ["", "## kernel_trace.csv grouped by (kernel, grid size = members rounded up to 256)",
          "kernel, grid, calls, avg_ms, min_ms, max_ms"]
for (name, grid), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"{name}, {grid}, {len(v)}, {sum(v) / len(v) / 1e6:.4f}, {min(v) / 1e6:.4f}, {max(v) / 1e6:.4f}")
js = json.loads(open(f"{d}/bench.json").read().strip().splitlines()[-1])
# the headline's timed region: bench.py launches the headline kernel W times to warm up and then K times inside the timed region,
# before anything else uses that kernel at that ensemble size -- launches W+1 .. W+K of its (kernel, grid) group in start order
members = js.get("config", {}).get("members_per_gpu")
if members and starts:
    grid = (members + 255) // 256 * 256
    for (name, g), v in starts.items():
        if name.startswith(js["roofline"]["kernel"]) and g == grid and len(v) >= js["warmup"] + js["steps"]:
            timed = [dur for _, dur in sorted(v)[js["warmup"]:js["warmup"] + js["steps"]]]
            lines += ["", f"## the timed region of bench.py in the trace: launches {js['warmup'] + 1}..{js['warmup'] + js['steps']} of {name} at grid {g}",
                      f"avg_ms = {sum(timed) / len(timed) / 1e6:.4f}, min_ms = {min(timed) / 1e6:.4f}, max_ms = {max(timed) / 1e6:.4f}"
                      f"   (bench.py's own clock around the same launches: ms_per_step = {js['ms_per_step']:.4f})"]
lines += ["", "## bench.py's own line under the profiler (HIP events on the launch stream)",
          f"value = {js['value']:.6g} {js['unit']}; ms_per_step = {js['ms_per_step']:.4f}; "
          f"roofline.kernel_ms = {js['roofline']['kernel_ms']:.4f}; roofline.frac = {js['roofline']['frac']:.4f}",
          "extra = " + json.dumps(js.get("extra", {}))]
key = [k for k in groups if k[0].startswith("two_layer_kernel<0") and k[1] == 100096]
if key:
    v = groups[key[0]]
    lines.append(f"agreement: rocprofv3 average of two_layer_kernel<0,true> at grid 100096 = {sum(v) / len(v) / 1e6:.4f} ms "
                 f"vs bench.py roofline.kernel_ms = {js['roofline']['kernel_ms']:.4f} ms")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-8:]))
