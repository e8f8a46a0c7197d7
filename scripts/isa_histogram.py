"""Instruction histogram of the loops of one kernel, from the ISA hipcc emits (`make -C rscm_amd/csrc asm`
writes build/<file>.s).  For every self-loop (a block that ends in a backward branch to its own label:
the RK4 sub-step loops of the stepping kernels) the instructions are counted by mnemonic and grouped
into f64 VALU / other VALU / SALU / memory / control.  This is where DESIGN.md's "78 f64 + 12 int32
VALU instructions per RK4 step" comes from.

    python scripts/isa_histogram.py rscm_amd/csrc/build/two_layer.s 'two_layer_kernelILi0ELb1ELb1E' > profiles/r2_two_layer_isa_histogram.txt
"""
import collections
import re
import sys


def classify(m):
    if m.startswith("v_") and "_f64" in m:
        return "VALU f64"
    if m.startswith("v_"):
        return "VALU other (int32 / moves / compares)"
    if m.startswith("s_cbranch") or m.startswith("s_branch") or m in ("s_endpgm", "s_barrier"):
        return "control"
    if m.startswith("s_waitcnt") or m == "s_nop":
        return "wait"
    if m.startswith("s_load") or m.startswith("s_buffer"):
        return "scalar memory"
    if m.startswith("s_"):
        return "SALU"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vector memory"
    if m.startswith("ds_"):
        return "LDS"
    return "other"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(needle) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    name = lines[start].split(":")[0]
    print(f"# ISA instruction histogram of {name}")
    print(f"# source: {path} (hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -S), lines {start + 1}-{end + 1}")
    labels = {}
    for i in range(start, end + 1):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    total = collections.Counter()
    for i in range(start, end + 1):
        m = re.match(r"^\s+([a-z_0-9]+)", lines[i])
        if m and not lines[i].lstrip().startswith((".", ";")):
            total[classify(m.group(1))] += 1
    print(f"# whole kernel: {sum(total.values())} instructions: " + ", ".join(f"{k} {v}" for k, v in sorted(total.items())))
    for i in range(start, end + 1):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i])
        if not m or m.group(1) not in labels or labels[m.group(1)] > i:
            continue
        head = labels[m.group(1)]
        if any(re.match(r"^\.LBB", lines[j]) for j in range(head + 1, i)):
            continue  # not a single-block self loop
        hist, groups = collections.Counter(), collections.Counter()
        for j in range(head + 1, i + 1):
            mm = re.match(r"^\s+([a-z_0-9]+)", lines[j])
            if mm and not lines[j].lstrip().startswith((".", ";")):
                hist[mm.group(1)] += 1
                groups[classify(mm.group(1))] += 1
        n = sum(hist.values())
        comment = lines[head].split(";", 1)[1].strip() if ";" in lines[head] else ""
        print(f"\n## loop {m.group(1)} (lines {head + 1}-{i + 1}; {comment}): {n} instructions per iteration")
        for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
            print(f"  {k:42s} {v:4d}")
        print("  -- by mnemonic")
        for k, v in sorted(hist.items(), key=lambda kv: (-kv[1], kv[0])):
            print(f"  {k:42s} {v:4d}")


if __name__ == "__main__":
    main()
