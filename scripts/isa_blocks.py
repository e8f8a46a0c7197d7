"""Static view of a kernel's ISA (make -C rscm_amd/csrc asm-*): per basic block the instruction count by class,
largest blocks first -- where the unrolled loops are and what they are made of.
    python3 scripts/isa_blocks.py rscm_amd/csrc/build/udeb.s <substring of the mangled kernel name> [n_blocks]"""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 8


def klass(op):
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("v_rcp_f64"): return "rcp64"
    if op.startswith("v_") and "f64" in op: return "f64"
    if op.startswith("v_"): return "valu32"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"


inside, blocks, cur = False, [], None
for line in open(path):
    s = line.strip()
    if not inside:
        if re.match(r"^_Z\w+:", s) and want in s:
            inside = True
            cur = ["entry", collections.Counter(), 0]
        continue
    if s.startswith(".Lfunc_end"):
        break
    if re.match(r"^\.LBB\d+_\d+:", s):
        blocks.append(cur)
        cur = [s.split(":")[0], collections.Counter(), 0]
        continue
    if not s or s.startswith((";", ".", "//")):
        continue
    op = s.split()[0]
    cur[1][klass(op)] += 1
    cur[2] += 1
blocks.append(cur)
total = collections.Counter()
for b in blocks:
    total.update(b[1])
print(f"{want}: {len(blocks)} blocks, {sum(total.values())} instructions: " + ", ".join(f"{k} {v}" for k, v in total.most_common()))
for b in sorted(blocks, key=lambda b: -b[2])[:top]:
    print(f"  {b[0]:<14} {b[2]:5d}: " + ", ".join(f"{k} {v}" for k, v in b[1].most_common()))
