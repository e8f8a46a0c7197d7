"""One character per instruction of a basic block of a kernel's ISA (make -C rscm_amd/csrc asm-*), in program order:
. VALU  R v_rcp_f64  a v_accvgpr_*  L LDS  M scalar load  W s_waitcnt  s other scalar  G global/scratch memory
    python3 scripts/isa_trace.py rscm_amd/csrc/build/udeb.s <kernel label prefix> <block label, e.g. .LBB16_115>"""
import re
import sys

path, kernel, block = sys.argv[1:4]
lines = open(path).read().split("\n")
start = next(k for k, l in enumerate(lines) if l.startswith(kernel) and l.split(":")[0].startswith(kernel) and ":" in l)
blk = next(k for k in range(start, len(lines)) if lines[k].startswith(block + ":"))
end = next(k for k in range(blk + 1, len(lines)) if re.match(r"^\.LBB\d+_\d+:", lines[k]) or lines[k].startswith(".Lfunc_end"))


def klass(op):
    if op.startswith("ds_"): return "L"
    if op.startswith("s_waitcnt"): return "W"
    if op.startswith(("s_load", "s_buffer_load")): return "M"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("v_rcp_f64"): return "R"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "G"
    if op.startswith("v_"): return "."
    return "s"


body = [l.strip() for l in lines[blk + 1:end] if l.strip() and not l.strip().startswith((";", "."))]
trace = "".join(klass(l.split()[0]) for l in body)
print(len(body), "instructions")
for k in range(0, len(trace), 150):
    print(trace[k:k + 150])
