#!/bin/bash
# Run-to-run spread of the headline: bench.py as the driver calls it (headline only), ten fresh processes.
set -o pipefail
OUT="${1:-gpurun_out/r5_bench_variance.txt}"
: > "$OUT"
for k in 1 2 3 4 5 6 7 8 9 10; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), f\"{d['value']:.4e}\")" | tee -a "$OUT"
done
python3 - "$OUT" <<'P'
import sys, statistics as st
v=[float(l.split()[0]) for l in open(sys.argv[1]) if l.strip()]
s=f"# ms_per_step over {len(v)} fresh processes: min {min(v):.4f}  median {st.median(v):.4f}  max {max(v):.4f}  (max/min - 1 = {max(v)/min(v)-1:.3%})"
print(s); open(sys.argv[1],'a').write(s+"\n")
P
