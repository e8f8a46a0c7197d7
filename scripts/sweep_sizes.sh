#!/bin/bash
# (round 5 experiment: RSCM_QUEUE_RUNS / RSCM_QUEUE_WAVES / RSCM_QUEUE_CHUNK drove the work-queue launch, which was removed again -- the
# queue lines below now measure the default plan; results of the experiment: profiles/r5_queue_experiment.txt)
# How fast is ONE wavefront alone on a SIMD?  Plain launches (no cut, no queue) at sizes that put exactly 0.5 / 1 / 1.5 / 2 / 3 / 4
# wavefronts on every SIMD, EXACT and FAST: ms per pass of bench.py's workload.
set -o pipefail
OUT="${1:-gpurun_out/r5_sweep_sizes.txt}"
one() { python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 "${@}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['roofline'].get('tasks_per_pass'), d['roofline']['launches_per_pass'])"; }
{
echo "# members: ms_per_step kernel_ms tasks launches   (bench.py --no-extra --steps 30 --warmup 5; plain = RSCM_QUEUE_RUNS=0 RSCM_SPLIT_RUNS=0)"
for m in 32768 65536 98304 131072 196608 262144; do
  echo "exact plain members=$m: $(RSCM_QUEUE_RUNS=0 RSCM_SPLIT_RUNS=0 one --members $m)"
done
for m in 65536 131072 262144; do
  echo "fast plain members=$m: $(RSCM_QUEUE_RUNS=0 RSCM_SPLIT_RUNS=0 one --members $m --mode fast)"
done
for w in 1 2; do for m in 100000 131072 200000 400000 1000000; do
  echo "exact queue waves=$w chunk=40 members=$m: $(RSCM_QUEUE_WAVES=$w RSCM_QUEUE_CHUNK=40 one --members $m)"
done; done
for m in 131072 400000; do echo "exact cut members=$m: $(RSCM_QUEUE_RUNS=0 one --members $m)"; done
} | tee "$OUT"
