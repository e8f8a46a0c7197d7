#!/bin/bash
# (round 5 experiment: RSCM_QUEUE_RUNS / RSCM_QUEUE_WAVES / RSCM_QUEUE_CHUNK drove the work-queue launch, which was removed again -- the
# queue lines below now measure the default plan; results of the experiment: profiles/r5_queue_experiment.txt)
# The work-queue launch of the two-layer kind against the two-stream cut and the plain launch: ms per pass of the default bench
# (1e5 members x 750 years, EXACT) over task lengths and resident wavefronts per SIMD, then other sizes.
set -o pipefail
OUT="${1:-gpurun_out/r5_sweep_queue.txt}"
one() { python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 "${@}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['roofline'].get('tasks_per_pass'), d['roofline']['launches_per_pass'], d['check']['failed_members_rank0'])"; }
{
echo "# ms_per_step kernel_ms tasks launches failed   (bench.py --no-extra --steps 30 --warmup 5)"
echo "plain launch (RSCM_QUEUE_RUNS=0 RSCM_SPLIT_RUNS=0): $(RSCM_QUEUE_RUNS=0 RSCM_SPLIT_RUNS=0 one)"
echo "two-stream cut (RSCM_QUEUE_RUNS=0): $(RSCM_QUEUE_RUNS=0 one)"
for w in 1 2 3 4; do for c in 10 15 25 40 64; do
  echo "queue waves/SIMD=$w chunk=$c: $(RSCM_QUEUE_WAVES=$w RSCM_QUEUE_CHUNK=$c one)"
done; done
for m in 70000 125000 200000 1000000; do
  echo "members=$m cut: $(RSCM_QUEUE_RUNS=0 one --members $m)"
  echo "members=$m queue (default): $(one --members $m)"
done
echo "fast mode cut: $(RSCM_QUEUE_RUNS=0 one --mode fast)"
echo "fast mode queue: $(one --mode fast)"
} | tee "$OUT"
