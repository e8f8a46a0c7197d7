#!/bin/bash
# (a record of round 3: the whole-graph launch -- fusion mode 4 -- and the four-wavefront ClimateUDEB kernel -- variant 4 -- it exercises were removed in round 4)
# round 3, session a: the two-wavefront ClimateUDEB kernel against the one-thread kernel (parity tests + timing per variant)
set -o pipefail
mkdir -p gpurun_out
for v in 0 2; do   # 0: one thread per member; 2: a hemisphere per wavefront (the occupancy-2 variants 1 and 3 of the first session: commit 8ce8fee)
  echo "=== RSCM_UDEB_VARIANT=$v" | tee -a gpurun_out/r3a_udeb.log
  RSCM_UDEB_VARIANT=$v timeout -k 10 300 python -m pytest tests/test_gpu_udeb.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r3a_udeb.log || exit 1
  RSCM_UDEB_VARIANT=$v timeout -k 10 300 python scripts/bench_udeb.py 65536 100000 125000 2>&1 | tee -a gpurun_out/r3a_udeb.log || exit 1
  RSCM_UDEB_VARIANT=$v timeout -k 10 300 python scripts/bench_udeb_steps.py 65536 125000 2>&1 | tee -a gpurun_out/r3a_udeb.log || exit 1
done
