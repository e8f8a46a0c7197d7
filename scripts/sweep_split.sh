#!/bin/bash
# NEEDS the experiments build (make -C rscm_amd/csrc EXPERIMENTS=1; plain `make` afterwards): the shipped library does not read these knobs (csrc/experiment_env.hpp)
# the member split's two knobs (rscm_gpu.cpp, plan_member_split) against the headline: RSCM_SPLIT_CHUNK (model steps per launch) and
# RSCM_SPLIT_FIRST (members of the first block); MODE=fast for the FAST arithmetic
MODE="${MODE:-exact}"
for chunk in ${CHUNKS:-32 48 64 96 128 192}; do for first in ${FIRSTS:-65536 57344}; do
  v=$(RSCM_SPLIT_CHUNK=$chunk RSCM_SPLIT_FIRST=$first python bench.py --mode $MODE --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4))")
  echo "mode $MODE chunk $chunk first $first: $v ms"
done; done
