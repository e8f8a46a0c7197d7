for c0 in 64 128 750; do for c1 in 24 32 48 64 96; do
# NEEDS the experiments build (make -C rscm_amd/csrc EXPERIMENTS=1; plain `make` afterwards): the shipped library does not read these knobs (csrc/experiment_env.hpp)
  v=$(RSCM_SPLIT_CHUNK=$c0 RSCM_SPLIT_CHUNK2=$c1 python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4))")
  echo "chunk0 $c0 chunk1 $c1: $v ms"
done; done
