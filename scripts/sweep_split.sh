for chunk in 32 48 64 96 128 192; do for first in 65536 49152 57344 73728; do
  v=$(RSCM_SPLIT_CHUNK=$chunk RSCM_SPLIT_FIRST=$first python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4))")
  echo "chunk $chunk first $first: $v ms"
done; done
