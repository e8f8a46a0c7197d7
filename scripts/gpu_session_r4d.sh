#!/bin/bash
# round 4, session d: the (Ts, Ts - Td) two-layer FAST step and log_f64 in the coupled chain -- GPU tier, deviation from the oracle,
# the round's PMC passes of the hot kernels (profiles/traffic.json points at these), then bench.py as the driver runs it
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4d_tests.log 2>&1 || { tail -40 gpurun_out/r4d_tests.log; exit 1; }
tail -3 gpurun_out/r4d_tests.log
timeout -k 10 600 python scripts/fast_mode_error.py > gpurun_out/r4d_fast_mode_error.log 2>&1 || { tail -20 gpurun_out/r4d_fast_mode_error.log; exit 1; }
cat gpurun_out/r4d_fast_mode_error.log
for spec in "r4_coupled_fast_1e6 1000000 1 1 coupled_fast_kernel" "r4_coupled_1e6 1000000 0 1 coupled_kernel" "r4_exact_1e6 1000000 0 0 two_layer_kernel" \
            "r4_fast_1e6 1000000 1 0 two_layer_kernel" "r4_exact_1e5 100000 0 0 two_layer_kernel"; do
  set -- $spec
  timeout -k 10 600 bash scripts/gpu_profile.sh "$1" "$2" "$3" "$4" > "gpurun_out/$1.profile.log" 2>&1 || { tail -20 "gpurun_out/$1.profile.log"; exit 1; }
  python3 scripts/summarize_profile.py "$1" "gpurun_out/$1.txt" "$5" || exit 1
  grep -E "VALU instructions per wavefront|VALU issue utilisation|HBM traffic|un-profiled average|effective shader clock" "gpurun_out/$1.txt"
  find "gpurun_out/prof_$1" -name "*.csv" -delete
done
timeout -k 10 900 python bench.py > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err || { tail -20 gpurun_out/r4d_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4d_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["roofline"]["frac"], d["cpu_baseline"])
for k, v in d["extra"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, list, str))} if isinstance(v, dict) else v)
PY
