#!/bin/bash
# round 6, session b: the rebuilt library (experiment knobs compiled out, table sizes tied to capacities) through the whole GPU tier;
# then what one model step of configs[3] is made of: every component as its OWN launch (fusion 0) and the light segments unsplit
# (fusion 4) under the kernel trace, and the EXACT share's counter passes at full length (OceanCarbon's O(T^2) history: its bytes
# per step grow with the step index, so no short run stands for it).  Each step once.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6b_smoke.log 2>&1 || { tail -20 gpurun_out/r6b_smoke.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6b_tests.log 2>&1 || { tail -40 gpurun_out/r6b_tests.log; exit 1; }
tail -n 2 gpurun_out/r6b_tests.log
cd /tmp
for f in 0 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6b_share_trace_fusion$f" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 10 --fusion $f --no-anchor > "$ROOT/gpurun_out/r6b_share_fusion$f.json" 2> "$ROOT/gpurun_out/r6b_share_fusion$f.err" || { tail -5 "$ROOT/gpurun_out/r6b_share_fusion$f.err"; exit 1; }
  (cd "$ROOT" && python3 scripts/trace_table.py gpurun_out/r6b_share_trace_fusion$f 100000 > gpurun_out/r6b_share_fusion${f}_table.txt && head -24 gpurun_out/r6b_share_fusion${f}_table.txt)
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d "$ROOT/gpurun_out/r6b_exact_pmc_sq" -- python3 "$ROOT/scripts/run_configs3_share.py" --exact --no-anchor > "$ROOT/gpurun_out/r6b_exact_pmc_sq.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6b_exact_pmc_sq.log"; exit 1; }
echo "exact sq pass done"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/gpurun_out/r6b_exact_pmc_fetch" -- python3 "$ROOT/scripts/run_configs3_share.py" --exact --no-anchor > "$ROOT/gpurun_out/r6b_exact_pmc_fetch.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6b_exact_pmc_fetch.log"; exit 1; }
echo "exact fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$ROOT/gpurun_out/r6b_exact_pmc_write" -- python3 "$ROOT/scripts/run_configs3_share.py" --exact --no-anchor > "$ROOT/gpurun_out/r6b_exact_pmc_write.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6b_exact_pmc_write.log"; exit 1; }
echo "exact write pass done"
cd "$ROOT"
python3 scripts/summarize_share_pmc.py --sq gpurun_out/r6b_exact_pmc_sq --fetch gpurun_out/r6b_exact_pmc_fetch --write gpurun_out/r6b_exact_pmc_write \
    --step-kernels "udeb_kernel,ocean,group_split_kernel,group_kernel_args" --title "configs[3] share, MAGICC graph, EXACT, all 9000 steps" \
    --out gpurun_out/r6_configs3_share_exact_pmc.txt > /dev/null || exit 1
cut -c1-260 gpurun_out/r6_configs3_share_exact_pmc.txt
find gpurun_out/r6b_* -name '*.csv' -size +2M -delete
du -sh gpurun_out/r6b_*
