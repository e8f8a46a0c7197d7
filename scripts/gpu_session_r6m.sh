#!/bin/bash
# round 6, session m: the final tree (merged launches through the kernel compiled for their sequence of kinds) -- smoke, the whole GPU tier, bench.py as the
# driver runs it, configs[3]'s share (fusion 1 and 5), its kernel trace (50 years) and the three counter passes of the merged plan.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6m_smoke.log 2>&1 || { tail -20 gpurun_out/r6m_smoke.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6m_tests.log 2>&1 || { tail -40 gpurun_out/r6m_tests.log; exit 1; }
tail -n 2 gpurun_out/r6m_tests.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6m_bench_details.json > gpurun_out/r6m_bench.json 2> gpurun_out/r6m_bench.err || { tail -20 gpurun_out/r6m_bench.err; exit 1; }
wc -c gpurun_out/r6m_bench.json
for f in 1 6 5; do
  timeout -k 10 600 python3 scripts/run_configs3_share.py --fusion $f > gpurun_out/r6m_share_fusion$f.json 2> gpurun_out/r6m_share_fusion$f.err || { tail -5 gpurun_out/r6m_share_fusion$f.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/r6m_share_fusion$f.json').read().strip().splitlines()[-1]); print('fusion $f:', round(d['run_s'],4), 's', round(d['ms_per_model_step']*1e3,1), 'us/step', d['launches_per_step'], 'launches/step', all(d['first_64_members_equal_a_64_member_run'].values()))"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6m_share_trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 > "$ROOT/gpurun_out/r6m_share_traced.json" 2> "$ROOT/gpurun_out/r6m_share_traced.err" || { tail -5 "$ROOT/gpurun_out/r6m_share_traced.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d "$ROOT/gpurun_out/r6m_share_pmc_sq" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r6m_pmc_sq.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6m_pmc_sq.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/gpurun_out/r6m_share_pmc_fetch" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r6m_pmc_fetch.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6m_pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$ROOT/gpurun_out/r6m_share_pmc_write" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r6m_pmc_write.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6m_pmc_write.log"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r6m_share_trace 20 > gpurun_out/r6m_configs3_fast_50yr_kernel_table.txt; head -12 gpurun_out/r6m_configs3_fast_50yr_kernel_table.txt
python3 scripts/summarize_share_pmc.py --sq gpurun_out/r6m_share_pmc_sq --fetch gpurun_out/r6m_share_pmc_fetch --write gpurun_out/r6m_share_pmc_write \
    --trace gpurun_out/r6m_share_trace --step-kernels "udeb_kernel,ocean_recur_kernel,group_split_seq_kernel" --min-dispatches 10 \
    --title "configs[3] share, MAGICC graph, FAST, merged launches (3 per step)" --out gpurun_out/r6_configs3_share_merged_pmc.txt > /dev/null || exit 1
cut -c1-230 gpurun_out/r6_configs3_share_merged_pmc.txt
find gpurun_out/r6m_share_trace gpurun_out/r6m_share_pmc_sq gpurun_out/r6m_share_pmc_fetch gpurun_out/r6m_share_pmc_write -name '*.csv' -size +2M -delete
