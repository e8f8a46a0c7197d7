#!/bin/bash
# round 6, session a: the compact bench line on hardware (the multi-rank / bench tests, then bench.py exactly as the driver runs it),
# then the configs[3] counter passes that aborted in round 5 -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes this time (3 + 2 of
# the 4 TCC slots: one pass cannot hold both), the program directly after `--`.  Each step once; a failing step ends the session.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_multirank_gpu.py tests/test_bench_line.py -x -q -m "gpu or not gpu" --deselect "tests/test_multirank_gpu.py::test_sharded_sampler_reproduces_the_single_rank_chain[4]" > gpurun_out/r6a_tests.log 2>&1 || { tail -60 gpurun_out/r6a_tests.log; exit 1; }
tail -n 2 gpurun_out/r6a_tests.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6a_bench_details.json > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err || { tail -20 gpurun_out/r6a_bench.err; exit 1; }
wc -c gpurun_out/r6a_bench.json gpurun_out/r6a_bench_details.json
cat gpurun_out/r6a_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r6a_share_trace" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 50 > "$ROOT/gpurun_out/r6a_share_traced.json" 2> "$ROOT/gpurun_out/r6a_share_traced.err" || { tail -5 "$ROOT/gpurun_out/r6a_share_traced.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d "$ROOT/gpurun_out/r6a_share_pmc_sq" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r6a_pmc_sq.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6a_pmc_sq.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/gpurun_out/r6a_share_pmc_fetch" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r6a_pmc_fetch.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6a_pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$ROOT/gpurun_out/r6a_share_pmc_write" -- python3 "$ROOT/scripts/run_configs3_share.py" --years 4 > "$ROOT/gpurun_out/r6a_pmc_write.log" 2>&1 || { tail -5 "$ROOT/gpurun_out/r6a_pmc_write.log"; exit 1; }
cd "$ROOT"
python3 scripts/trace_table.py gpurun_out/r6a_share_trace 20 > gpurun_out/r6_configs3_fast_50yr_kernel_table.txt; head -14 gpurun_out/r6_configs3_fast_50yr_kernel_table.txt
python3 scripts/summarize_share_pmc.py --sq gpurun_out/r6a_share_pmc_sq --fetch gpurun_out/r6a_share_pmc_fetch --write gpurun_out/r6a_share_pmc_write \
    --trace gpurun_out/r6a_share_trace --out gpurun_out/r6_configs3_share_pmc.txt > /dev/null || exit 1
cut -c1-260 gpurun_out/r6_configs3_share_pmc.txt
# the two ClimateUDEB kernels that had no counter pass of their own: the two-wavefront kernel at 32 768 members, the LDS-c' kernel at 65 layers
bash scripts/gpu_profile.sh r6_udeb2_32768 32768 0 2 > gpurun_out/r6a_prof1.log 2>&1 || { tail -20 gpurun_out/r6a_prof1.log; exit 1; }
UDEB_LAYERS=65 bash scripts/gpu_profile.sh r6_udeb_lds65_65536 65536 0 2 > gpurun_out/r6a_prof2.log 2>&1 || { tail -20 gpurun_out/r6a_prof2.log; exit 1; }
python3 scripts/summarize_profile.py r6_udeb2_32768 gpurun_out/r6_udeb2_32768.txt udeb2_kernel | tail -8
python3 scripts/summarize_profile.py r6_udeb_lds65_65536 gpurun_out/r6_udeb_lds_65.txt udeb2_lds_kernel | tail -8
find gpurun_out/r6a_share_trace gpurun_out/r6a_share_pmc_sq gpurun_out/r6a_share_pmc_fetch gpurun_out/r6a_share_pmc_write gpurun_out/prof_r6_udeb* -name '*.csv' -size +2M -delete
du -sh gpurun_out/r6a_share_* gpurun_out/prof_r6_udeb*
