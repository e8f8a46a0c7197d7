#!/bin/bash
# sharded device sampler rehearsal: 2 and 4 ranks on GPU 0 over gloo
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
export RSCM_BENCH_BACKEND=gloo RSCM_BENCH_DEVICE=0
for R in 2 4; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 2952$R \
    scripts/rehearse_sharded_sampler.py --out gpurun_out/sharded_sampler_$R > gpurun_out/sharded_sampler_$R.log 2> gpurun_out/sharded_sampler_$R.err || { tail -30 gpurun_out/sharded_sampler_$R.err; exit 1; }
tail -c 2500 gpurun_out/sharded_sampler_$R.log; echo
done
