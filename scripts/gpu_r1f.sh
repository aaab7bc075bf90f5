#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_16384.log 2>&1; tail -1 gpurun_out/bench_16384.log | cut -c1-3000
TDX_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --size 4096 > gpurun_out/bench_2rank_gloo.log 2>&1; tail -2 gpurun_out/bench_2rank_gloo.log | cut -c1-2500
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b -o r1 -- python $R/bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_b.log 2>&1)
find gpurun_out/prof_b -name "*kernel_trace.csv" -delete; find gpurun_out/prof_b -name "*.db" -delete
f=$(find gpurun_out/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-260
