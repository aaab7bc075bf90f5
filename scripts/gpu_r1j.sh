#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
# 1) RCCL code path with a single rank (device tensors through nccl all_reduce; no neighbours)
TDX_BENCH_FORCE_STRIPS=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 1 --size 4096 --cpu-sample 0 2>&1 | tail -2 | cut -c1-600
# 2) kernel trace of the default bench command
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c -o r1 -- python $R/bench.py --cpu-sample 0 > $R/gpurun_out/prof_c.log 2>&1)
find gpurun_out/prof_c -name "*kernel_trace.csv" -delete; find gpurun_out/prof_c -name "*.db" -delete
f=$(find gpurun_out/prof_c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
tail -1 gpurun_out/prof_c.log | cut -c1-400
