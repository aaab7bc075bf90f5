#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/gpu_exp_ad8.py > gpurun_out/exp_ad8.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --size 16384 --steps 1 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1)
find gpurun_out/prof1 -name "*stats*" | head; 
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" > gpurun_out/prof1_kernel_stats_head.csv
# keep only the small summaries
find gpurun_out/prof1 -name "*kernel_trace.csv" -delete; find gpurun_out/prof1 -name "*.db" -delete
cat gpurun_out/exp_ad8.log; tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/prof1_kernel_stats_head.csv
