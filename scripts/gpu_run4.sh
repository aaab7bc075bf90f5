#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 600 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench3_16384.log 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --size 16384 --steps 1 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1)
find gpurun_out/prof1 -name "*kernel_trace.csv" -delete; find gpurun_out/prof1 -name "*.db" -delete
find gpurun_out/prof1 -type f | head
tail -15 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench3_16384.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['kernel_class_launches_per_step'])"
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
