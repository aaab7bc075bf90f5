#!/bin/bash
# round-1 GPU pass A: parity tests, bench line, rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --size 4096 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_4096.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_16384.log 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_a -o r1 -- python $R/bench.py --size 16384 --steps 1 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_a.log 2>&1)
find gpurun_out/prof_a -name "*kernel_trace.csv" -delete; find gpurun_out/prof_a -name "*.db" -delete
for f in gpurun_out/bench_4096.log gpurun_out/bench_16384.log; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['kernel_class_launches_per_step'], d['flats'], d.get('cpu_baseline'))"; done
f=$(find gpurun_out/prof_a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
