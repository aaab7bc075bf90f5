#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -15
timeout 300 python bench.py --size 4096 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_4096.log 2>&1
timeout 900 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_16384.log 2>&1
for f in gpurun_out/bench_4096.log gpurun_out/bench_16384.log; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['kernel_class_launches_per_step'], d['flats'])" || tail -5 $f; done
