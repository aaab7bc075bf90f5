#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8
TDX_DEBUG_ROUNDS=1 timeout 300 python bench.py --size 4096 --steps 1 --warmup 1 --cpu-sample 0 2>&1 | grep "tile_relax_run\|gave up" | tail -8
timeout 900 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_16384.log 2>&1
TDX_RELAX_ROUNDS=1 timeout 900 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_16384_rounds.log 2>&1
for f in gpurun_out/bench_16384.log gpurun_out/bench_16384_rounds.log; do grep "gave up" $f | head -2; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['kernel_class_launches_per_step'])" || tail -5 $f; done
