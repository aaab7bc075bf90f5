#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for ms in 8 16 32 96; do
  export TDX_MAX_SWEEPS=$ms
  timeout 900 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bench_16384_$ms.log 2>&1
  tail -1 gpurun_out/bench_16384_$ms.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ms', d['value'], d['stage_ms_per_step'], d['kernel_class_launches_per_step'], d['flats']['pit_rounds'])"
done
