#!/bin/bash
# A/B of environment knobs on the 16384^2 bench: gpu_ab.sh "VAR=val ..." "VAR=val ..." ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout 200 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/ab.json
  python -c "import json,sys; d=json.loads(open('gpurun_out/ab.json').read()); print(sys.argv[1], '->', round(d['value']), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, d['kernel_class_launches_per_step'])" "$cfg"
done
