#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
timeout 200 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/b.json; python -c "import sys,json; d=json.loads(open('gpurun_out/b.json').read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['kernel_class_launches_per_step'])"
