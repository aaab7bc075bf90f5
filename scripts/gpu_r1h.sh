#!/bin/bash
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_cli.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -5
TDX_DEBUG_ROUNDS=1 timeout 60 python bench.py --size 4096 --steps 1 --warmup 1 --cpu-sample 0 2>&1 | grep "tile_relax_run\|gave up" | tail -8
timeout 100 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['kernel_class_launches_per_step'])"
