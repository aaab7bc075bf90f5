#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_d8.py tests/test_gpu_large_golden.py tests/test_strips.py tests/test_gpu_fullsize.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error|Error|assert|differ" | tail -6
TDX_AD8_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "ad8_tile_local" | tail -1
timeout 600 python bench.py --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
