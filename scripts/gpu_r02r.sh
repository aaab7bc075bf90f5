#!/bin/bash
# later flat iterations from lists (no whole-raster passes): parity + timing
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_d8.py tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_strips.py tests/test_gpu_multigpu.py tests/test_gpu_fullsize.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error|Error|assert|differ" | tail -6
for d in 1 0; do
if [ $d = 1 ]; then export TDX_FLATS_DENSE=1; else unset TDX_FLATS_DENSE; fi
echo "== dense=$d"
timeout 600 python bench.py --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['flats'])"
done
timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dinf 16384', d['ms_per_step'], d['dinfflowdir_ms'], d['areadinf_ms'])"
