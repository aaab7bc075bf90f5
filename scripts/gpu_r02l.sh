#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_gpu_cli.py tests/test_gpu_fullsize.py -m gpu -q --no-header -p no:cacheprovider -x --deselect tests/test_gpu_cli.py::test_bigtiff_above_4gb_round_trip 2>&1 | grep -E "passed|failed|error|Error|assert|differ" | tail -6
timeout 900 python -m pytest tests/test_strips.py tests/test_gpu_multigpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "dinf or cli_gpus" 2>&1 | tail -3
for u in 6000 0 2000 20000; do
TDX_DINF_BULK_UNTIL=$u timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('until $u: 16384', d['areadinf_ms'], d['areadinf_classes'], d['areadinf_rounds'])"
done
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('32768', d['ms_per_step'], d['dinfflowdir_ms'], d['areadinf_ms'], d['areadinf_rounds'])"
