#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dinf 16384', d['ms_per_step'], d['dinfflowdir_ms'], d['dinfflowdir_classes'], d['areadinf_ms'])"
timeout 300 python bench.py --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline_streaming_stencil'])"
