#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -5
for mode in hybrid tiles walk; do
  unset TDX_DINF_TILES TDX_DINF_WALK
  if [ $mode = tiles ]; then export TDX_DINF_TILES=1; fi
  if [ $mode = walk ]; then export TDX_DINF_WALK=1; fi
  timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode 16384', d['areadinf_ms'], d['areadinf_classes'], d['areadinf_rounds'])"
done
unset TDX_DINF_WALK; export TDX_DINF_TILES=1
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiles 32768', d['areadinf_ms'], d['areadinf_classes'], d['areadinf_rounds'])"
TDX_DEBUG_ROUNDS=1 timeout 600 python scripts/bench_dinf.py --size 16384 --steps 1 --warmup 0 2>&1 | grep "rounds .*activations" | head -16
unset TDX_DINF_TILES
timeout 600 python scripts/bench_gridnet.py 2>&1 | tail -1 | cut -c1-600
timeout 300 python bench.py --cpu-sample 0 2>&1 | tail -1 | cut -c1-1200
