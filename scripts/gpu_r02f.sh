#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
TDX_DINF_TILES=1 timeout 1200 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
TDX_DINF_TILES=1 timeout 900 python -m pytest tests/test_strips.py -m gpu -q --no-header -p no:cacheprovider -x -k dinf 2>&1 | tail -3
for mode in hybrid tiles; do
  if [ $mode = tiles ]; then export TDX_DINF_TILES=1; else unset TDX_DINF_TILES; fi
  timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode 16384', d['areadinf_ms'], d['areadinf_classes'], d['areadinf_rounds'])"
done
export TDX_DINF_TILES=1
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiles 32768', d['areadinf_ms'], d['areadinf_classes'], d['areadinf_rounds'])"
bash scripts/gpu_r02c.sh 2>&1 | head -12
python - <<'PY'
import re
rows=[]
for ln in open('gpurun_out/r02c_dinf_round_times.txt'):
    m=re.match(r"\s+(\d+)\s+(\d+) tiles\s+([\d.]+) us",ln)
    if m: rows.append((int(m.group(1)),int(m.group(2)),float(m.group(3))))
for a,b in ((0,1),(1,3),(3,10),(10,30),(30,100),(100,300),(300,700)):
    r=[x for x in rows if a<=x[0]<b]; t=sum(x[2] for x in r)/1e3
    print(f"rounds {a}-{b}: {t:.1f} ms, {sum(x[1] for x in r)} activations, {t*1e3/max(1,b-a):.0f} us/round")
PY
