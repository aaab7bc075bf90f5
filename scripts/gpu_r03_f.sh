#!/bin/bash
# Round-3 step F: next-tile prefetch of the relaxation kernels, A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03f}
for v in 1 0 1 0; do
  TDX_RELAX_PREFETCH=$v timeout 90 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_pf$v.json 2>> gpurun_out/${T}.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${T}_d8_pf$v.json'))
print('prefetch $v', {k:v for k,v in d.items() if not isinstance(v,dict)}, d['crc'], d['pitremove']['ms_class'][1], d['d8flowdir']['ms_class'][2])
"
done
TDX_DEBUG_ROUNDS=1 timeout 90 $B d8 -n 16384 -steps 1 -warmup 0 2>&1 >/dev/null | grep -A2 "65536 tiles" | cut -c1-200 | tail -n 12
tail -n 3 gpurun_out/${T}.err
timeout 600 python -m pytest tests/test_gpu_d8.py tests/test_gpu_large_golden.py tests/test_gpu_dinf.py -m gpu -q --no-header -p no:cacheprovider -x --timeout=300 --timeout-method=thread 2>&1 | tail -n 4
