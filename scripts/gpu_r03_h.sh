#!/bin/bash
# Round-3 step H: where the sweeps hand over from 32 x 32 to 64 x 64 tiles (active tiles per round), per tool
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03h}
for u in 1500 400 100; do
  TDX_D8_BULK_UNTIL=$u TDX_DINF_BULK_UNTIL=$u timeout 300 python scripts/bench_flowalg.py 2>> gpurun_out/${T}.err | tail -n 1 > gpurun_out/${T}_flowalg_until$u.json
  python3 -c "
import json
d=json.load(open('gpurun_out/${T}_flowalg_until$u.json'))
print('until $u', {k: round(v,1) for k,v in d['ms'].items()})
"
done
for u in 12000 3000 1500; do
  TDX_DINF_BULK_UNTIL=$u timeout 90 $B dinf -n 16384 -steps 2 > gpurun_out/${T}_dinf_until$u.json 2>> gpurun_out/${T}.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${T}_dinf_until$u.json'))
print('dinf until $u', d['areadinf_ms'], d['areadinf']['rounds'])
"
done
tail -n 3 gpurun_out/${T}.err
