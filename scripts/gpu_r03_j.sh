#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03j}
for u in 6000 400 64 16; do
  TDX_DINF_BULK_UNTIL=$u timeout 90 $B dinf -n 16384 -steps 2 > gpurun_out/${T}_dinf_until$u.json 2>> gpurun_out/${T}.err
  TDX_DINF_BULK_UNTIL=$u timeout 200 $B dinf -n 32768 -steps 1 > gpurun_out/${T}_dinf_32768_until$u.json 2>> gpurun_out/${T}.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${T}_dinf_until$u.json')); e=json.load(open('gpurun_out/${T}_dinf_32768_until$u.json'))
print('dinf until $u: 16384:', d['areadinf_ms'], d['areadinf']['rounds'], ' 32768:', e['areadinf_ms'], e['areadinf']['rounds'])
"
done
