#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03l}
for w in 4 5 4 5; do
  TDX_DINF_WAVES=$w timeout 90 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf_w$w.json 2>> gpurun_out/${T}.err
  TDX_DINF_WAVES=$w timeout 200 $B dinf -n 32768 -steps 1 > gpurun_out/${T}_dinf_32768_w$w.json 2>> gpurun_out/${T}.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${T}_dinf_w$w.json')); e=json.load(open('gpurun_out/${T}_dinf_32768_w$w.json'))
print('waves $w: 16384:', d['areadinf_ms'], d['crc']['sca'], ' 32768:', e['areadinf_ms'])
"
done
TDX_DINF_WAVES=5 timeout 200 $B decay -nx 65536 -ny 8192 -steps 2 > gpurun_out/${T}_decay_w5.json 2>> gpurun_out/${T}.err
timeout 200 $B decay -nx 65536 -ny 8192 -steps 2 > gpurun_out/${T}_decay_w4.json 2>> gpurun_out/${T}.err
python3 -c "
import json
for w in (4,5):
    d=json.load(open('gpurun_out/${T}_decay_w%d.json' % w)); print('decay strip waves', w, d['ms_per_step'])
"
