#!/bin/bash
# Round-3 step D: A/B of the slope segment length and the batch policy, set2flat rewrite, full suite
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03d}
timeout 90 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8.json 2> gpurun_out/${T}_d8.err
TDX_SLOPE_SEG=16 timeout 90 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_seg16.json 2>> gpurun_out/${T}_d8.err
TDX_RELAX_LONG_TAIL=1 timeout 90 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_longtail.json 2>> gpurun_out/${T}_d8.err
timeout 90 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf.json 2> gpurun_out/${T}_dinf.err
timeout 200 $B dinf -n 32768 -steps 1 -crc > gpurun_out/${T}_dinf_32768.json 2>> gpurun_out/${T}_dinf.err
for f in gpurun_out/${T}_*.json; do echo "== $f"; python3 -c "
import json,sys
d=json.load(open('$f'))
print({k:v for k,v in d.items() if not isinstance(v,dict)})
for k,v in d.items():
    if isinstance(v,dict): print('  ',k,{a:b for a,b in v.items() if a in ('ms_total','ms_class','rounds')} if 'ms_total' in v else v)
"; done
cat gpurun_out/${T}_d8.err gpurun_out/${T}_dinf.err | tail -n 5
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --timeout=600 --timeout-method=thread --durations=5 2>&1 | tail -n 24 > gpurun_out/${T}_pytest_gpu.txt; tail -n 16 gpurun_out/${T}_pytest_gpu.txt
