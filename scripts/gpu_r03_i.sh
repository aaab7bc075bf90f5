#!/bin/bash
# Round-3 step I: single call site of the tile body (register allocation), per-policy geometry hand-over; tools + both pipelines
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03i}
timeout 90 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8.json 2> gpurun_out/${T}.err
timeout 90 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf.json 2>> gpurun_out/${T}.err
timeout 200 $B dinf -n 32768 -steps 1 -crc > gpurun_out/${T}_dinf_32768.json 2>> gpurun_out/${T}.err
for f in d8 dinf dinf_32768; do python3 -c "
import json
d=json.load(open('gpurun_out/${T}_$f.json'))
print({k:v for k,v in d.items() if not isinstance(v,dict)}, d['crc'])
"; done
timeout 300 python scripts/bench_flowalg.py 2>> gpurun_out/${T}.err | tail -n 1 > gpurun_out/${T}_bench_flowalg_16384.json
python3 -c "
import json
d=json.load(open('gpurun_out/${T}_bench_flowalg_16384.json'))
print('default', {k: round(v,1) for k,v in d['ms'].items()})
"
for u in 16 1; do
TDX_D8_BULK_UNTIL=$u timeout 300 python scripts/bench_flowalg.py 2>> gpurun_out/${T}.err | tail -n 1 > gpurun_out/${T}_flowalg_until$u.json
python3 -c "
import json
d=json.load(open('gpurun_out/${T}_flowalg_until$u.json'))
print('until $u', {k: round(v,1) for k,v in d['ms'].items()})
"
done
tail -n 2 gpurun_out/${T}.err
timeout 900 python -m pytest tests/test_flowalg.py tests/test_gpu_gridnet.py tests/test_gpu_dinf.py tests/test_gpu_d8.py tests/test_gpu_fullsize.py -m gpu -q --no-header -p no:cacheprovider -x --timeout=600 --timeout-method=thread 2>&1 | tail -n 4
