#!/bin/bash
# Round-3 step E: the D-infinity bulk geometry on angles (on-the-fly proportions) at 6 / 5 tiles per CU vs the precomputed proportions
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03e}
for v in 6 5 p; do
  TDX_DINF_BULK=$v timeout 90 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf_$v.json 2>> gpurun_out/${T}.err
  TDX_DINF_BULK=$v timeout 200 $B dinf -n 32768 -steps 1 -crc > gpurun_out/${T}_dinf_32768_$v.json 2>> gpurun_out/${T}.err
done
TDX_SWEEP_VERIFY=2 timeout 90 $B dinf -n 16384 -steps 1 -warmup 0 > /dev/null 2>> gpurun_out/${T}.err
timeout 200 $B decay -nx 65536 -ny 8192 -steps 1 -crc > gpurun_out/${T}_decay_6.json 2>> gpurun_out/${T}.err
TDX_DINF_BULK=p timeout 200 $B decay -nx 65536 -ny 8192 -steps 1 -crc > gpurun_out/${T}_decay_p.json 2>> gpurun_out/${T}.err
for f in gpurun_out/${T}_*.json; do echo "== $f"; python3 -c "
import json
d=json.load(open('$f'))
print({k:v for k,v in d.items() if not isinstance(v,dict)})
for k,v in d.items():
    if isinstance(v,dict): print('  ',k,{a:b for a,b in v.items() if a in ('ms_total','ms_class','rounds')} if 'ms_total' in v else v)
"; done
tail -n 6 gpurun_out/${T}.err
timeout 900 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_fullsize.py tests/test_flowalg.py tests/test_gpu_multigpu.py -m gpu -q --no-header -p no:cacheprovider -x --timeout=600 --timeout-method=thread 2>&1 | tail -n 8
