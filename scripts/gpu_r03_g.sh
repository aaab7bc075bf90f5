#!/bin/bash
# Round-3 step G: ready check on heads only in the lockstep sweeps - every accumulation tool at 16384^2, AreaDinf at both sizes
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03g}
timeout 90 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf.json 2> gpurun_out/${T}.err
timeout 200 $B dinf -n 32768 -steps 1 -crc > gpurun_out/${T}_dinf_32768.json 2>> gpurun_out/${T}.err
for f in gpurun_out/${T}_dinf.json gpurun_out/${T}_dinf_32768.json; do python3 -c "
import json
d=json.load(open('$f'))
print({k:v for k,v in d.items() if not isinstance(v,dict)}, d['crc'], d['areadinf']['rounds'])
"; done
timeout 800 python scripts/bench_flowalg.py 2>> gpurun_out/${T}.err | tail -n 1 > gpurun_out/${T}_bench_flowalg_16384.json; cut -c1-900 gpurun_out/${T}_bench_flowalg_16384.json
timeout 600 python scripts/bench_gridnet.py 2>> gpurun_out/${T}.err | tail -n 1 > gpurun_out/${T}_bench_gridnet_16384.json; cut -c1-400 gpurun_out/${T}_bench_gridnet_16384.json
tail -n 3 gpurun_out/${T}.err
timeout 900 python -m pytest tests/test_flowalg.py tests/test_gpu_gridnet.py tests/test_gpu_dinf.py tests/test_gpu_fullsize.py -m gpu -q --no-header -p no:cacheprovider -x --timeout=600 --timeout-method=thread 2>&1 | tail -n 4
