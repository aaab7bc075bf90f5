#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r03k}
for w in 4 5 4 5; do
TDX_D8_WAVES=$w timeout 300 python scripts/bench_flowalg.py 2>> gpurun_out/${T}.err | tail -n 1 > gpurun_out/${T}_flowalg_waves$w.json
python3 -c "
import json
d=json.load(open('gpurun_out/${T}_flowalg_waves$w.json'))
print('waves $w', {k: round(v,1) for k,v in d['ms'].items()})
"
done
