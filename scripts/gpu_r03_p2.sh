#!/bin/bash
# canary (4096^2 CRCs against the known values), then timings of the D8 and D-infinity pipelines
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03q}
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:v for k,v in d.items() if not isinstance(v,dict)}, d.get('crc'), [d[k]['ms_class'] for k in d if isinstance(d[k],dict) and 'ms_class' in d[k]], [d[k]['rounds'] for k in d if isinstance(d[k],dict) and 'rounds' in d[k]])
" $1; }
crc() { python3 -c "
import json,sys
print(json.load(open(sys.argv[1]))['crc'])" $1 2>/dev/null; }
timeout 40 $B d8 -n 4096 -steps 1 -crc > gpurun_out/${T}_c.json 2>> gpurun_out/${T}.err || { echo "CANARY FAILED (rc $?)"; tail -n 5 gpurun_out/${T}.err; exit 1; }
P=$(crc gpurun_out/${T}_c.json); echo "canary crc $P"
[ "$P" == "{'fel': 562431989, 'p': 3564740248, 'sd8': 2291047274, 'ad8': 676898407}" ] || { echo "CANARY CRC MISMATCH"; exit 1; }
for i in 1 2; do timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_$i.json 2>> gpurun_out/${T}.err || { echo FAILED; exit 1; }; show gpurun_out/${T}_d8_$i.json; done
timeout 60 $B dinf -n 16384 -steps 3 -crc > gpurun_out/${T}_dinf.json 2>> gpurun_out/${T}.err || { echo FAILED; exit 1; }; show gpurun_out/${T}_dinf.json
for e in "$2" "$3" "$4"; do
  [ -n "$e" ] || continue
  env $e timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_$e.json 2>> gpurun_out/${T}.err; show gpurun_out/${T}_d8_$e.json
done
tail -n 5 gpurun_out/${T}.err
