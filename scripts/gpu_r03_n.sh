#!/bin/bash
# plain-tile form of the level operator + v_med3 PitRemove operator: canary first (a hang costs GPU minutes), then CRCs, A/B against the
# masked form, the flats-related tests
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03n}
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:v for k,v in d.items() if not isinstance(v,dict)}, d.get('crc'), [d[k]['ms_class'] for k in d if isinstance(d[k],dict) and 'ms_class' in d[k]], [d[k]['rounds'] for k in d if isinstance(d[k],dict) and 'rounds' in d[k]])
" $1; }
crc() { python3 -c "
import json,sys
print(json.load(open(sys.argv[1]))['crc'])" $1 2>/dev/null; }
# canary: 4096^2, plain (default) against masked
TDX_FLATS_MASKED=1 timeout 40 $B d8 -n 4096 -steps 1 -crc > gpurun_out/${T}_c_masked.json 2>> gpurun_out/${T}.err
timeout 40 $B d8 -n 4096 -steps 1 -crc > gpurun_out/${T}_c_plain.json 2>> gpurun_out/${T}.err || { echo "CANARY FAILED (rc $?)"; tail -n 5 gpurun_out/${T}.err; exit 1; }
TDX_FLATS_MASKED=2 timeout 40 $B d8 -n 4096 -steps 1 -crc > gpurun_out/${T}_c_half.json 2>> gpurun_out/${T}.err || { echo "CANARY (half) FAILED"; exit 1; }
A=$(crc gpurun_out/${T}_c_masked.json); P=$(crc gpurun_out/${T}_c_plain.json); H=$(crc gpurun_out/${T}_c_half.json)
echo "canary crc masked $A"; echo "canary crc plain  $P"; echo "canary crc half   $H"
[ "$A" == "$P" ] && [ "$A" == "$H" ] || { echo "CANARY CRC MISMATCH"; exit 1; }
for i in 1 2; do timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_$i.json 2>> gpurun_out/${T}.err || { echo FAILED; exit 1; }; show gpurun_out/${T}_d8_$i.json; done
TDX_FLATS_MASKED=1 timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_masked.json 2>> gpurun_out/${T}.err; show gpurun_out/${T}_d8_masked.json
timeout 60 $B dinf -n 16384 -steps 3 -crc > gpurun_out/${T}_dinf.json 2>> gpurun_out/${T}.err || { echo FAILED; exit 1; }; show gpurun_out/${T}_dinf.json
timeout 90 $B dinf -n 32768 -steps 2 -crc > gpurun_out/${T}_dinf_32768.json 2>> gpurun_out/${T}.err; show gpurun_out/${T}_dinf_32768.json
timeout 400 python -m pytest tests/test_gpu_d8.py tests/test_gpu_large_golden.py tests/test_gpu_dinf.py tests/test_gpu_multigpu.py -m gpu -q --no-header -p no:cacheprovider -x --timeout=120 --timeout-method=thread 2>&1 | tail -n 4
tail -n 5 gpurun_out/${T}.err
