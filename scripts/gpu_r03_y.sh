#!/bin/bash
# fused rounds for the two level fields (one launch per round for both) against the two-stream form: canary, timings, flats-related tests
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03y}
crc() { python3 -c "
import json,sys
print(json.load(open(sys.argv[1]))['crc'])" $1 2>/dev/null; }
line() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
ok = d['crc']=={'fel': 3868594109, 'p': 3675299354, 'sd8': 3615701311, 'ad8': 2167656781}
print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'pit', d['pitremove_ms'], 'd8', d['d8flowdir_ms'], 'ad8', d['aread8_ms'], 'classes', d['d8flowdir']['ms_class'][:4], 'rounds', d['d8flowdir']['rounds'], 'CRC_OK' if ok else 'CRC_MISMATCH')
" $1; }
timeout 40 $B d8 -n 4096 -steps 1 -crc > gpurun_out/${T}_c.json 2>> gpurun_out/${T}.err || { echo "CANARY FAILED (rc $?)"; tail -n 5 gpurun_out/${T}.err; exit 1; }
P=$(crc gpurun_out/${T}_c.json); echo "canary crc $P"
[ "$P" == "{'fel': 562431989, 'p': 3564740248, 'sd8': 2291047274, 'ad8': 676898407}" ] || { echo "CANARY CRC MISMATCH"; exit 1; }
for i in 1 2; do
timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_fused_$i.json 2>> gpurun_out/${T}.err || { echo FAILED; exit 1; }; line gpurun_out/${T}_fused_$i.json
TDX_FLATS_TWO_STREAMS=1 timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_streams_$i.json 2>> gpurun_out/${T}.err; line gpurun_out/${T}_streams_$i.json
done
timeout 90 $B dinf -n 16384 -steps 3 -crc > gpurun_out/${T}_dinf.json 2>> gpurun_out/${T}.err; python3 -c "
import json; d=json.load(open('gpurun_out/${T}_dinf.json')); print('dinf', d['ms_per_step'], d['dinfflowdir_ms'], d['areadinf_ms'], d['crc'])"
timeout 500 python -m pytest tests/test_gpu_d8.py tests/test_gpu_large_golden.py tests/test_gpu_dinf.py tests/test_gpu_multigpu.py -m gpu -q --no-header -p no:cacheprovider -x --timeout=120 --timeout-method=thread 2>&1 | grep -E "passed|failed|error|Error" | tail -n 5
tail -n 3 gpurun_out/${T}.err
