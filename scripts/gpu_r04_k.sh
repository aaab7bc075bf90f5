#!/bin/bash
# Round 4, call K: A/B of environment knobs with the native harness: gpu_r04_k.sh MODE SIZE "ENV=..." "ENV=... ENV=..." ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
MODE=$1; SIZE=$2; shift; shift
for e in "$@"; do
  for i in 1 2; do
    L=$(env $e timeout 300 taudem_amd/bin/tdxbench $MODE -n $SIZE -steps 2 -crc 2>&1 | tail -n 1)
    echo "$e | $L" >> $O/ab_${MODE}_${SIZE}.txt
    echo "$e | $(echo "$L" | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d.get('ms_per_step'), {k: round(v, 2) for k, v in d.items() if k.endswith('_ms')}, d.get('crc'))
except Exception as ex: print('unparsed', ex)
")"
  done
done
