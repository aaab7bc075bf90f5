#!/bin/bash
# Round 4, call T: A/B of environment knobs on the flow-algebra tools: gpu_r04_t.sh SIZE TOOLS "ENV=.." ...
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t
SIZE=$1; TOOLS=$2; shift; shift
for e in "$@"; do
  L=$(env $e timeout 600 python scripts/bench_flowalg.py --size $SIZE --only $TOOLS 2>/dev/null | tail -n 1)
  echo "$e | $(echo "$L" | python -c "
import sys, json
try: print({k: round(v, 1) for k, v in json.loads(sys.stdin.read())['ms'].items()})
except Exception as ex: print('unparsed', ex)")" | tee -a gpurun_out/r04t/ab_${SIZE}.txt
done
