#!/bin/bash
# Round 4, call S: the D-infinity sweeps after a change of the walk - CRCs and times at 4096^2 / 16384^2 (/ 32768^2 with "big"), the decay strip, then the D-infinity tests
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
bash scripts/gpu_r04_k.sh dinf 4096 "X=1" 2>&1 | tee gpurun_out/r04s/dinf_4096.txt
bash scripts/gpu_r04_k.sh dinf 16384 "X=1" 2>&1 | tee gpurun_out/r04s/dinf_16384.txt
if [ "$1" = "big" ]; then bash scripts/gpu_r04_k.sh dinf 32768 "X=1" 2>&1 | tee gpurun_out/r04s/dinf_32768.txt; fi
L=$(timeout 300 taudem_amd/bin/tdxbench decay -steps 1 -crc 2>&1 | tail -n 1); echo "$L" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('decay', d.get('ms_per_step'), d.get('crc'))" | tee gpurun_out/r04s/decay.txt
timeout 900 python -m pytest tests/test_gpu_dinf.py tests/test_flowalg.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -n 4 | tee gpurun_out/r04s/pytest.txt
