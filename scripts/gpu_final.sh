#!/bin/bash
# Closing run of a round: the full GPU suite, smoke, then the evidence set (bench line, kernel stats, PMC passes, one-step timeline, CRCs of the native harness)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05z}
timeout 1700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=1200 --timeout-method=thread --durations=8 2>&1 | tail -n 20 > gpurun_out/${T}_pytest_gpu.txt; tail -n 14 gpurun_out/${T}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash scripts/gpu_profile.sh $T
bash scripts/gpu_timeline.sh > /dev/null; cp gpurun_out/timeline/timeline.txt gpurun_out/${T}_timeline_d8_16384.txt
for m in "d8 16384" "d8 4096" "dinf 16384" "dinf 4096"; do set -- $m; taudem_amd/bin/tdxbench $1 -n $2 -steps 3 -crc 2>/dev/null | tail -n 1; done > gpurun_out/${T}_tdxbench_crc.jsonl
taudem_amd/bin/tdxbench decay -steps 1 -crc 2>/dev/null | tail -n 1 >> gpurun_out/${T}_tdxbench_crc.jsonl
timeout 600 python scripts/bench_flowalg.py --size 16384 2>/dev/null | tail -n 1 > gpurun_out/${T}_flowalg_16384.json
cut -c1-400 gpurun_out/${T}_tdxbench_crc.jsonl; cut -c1-700 gpurun_out/${T}_flowalg_16384.json
