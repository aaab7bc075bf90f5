#!/bin/bash
# Round-4 closing run: the full GPU suite, smoke, then the evidence set (bench line, kernel stats, PMC passes)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04z}
timeout 1700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=1200 --timeout-method=thread --durations=8 2>&1 | tail -n 20 > gpurun_out/${T}_pytest_gpu.txt; tail -n 14 gpurun_out/${T}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash scripts/gpu_r04_profile.sh $T
