#!/bin/bash
# Round-3 closing run: the full GPU suite, then the evidence set (bench line, kernel stats, PMC passes)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r03z}
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=600 --timeout-method=thread --durations=6 2>&1 | tail -n 16 > gpurun_out/${T}_pytest_gpu.txt; tail -n 12 gpurun_out/${T}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash scripts/gpu_r03_profile.sh $T
