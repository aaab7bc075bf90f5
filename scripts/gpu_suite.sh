#!/bin/bash
# the whole GPU suite + smoke (+ optionally the default bench line): bash scripts/gpu_suite.sh TAG [bench]
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05}
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=900 --timeout-method=thread --durations=8 -x 2>&1 | tail -n 40 > gpurun_out/${T}_pytest_gpu.txt; tail -n 25 gpurun_out/${T}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
if [ "$2" = "bench" ]; then timeout 400 python bench.py 2>/dev/null | tail -n 1 > gpurun_out/${T}_bench_default.json; cut -c1-400 gpurun_out/${T}_bench_default.json; fi
