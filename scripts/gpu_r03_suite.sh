#!/bin/bash
# the whole GPU suite + smoke + the default bench line (closing check after late changes)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r03zzz}
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=600 --timeout-method=thread --durations=4 2>&1 | tail -n 12 > gpurun_out/${T}_pytest_gpu.txt; grep -E "passed|failed|error" gpurun_out/${T}_pytest_gpu.txt | tail -n 3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 300 python bench.py 2>/dev/null | tail -n 1 > gpurun_out/${T}_bench_default.json; cut -c1-330 gpurun_out/${T}_bench_default.json
