#!/bin/bash
# D8 sweep engine on two tile geometries: parity of every user + timings
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gridnet.py tests/test_flowalg.py tests/test_gpu_d8.py tests/test_gpu_large_golden.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error|Error|assert|differ" | tail -6
timeout 900 python -m pytest tests/test_strips.py tests/test_gpu_multigpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
for u in 6000 0 2000 20000; do
echo "until $u"; TDX_D8_BULK_UNTIL=$u timeout 600 python scripts/bench_gridnet.py 2>&1 | tail -3
done
