#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --deselect tests/test_gpu_cli.py::test_bigtiff_above_4gb_round_trip 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -12
timeout 600 python scripts/bench_gridnet.py 2>&1 | tail -1 | cut -c1-500
TDX_GN_WALK=1 timeout 600 python scripts/bench_gridnet.py 2>&1 | tail -1 | cut -c1-300
