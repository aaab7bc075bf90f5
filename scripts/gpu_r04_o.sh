#!/bin/bash
# Round 4, call O: configs[3] strip parity on every cell (D8FlowDir vs the restatement's breadth-first form, AreaD8 through its loop body)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04o
mkdir -p $O
cd $R
ORC_TIMING=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q --no-header -p no:cacheprovider --timeout=1200 --timeout-method=thread --durations=4 -k "d8_config4_strip" > $O/pytest_d8_strip.txt 2>&1; tail -n 12 $O/pytest_d8_strip.txt
