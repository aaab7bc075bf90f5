#!/bin/bash
# Round 4, call E: the D-infinity full-size tests with every cell of ang pinned by the restatement's breadth-first form
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
nproc
ORC_TIMING=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q --no-header -p no:cacheprovider --timeout=1200 --timeout-method=thread --durations=6 -k "dinf_properties or dinf_config3_at" -s > $O/pytest_dinf.txt 2>&1
grep "orc\]" $O/pytest_dinf.txt | tail -n 40; tail -n 14 $O/pytest_dinf.txt
