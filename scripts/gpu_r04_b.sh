#!/bin/bash
# Round 4, call B: diagnosis of the configs[4] strip check
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 1500 python scripts/diag/decay_closure.py > $O/decay_closure.txt 2>&1
cat $O/decay_closure.txt | cut -c1-900
