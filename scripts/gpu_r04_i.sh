#!/bin/bash
# Round 4, call I: phase clocks of the D-infinity sweeps with the cheap instrumentation (LDS accumulation, one flush per workgroup and launch)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
TDX_DEBUG_ROUNDS=1 timeout 300 taudem_amd/bin/tdxbench dinf -n 16384 -steps 1 -warmup 0 > $O/dinf_phases.json 2> $O/dinf_phases.txt; grep "\[rounds" $O/dinf_phases.txt | cut -c1-200 | head -n 40; cut -c1-300 $O/dinf_phases.json
