#!/bin/bash
# Round 4, call J: timeline + phase clocks of the D-infinity sweeps (after a change)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04j
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- $R/taudem_amd/bin/tdxbench dinf -n 16384 -steps 1 -warmup 0 > $O/trace.log 2>&1
python $R/scripts/timeline.py $O/tr dsweep > $O/timeline_dinf_16384.txt; head -n 3 $O/timeline_dinf_16384.txt; tail -n 1 $O/timeline_dinf_16384.txt | cut -c1-400
rm -rf $O/tr
cd $R
TDX_DEBUG_ROUNDS=1 timeout 300 taudem_amd/bin/tdxbench dinf -n 16384 -steps 1 -warmup 0 > $O/dinf_phases.json 2> $O/dinf_phases.txt; grep "\[rounds" $O/dinf_phases.txt | cut -c1-200 | head -n 8
