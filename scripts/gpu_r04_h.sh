#!/bin/bash
# Round 4, call H: kernel timeline of AreaDinf at 16384^2 (where the 94 ms are: bulk rounds on 32 x 32 tiles vs tail on 64 x 64 tiles)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04h
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- $R/taudem_amd/bin/tdxbench dinf -n 16384 -steps 1 -warmup 0 > $O/trace.log 2>&1
python $R/scripts/timeline.py $O/tr dsweep > $O/timeline_dinf_16384.txt; head -n 14 $O/timeline_dinf_16384.txt; tail -n 1 $O/timeline_dinf_16384.txt | cut -c1-6000
rm -rf $O/tr
