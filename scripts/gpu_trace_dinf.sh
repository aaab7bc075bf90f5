#!/bin/bash
# rocprofv3 kernel stats of the D-infinity configuration (BASELINE config 3: 32768^2) -> gpurun_out/<tag>_kernel_stats_dinf_32768.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02z}
mkdir -p $R/gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dinf_$TAG -o r -- python $R/scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 > $R/gpurun_out/prof_dinf_$TAG.log 2>&1)
find $R/gpurun_out/prof_dinf_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats_dinf_32768.csv
rm -rf $R/gpurun_out/prof_dinf_$TAG
head -14 $R/gpurun_out/${TAG}_kernel_stats_dinf_32768.csv | cut -c1-170
