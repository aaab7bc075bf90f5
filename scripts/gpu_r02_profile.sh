#!/bin/bash
# Round-2 evidence: default bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE / SQ passes
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-r02m}
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.log 2>&1; tail -1 gpurun_out/${TAG}_bench_default.log > gpurun_out/${TAG}_bench_default.json; cut -c1-900 gpurun_out/${TAG}_bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r -- python $R/bench.py --cpu-sample 0 > $R/gpurun_out/prof_$TAG.log 2>&1)
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats_16384_default_bench.csv
rm -rf gpurun_out/prof_$TAG
cd /tmp
for pass in fetch write sq; do
  case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU";; esac
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$pass -o p -- python $R/bench.py --cpu-sample 0 --steps 2 --warmup 1 > $R/gpurun_out/pmc_$pass.log 2>&1
  python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$pass $R/gpurun_out/${TAG}_pmc_${pass}_summary.json | head -4
  rm -rf $R/gpurun_out/pmc_$pass
done
cd $R
head -16 gpurun_out/${TAG}_kernel_stats_16384_default_bench.csv | cut -c1-140
