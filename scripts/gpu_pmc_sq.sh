#!/bin/bash
# SQ issue / wait breakdown of the tile-relaxation kernels (one 16384^2 step), plus the list of available counters.
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc_avail.txt 2>&1
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python $R/bench.py --size 16384 --steps 1 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmc_sq.log 2>&1
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_sq_summary.json | head -40
find $R/gpurun_out/pmc_sq -name "*.csv" -size +2M -delete; find $R/gpurun_out/pmc_sq -name "*.db" -delete
