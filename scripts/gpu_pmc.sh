#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp
for pass in sq fetch write; do
  case $pass in
    sq) C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES";;
    fetch) C="FETCH_SIZE";;
    write) C="WRITE_SIZE";;
  esac
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$pass -o p -- python $R/bench.py --size 16384 --steps 1 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmc_$pass.log 2>&1
  python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$pass $R/gpurun_out/pmc_${pass}_summary.json | head -60
  find $R/gpurun_out/pmc_$pass -name "*.csv" -size +2M -delete; find $R/gpurun_out/pmc_$pass -name "*.db" -delete
done
