#!/bin/bash
# Evidence run: parity tests, bench (default command), rocprofv3 kernel stats and the three PMC passes of the same command.
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/pytest_gpu.txt; cat gpurun_out/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-1500
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -o r1 -- python $R/bench.py --cpu-sample 0 > $R/gpurun_out/prof_d.log 2>&1)
find gpurun_out/prof_d -name "*kernel_trace.csv" -delete; find gpurun_out/prof_d -name "*.db" -delete
cd /tmp
for pass in fetch write sq; do
  case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU";; esac
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$pass -o p -- python $R/bench.py --cpu-sample 0 > $R/gpurun_out/pmc_$pass.log 2>&1
  python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$pass $R/gpurun_out/pmc_${pass}_summary.json | head -8
  find $R/gpurun_out/pmc_$pass -name "*.csv" -size +2M -delete; find $R/gpurun_out/pmc_$pass -name "*.db" -delete
done
cd $R; timeout 300 python scripts/bench_dinf.py --size 16384 2>&1 | tail -1 > gpurun_out/bench_dinf_16384.json; cut -c1-300 gpurun_out/bench_dinf_16384.json
