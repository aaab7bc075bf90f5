#!/bin/bash
# Evidence set of a round (TAG = the prefix of the files in gpurun_out/, copied into profiles/ by hand): the default bench line (with the config3 / config5_strip legs), rocprofv3 kernel stats of the same pipeline command and of
# config 3 (native harness), FETCH_SIZE / WRITE_SIZE / SQ counter passes (each --pmc pass on its own, no trace domains).
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-r05p}
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench_default.log > gpurun_out/${TAG}_bench_default.json; cut -c1-600 gpurun_out/${TAG}_bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r -- python $R/bench.py --cpu-sample 0 --no-extras > $R/gpurun_out/prof_$TAG.log 2>&1)
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats_16384_default_bench.csv
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_$TAG -o r -- $R/taudem_amd/bin/tdxbench dinf -n 32768 -steps 2 -warmup 1 > $R/gpurun_out/${TAG}_tdxbench_dinf_32768.json 2> $R/gpurun_out/prof3_$TAG.log)
find gpurun_out/prof3_$TAG -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats_dinf_32768.csv
rm -rf gpurun_out/prof3_$TAG
cd /tmp
for pass in fetch write sq; do
  case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU";; esac
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$pass -o p -- python $R/bench.py --cpu-sample 0 --no-extras --steps 2 --warmup 1 > $R/gpurun_out/pmc_$pass.log 2>&1
  python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$pass $R/gpurun_out/${TAG}_pmc_${pass}_summary.json | head -n 4
  rm -rf $R/gpurun_out/pmc_$pass
done
cd $R
head -n 14 gpurun_out/${TAG}_kernel_stats_16384_default_bench.csv | cut -c1-150
head -n 10 gpurun_out/${TAG}_kernel_stats_dinf_32768.csv | cut -c1-150

# BASELINE.json configs[3] / [4] in eight strips on this one GPU: functional step + segment trace (one rank on the device at a time) -> projected 8-GPU critical path
for W in d8 decay; do
  TDX_COMM_TRACE=1 timeout 900 python bench.py --gpus 8 --in-process --workload $W --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/${TAG}_segments_8strips_$W.json > gpurun_out/${TAG}_8strips_65536_$W.log 2> gpurun_out/${TAG}_8strips_65536_${W}_comm_trace.txt
  tail -n 1 gpurun_out/${TAG}_8strips_65536_$W.log > gpurun_out/${TAG}_8strips_65536_$W.json
  python scripts/project_8gpu.py gpurun_out/${TAG}_segments_8strips_$W.json > gpurun_out/${TAG}_projection_8gpu_$W.txt; cat gpurun_out/${TAG}_projection_8gpu_$W.txt
  sort gpurun_out/${TAG}_8strips_65536_${W}_comm_trace.txt | uniq | grep taudem_amd > gpurun_out/${TAG}_tmp.txt; mv gpurun_out/${TAG}_tmp.txt gpurun_out/${TAG}_8strips_65536_${W}_comm_trace.txt
done
