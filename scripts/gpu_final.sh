#!/bin/bash
# Closing run of a round: the full GPU suite, smoke, then the evidence set in the order of its importance (a run that is cut short keeps what came first): the default
# bench line, kernel stats of the same command, the eight-strip runs of BASELINE.json configs[3] / [4] with segment traces and projections, one-step timeline, CRCs of the
# native harness, config 3's kernel stats, SQ / FETCH / WRITE counter passes (each --pmc pass on its own, no trace domains).
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
T=${1:-r06z}
[ "$2" = "nopytest" ] || timeout 1700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=1200 --timeout-method=thread --durations=8 2>&1 | tail -n 20 > gpurun_out/${T}_pytest_gpu.txt; tail -n 14 gpurun_out/${T}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 600 python bench.py > gpurun_out/${T}_bench_default.log 2>&1; tail -n 1 gpurun_out/${T}_bench_default.log > gpurun_out/${T}_bench_default.json; cut -c1-700 gpurun_out/${T}_bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -o r -- python $R/bench.py --cpu-sample 0 --no-extras > $R/gpurun_out/prof_$T.log 2>&1)
find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats_16384_default_bench.csv
rm -rf gpurun_out/prof_$T
head -n 12 gpurun_out/${T}_kernel_stats_16384_default_bench.csv | cut -c1-150
for W in d8 decay; do
  TDX_COMM_TRACE=1 timeout 600 python bench.py --gpus 8 --in-process --workload $W --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/${T}_segments_8strips_$W.json > gpurun_out/${T}_8strips_65536_$W.log 2> gpurun_out/${T}_8strips_65536_${W}_comm_trace.txt
  tail -n 1 gpurun_out/${T}_8strips_65536_$W.log > gpurun_out/${T}_8strips_65536_$W.json; rm -f gpurun_out/${T}_8strips_65536_$W.log
  python scripts/project_8gpu.py gpurun_out/${T}_segments_8strips_$W.json > gpurun_out/${T}_projection_8gpu_$W.txt; cat gpurun_out/${T}_projection_8gpu_$W.txt | cut -c1-160
  sort gpurun_out/${T}_8strips_65536_${W}_comm_trace.txt | uniq | grep taudem_amd > gpurun_out/${T}_tmp.txt; mv gpurun_out/${T}_tmp.txt gpurun_out/${T}_8strips_65536_${W}_comm_trace.txt
done
bash scripts/gpu_timeline.sh > /dev/null; cp gpurun_out/timeline/timeline.txt gpurun_out/${T}_timeline_d8_16384.txt
for m in "d8 16384" "d8 4096" "dinf 4096"; do set -- $m; taudem_amd/bin/tdxbench $1 -n $2 -steps 3 -crc 2>/dev/null | tail -n 1; done > gpurun_out/${T}_tdxbench_crc.jsonl
taudem_amd/bin/tdxbench decay -steps 1 -crc 2>/dev/null | tail -n 1 >> gpurun_out/${T}_tdxbench_crc.jsonl
cut -c1-300 gpurun_out/${T}_tdxbench_crc.jsonl
TDX_AD8_DEBUG=1 taudem_amd/bin/tdxbench d8 -n 16384 -steps 1 -warmup 0 2>&1 | grep "cycles per tile" > gpurun_out/${T}_ad8_tile_phases.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_$T -o r -- $R/taudem_amd/bin/tdxbench dinf -n 32768 -steps 2 -warmup 1 > $R/gpurun_out/${T}_tdxbench_dinf_32768.json 2> $R/gpurun_out/prof3_$T.log)
find gpurun_out/prof3_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats_dinf_32768.csv
rm -rf gpurun_out/prof3_$T
cd /tmp
for pass in sq fetch write; do
  case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; sq) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU";; esac
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$pass -o p -- python $R/bench.py --cpu-sample 0 --no-extras --steps 2 --warmup 1 > $R/gpurun_out/pmc_$pass.log 2>&1
  python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_$pass $R/gpurun_out/${T}_pmc_${pass}_summary.json | head -n 3
  rm -rf $R/gpurun_out/pmc_$pass
done
