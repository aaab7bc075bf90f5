#!/bin/bash
# Round-3 step C: new slope kernel A/B, relaxation phase clocks, bench line with the config legs, 8 strips of 65536 x 8192 on one GPU, full suite
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03c}
timeout 90 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8.json 2> gpurun_out/${T}_d8.err
TDX_DEBUG_ROUNDS=1 timeout 90 $B d8 -n 16384 -steps 1 -warmup 0 > /dev/null 2> gpurun_out/${T}_d8_rounds.txt
cut -c1-400 gpurun_out/${T}_d8.json; grep -c . gpurun_out/${T}_d8_rounds.txt
timeout 600 python bench.py --steps 5 --warmup 2 2> gpurun_out/${T}_bench.err | tail -n 1 > gpurun_out/${T}_bench_default.json; cut -c1-300 gpurun_out/${T}_bench_default.json; tail -n 3 gpurun_out/${T}_bench.err
TDX_COMM_TRACE=1 timeout 900 python bench.py --gpus 8 --in-process --steps 1 --warmup 0 2> gpurun_out/${T}_8strips_d8.err | tail -n 1 > gpurun_out/${T}_8strips_d8.json; cut -c1-1200 gpurun_out/${T}_8strips_d8.json; tail -n 30 gpurun_out/${T}_8strips_d8.err
TDX_COMM_TRACE=1 timeout 900 python bench.py --gpus 8 --in-process --workload decay --steps 1 --warmup 0 2> gpurun_out/${T}_8strips_decay.err | tail -n 1 > gpurun_out/${T}_8strips_decay.json; cut -c1-1200 gpurun_out/${T}_8strips_decay.json; tail -n 12 gpurun_out/${T}_8strips_decay.err
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --timeout=600 --timeout-method=thread --durations=8 2>&1 | tail -n 30 > gpurun_out/${T}_pytest_gpu.txt; tail -n 22 gpurun_out/${T}_pytest_gpu.txt
