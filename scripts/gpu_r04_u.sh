#!/bin/bash
# Round 4, call U: the eight-strips-on-one-GPU functional lines of both workloads on the closing engine (TAG_8strips_65536_*.json + the per-rank comm traces)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04zz}
cd $R
mkdir -p gpurun_out
TDX_COMM_TRACE=1 timeout 900 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 > gpurun_out/${T}_8strips_65536_d8.json 2> gpurun_out/${T}_8strips_65536_d8_comm_trace.txt; tail -n 1 gpurun_out/${T}_8strips_65536_d8.json | cut -c1-1800
TDX_COMM_TRACE=1 timeout 900 python bench.py --gpus 8 --in-process --workload decay --steps 1 --warmup 0 > gpurun_out/${T}_8strips_65536_decay.json 2> gpurun_out/${T}_8strips_65536_decay_comm_trace.txt; tail -n 1 gpurun_out/${T}_8strips_65536_decay.json | cut -c1-1500
