#!/bin/bash
# Round 4, call F: after the advisor fixes - multi-GPU tests, the eight-strips-on-one-GPU functional lines of both workloads on the current engine
# (exchange / all-reduce counts, largest level per flat iteration), reference CRCs of the 16384^2 pipeline for later A/B runs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_multigpu.py tests/test_strips.py tests/test_gpu_d8.py -m gpu -q --no-header -p no:cacheprovider --timeout=600 --timeout-method=thread -x 2>&1 | tail -n 6
TDX_COMM_TRACE=1 timeout 900 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 > $O/8strips_65536_d8.json 2> $O/8strips_65536_d8_comm_trace.txt; tail -n 1 $O/8strips_65536_d8.json | cut -c1-1500
TDX_COMM_TRACE=1 timeout 900 python bench.py --gpus 8 --in-process --workload decay --steps 1 --warmup 0 > $O/8strips_65536_decay.json 2> $O/8strips_65536_decay_comm_trace.txt; tail -n 1 $O/8strips_65536_decay.json | cut -c1-1200
timeout 200 taudem_amd/bin/tdxbench d8 -n 16384 -steps 3 -crc > $O/tdxbench_d8.json 2>&1; cut -c1-260 $O/tdxbench_d8.json; grep -o '"crc".*' $O/tdxbench_d8.json
timeout 200 taudem_amd/bin/tdxbench dinf -n 16384 -steps 2 -crc > $O/tdxbench_dinf.json 2>&1; cut -c1-260 $O/tdxbench_dinf.json; grep -o '"crc".*' $O/tdxbench_dinf.json
