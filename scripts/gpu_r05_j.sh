#!/bin/bash
# round 5: bounded rounds between exchanges in the generic sweeps and in the outlets' closure: the whole GPU suite, then the eight-strip decay trace
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -n 30 > gpurun_out/r05j_pytest_gpu.txt; grep -E "passed|failed" gpurun_out/r05j_pytest_gpu.txt; grep -B25 "short test summary" gpurun_out/r05j_pytest_gpu.txt | head -40
export TDX_COMM_TRACE=1
timeout 900 python bench.py --gpus 8 --in-process --workload decay --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/r05j_seg2_decay.json > gpurun_out/r05j_8strips_decay.json 2> gpurun_out/r05j_8strips_decay.err
echo "decay rc=$?"; python scripts/project_8gpu.py gpurun_out/r05j_seg2_decay.json | tee gpurun_out/r05j_projection_decay.txt | tail -4
cut -c1-700 gpurun_out/r05j_8strips_decay.json
