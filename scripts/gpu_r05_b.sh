#!/bin/bash
# round 5: the full GPU suite on the current tree, then the eight-strip segment traces (both workloads)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05b_pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r05b_pytest_gpu.txt
export TDX_COMM_TRACE=1
timeout 600 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/r05b_seg2_d8.json > gpurun_out/r05b_8strips_d8_seg2.json 2> gpurun_out/r05b_8strips_d8_seg2.err
echo "d8 seg2 rc=$?"
python scripts/project_8gpu.py gpurun_out/r05b_seg2_d8.json > gpurun_out/r05b_projection_d8.txt; cat gpurun_out/r05b_projection_d8.txt
timeout 600 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 > gpurun_out/r05b_8strips_d8.json 2> gpurun_out/r05b_8strips_d8.err
echo "d8 rc=$?"; cat gpurun_out/r05b_8strips_d8.json | cut -c1-1500
timeout 900 python bench.py --gpus 8 --in-process --workload decay --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/r05b_seg2_decay.json > gpurun_out/r05b_8strips_decay_seg2.json 2> gpurun_out/r05b_8strips_decay_seg2.err
echo "decay seg2 rc=$?"
python scripts/project_8gpu.py gpurun_out/r05b_seg2_decay.json > gpurun_out/r05b_projection_decay.txt; cat gpurun_out/r05b_projection_decay.txt
