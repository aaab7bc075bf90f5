#!/bin/bash
# round 5: new tests (pathological inputs, int32 level fields, many big cells in strips), then the suite's strip tests, then the eight-strip traces
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_pathological.py tests/test_gpu_d8.py tests/test_gpu_multigpu.py tests/test_gpu_dinf.py -m gpu -q -x --deselect tests/test_gpu_d8.py::test_aread8_counts_above_2_30 > gpurun_out/r05c_pytest.txt 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r05c_pytest.txt
export TDX_COMM_TRACE=1
timeout 600 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/r05c_seg2_d8.json > gpurun_out/r05c_8strips_d8_seg2.json 2> gpurun_out/r05c_8strips_d8_seg2.err
echo "d8 seg2 rc=$?"
python scripts/project_8gpu.py gpurun_out/r05c_seg2_d8.json > gpurun_out/r05c_projection_d8.txt; cat gpurun_out/r05c_projection_d8.txt
timeout 600 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 > gpurun_out/r05c_8strips_d8.json 2> gpurun_out/r05c_8strips_d8.err
echo "d8 rc=$?"; cut -c1-1200 gpurun_out/r05c_8strips_d8.json
