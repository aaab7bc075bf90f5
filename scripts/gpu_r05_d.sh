#!/bin/bash
# round 5: tree-parallel big-cell fold + int32 level fix: the D8 / strips / pathological tests, the 16384^2 properties, then the eight-strip trace and the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_d8.py tests/test_gpu_pathological.py tests/test_gpu_multigpu.py tests/test_gpu_fullsize.py -m gpu -q -x --deselect tests/test_gpu_d8.py::test_aread8_counts_above_2_30 -k "not dinf and not decay" > gpurun_out/r05d_pytest.txt 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r05d_pytest.txt
export TDX_COMM_TRACE=1
timeout 600 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/r05d_seg2_d8.json > gpurun_out/r05d_8strips_d8.json 2> gpurun_out/r05d_8strips_d8.err
echo "d8 seg2 rc=$?"
python scripts/project_8gpu.py gpurun_out/r05d_seg2_d8.json > gpurun_out/r05d_projection_d8.txt; cat gpurun_out/r05d_projection_d8.txt
cut -c1-900 gpurun_out/r05d_8strips_d8.json
unset TDX_COMM_TRACE
timeout 600 python bench.py --no-extras --cpu-sample 0 2>/dev/null | tail -n 1 > gpurun_out/r05d_bench_noextras.json; cut -c1-1500 gpurun_out/r05d_bench_noextras.json
