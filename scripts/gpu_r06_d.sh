export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06d}
timeout 1200 python -m pytest tests/test_gpu_d8.py tests/test_gpu_dinf.py tests/test_gpu_pathological.py tests/test_gpu_fuzz_strips.py tests/test_gpu_multigpu.py tests/test_gpu_large_golden.py -m gpu -q --no-header -p no:cacheprovider --timeout=600 -x 2>&1 | tail -n 12 > gpurun_out/${T}_pytest_subset.txt; tail -n 5 gpurun_out/${T}_pytest_subset.txt
taudem_amd/bin/tdxbench d8 -n 16384 -steps 8 -crc 2>/dev/null | tail -1 > gpurun_out/${T}_tdxbench_d8.json; cut -c1-200 gpurun_out/${T}_tdxbench_d8.json; grep -o '"crc[^}]*}' gpurun_out/${T}_tdxbench_d8.json
taudem_amd/bin/tdxbench dinf -n 16384 -steps 3 -crc 2>/dev/null | tail -1 > gpurun_out/${T}_tdxbench_dinf.json; cut -c1-200 gpurun_out/${T}_tdxbench_dinf.json; grep -o '"crc[^}]*}' gpurun_out/${T}_tdxbench_dinf.json
bash scripts/gpu_timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/${T}_timeline_d8_16384.txt; grep -c "fillBuffer\|copyBuffer" gpurun_out/${T}_timeline_d8_16384.txt
