export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06e}
timeout 1200 python -m pytest tests/test_gpu_d8.py tests/test_gpu_dinf.py tests/test_gpu_pathological.py tests/test_gpu_fuzz_strips.py tests/test_gpu_multigpu.py tests/test_gpu_large_golden.py tests/test_flowalg.py tests/test_gpu_gridnet.py -m gpu -q --no-header -p no:cacheprovider --timeout=600 -x 2>&1 | tail -n 12 > gpurun_out/${T}_pytest_subset.txt; tail -n 4 gpurun_out/${T}_pytest_subset.txt
taudem_amd/bin/tdxbench d8 -n 16384 -steps 8 -crc 2>/dev/null | tail -1 > gpurun_out/${T}_tdxbench_d8.json; cut -c1-200 gpurun_out/${T}_tdxbench_d8.json; grep -o '"crc[^}]*}' gpurun_out/${T}_tdxbench_d8.json
(TDX_DEBUG_ROUNDS=1 taudem_amd/bin/tdxbench dinf -n 16384 -steps 1 -warmup 0 2>&1 | grep -v "^{" | cut -c1-300) > gpurun_out/${T}_dinf_staged_vs_evaluated_16384.txt
(TDX_DEBUG_ROUNDS=1 taudem_amd/bin/tdxbench dinf -n 32768 -steps 1 -warmup 0 2>&1 | grep -v "^{" | cut -c1-300) > gpurun_out/${T}_dinf_staged_vs_evaluated_32768.txt
timeout 600 python bench.py 2>gpurun_out/${T}_bench_stderr.txt | tail -n 1 > gpurun_out/${T}_bench_default.json; cut -c1-300 gpurun_out/${T}_bench_default.json
bash scripts/gpu_timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/${T}_timeline_d8_16384.txt; grep -c "fillBuffer\|copyBuffer" gpurun_out/${T}_timeline_d8_16384.txt
