mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout=1200 --timeout-method=thread --durations=8 2>&1 | tail -n 20 > gpurun_out/r06y_pytest_gpu.txt; tail -n 14 gpurun_out/r06y_pytest_gpu.txt
timeout 600 python scripts/bench_flowalg.py 2>&1 | tail -1 > gpurun_out/r06y_flowalg.json; cat gpurun_out/r06y_flowalg.json
