mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_flowalg.py tests/test_gpu_pathological.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06q_tests.txt
cat gpurun_out/r06q_tests.txt
TDX_SWEEP_VERIFY=2 timeout 300 python scripts/bench_flowalg.py --only dinfrevaccum,dinfupdependence --digest 2>&1 | tail -4 > gpurun_out/r06q_reverse.txt
timeout 300 python scripts/bench_flowalg.py --only dinfrevaccum,dinfupdependence --digest 2>&1 | tail -1 >> gpurun_out/r06q_reverse.txt
cat gpurun_out/r06q_reverse.txt
