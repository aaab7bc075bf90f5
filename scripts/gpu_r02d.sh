#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_gpu_cli.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8
timeout 900 python -m pytest tests/test_strips.py -m gpu -q --no-header -p no:cacheprovider -x -k dinf 2>&1 | tail -4
timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/r02d_bench_dinf_16384.json; cut -c1-700 gpurun_out/r02d_bench_dinf_16384.json
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 > gpurun_out/r02d_bench_dinf_32768.json; cut -c1-700 gpurun_out/r02d_bench_dinf_32768.json
bash scripts/gpu_r02c.sh 2>&1 | head -30
sed -n 495,505p gpurun_out/r02c_dinf_round_times.txt
