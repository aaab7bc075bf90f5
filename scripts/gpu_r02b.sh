#!/bin/bash
# D-infinity tile dependency sweep: parity (goldens, strips, config-scale digests) and timing at 16384^2 / 32768^2
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_gpu_cli.py tests/test_gpu_multigpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -15
timeout 900 python -m pytest tests/test_strips.py -m gpu -q --no-header -p no:cacheprovider -x -k dinf 2>&1 | tail -8
timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/r02b_bench_dinf_16384.json; cut -c1-700 gpurun_out/r02b_bench_dinf_16384.json
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 > gpurun_out/r02b_bench_dinf_32768.json; cut -c1-700 gpurun_out/r02b_bench_dinf_32768.json
TDX_DINF_WALK=1 timeout 600 python scripts/bench_dinf.py --size 16384 --steps 1 --warmup 1 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py --cpu-sample 0 2>&1 | tail -1 | cut -c1-600
