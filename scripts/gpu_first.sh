#!/bin/bash
# first GPU run: parity tests, smoke, small + full bench
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt
timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
(oracle/_ref/pitremove 2>&1 | head -3; ldd oracle/_ref/pitremove | grep -i "not found") > gpurun_out/ref_check.log 2>&1
timeout 600 python bench.py --size 2048 --steps 2 --warmup 1 > gpurun_out/bench_2048.log 2>&1
timeout 900 python bench.py --size 8192 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/bench_8192.log 2>&1
timeout 1200 python bench.py --size 16384 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/bench_16384.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; tail -2 gpurun_out/bench_2048.log; tail -2 gpurun_out/bench_8192.log; tail -2 gpurun_out/bench_16384.log
