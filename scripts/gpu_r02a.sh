#!/bin/bash
# Round 2, first GPU pass: full parity suite (incl. config-scale digests, multi-GPU plumbing), default bench, config 3 at 32768^2.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --durations=15 2>&1 | tail -40 > gpurun_out/r02a_pytest_gpu.txt; tail -25 gpurun_out/r02a_pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r02a_bench_default.log 2>&1; tail -1 gpurun_out/r02a_bench_default.log | cut -c1-2500
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 > gpurun_out/r02a_bench_dinf_32768.json; cut -c1-600 gpurun_out/r02a_bench_dinf_32768.json
timeout 120 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/rccl_two_ranks_one_gpu.py 2>&1 | grep "^rank" | head
nproc; free -g | head -2
