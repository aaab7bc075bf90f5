#!/bin/bash
mkdir -p gpurun_out
timeout 900 python scripts/gpu_exp_ad8.py > gpurun_out/exp_ad8.log 2>&1
timeout 600 python bench.py --size 16384 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/bench2_16384.log 2>&1
cat gpurun_out/exp_ad8.log; tail -1 gpurun_out/bench2_16384.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'])"
