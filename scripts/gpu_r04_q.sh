#!/bin/bash
# Round 4, call Q: separable eight-neighbour minimum of the uniform tile operators - CRCs and times, then the D8 / golden tests
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q
bash scripts/gpu_r04_k.sh d8 16384 "X=1" 2>&1 | tee gpurun_out/r04q/d8_16384.txt
bash scripts/gpu_r04_k.sh d8 4096 "X=1" 2>&1 | tee gpurun_out/r04q/d8_4096.txt
bash scripts/gpu_r04_k.sh dinf 4096 "X=1" 2>&1 | tee gpurun_out/r04q/dinf_4096.txt
timeout 900 python -m pytest tests/test_gpu_d8.py tests/test_gpu_large_golden.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -n 5 | tee gpurun_out/r04q/pytest.txt
