#!/bin/bash
# Round 4, call N: rounds report their sizes straight to pinned host memory (no copy kernel between batches) - canary, parity, A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04n
mkdir -p $O
cd $R
timeout 100 taudem_amd/bin/tdxbench d8 -n 4096 -steps 1 -crc 2>&1 | grep -o '"crc".*'
timeout 100 taudem_amd/bin/tdxbench dinf -n 4096 -steps 1 -crc 2>&1 | grep -o '"crc".*'
timeout 900 python -m pytest tests/test_gpu_d8.py tests/test_gpu_dinf.py tests/test_gpu_gridnet.py tests/test_strips.py -m gpu -q --no-header -p no:cacheprovider --timeout=600 --timeout-method=thread -x 2>&1 | tail -n 3
bash scripts/gpu_r04_k.sh d8 16384 "A=1" "TDX_RELAX_COUNT_COPY=1" "A=2" "TDX_RELAX_COUNT_COPY=1" 2>&1 | cut -c1-330
bash scripts/gpu_r04_k.sh dinf 16384 "A=1" "TDX_RELAX_COUNT_COPY=1" 2>&1 | cut -c1-330
