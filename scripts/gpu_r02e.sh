#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_gpu_cli.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8
timeout 900 python -m pytest tests/test_strips.py -m gpu -q --no-header -p no:cacheprovider -x -k dinf 2>&1 | tail -4
for sw in 24 8 48; do
TDX_DINF_BULK_SWEEPS=$sw timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sweeps $sw', d['areadinf_ms'], d['areadinf_classes'], d['areadinf_rounds'])"
done
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 > gpurun_out/r02e_bench_dinf_32768.json; cut -c1-700 gpurun_out/r02e_bench_dinf_32768.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dinf -o r -- python $GRAFT_REPO_ROOT/scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 > /dev/null 2>&1)
find gpurun_out/prof_dinf -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02e_kernel_stats_dinf_16384.csv
find gpurun_out/prof_dinf -name "*.csv" -size +1M -delete; find gpurun_out/prof_dinf -name "*.db" -delete
head -12 gpurun_out/r02e_kernel_stats_dinf_16384.csv | cut -c1-160
