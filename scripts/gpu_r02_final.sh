#!/bin/bash
# Round-2 closing evidence: the full GPU suite, the default bench line + rocprofv3 kernel stats + PMC passes of the same command
# (scripts/gpu_r02_profile.sh), the D-infinity / GridNet / flow-algebra benches.  Everything lands in gpurun_out/<tag>_*.
export TMPDIR=/tmp
TAG=${1:-r02z}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt; tail -3 gpurun_out/${TAG}_pytest_gpu.txt
bash scripts/gpu_r02_profile.sh $TAG
timeout 600 python scripts/bench_dinf.py --size 16384 --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/${TAG}_bench_dinf_16384.json
timeout 600 python scripts/bench_dinf.py --size 32768 --steps 1 --warmup 1 2>&1 | tail -1 > gpurun_out/${TAG}_bench_dinf_32768.json
timeout 600 python scripts/bench_gridnet.py 2>&1 | tail -1 > gpurun_out/${TAG}_bench_gridnet_16384.json
timeout 800 python scripts/bench_flowalg.py 2>&1 | tail -1 > gpurun_out/${TAG}_bench_flowalg_16384.json
cut -c1-400 gpurun_out/${TAG}_bench_dinf_16384.json gpurun_out/${TAG}_bench_dinf_32768.json gpurun_out/${TAG}_bench_gridnet_16384.json
