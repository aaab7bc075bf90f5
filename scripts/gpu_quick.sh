#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_dir scripts/micro/dpp_dir.hip 2>/dev/null && timeout 30 /tmp/dpp_dir
timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
timeout 200 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/b.json; python -c "import sys,json; d=json.loads(open('gpurun_out/b.json').read()); print(d['value'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'], d['kernel_class_launches_per_step'])"
if [ -n "$QUICK_AB" ]; then
TDX_RELAX_LDS=1 timeout 200 python bench.py --size 16384 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > gpurun_out/b_lds.json; python -c "import sys,json; d=json.loads(open('gpurun_out/b_lds.json').read()); print('LDS variant:', d['value'], d['stage_ms_per_step'])"
fi
