#!/bin/bash
# the default bench line, twice, next to the short form (is the 29.9 ms of the closing run the invocation or the box?)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r03x}
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, {k: round(v,2) for k,v in d['kernel_class_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'],4), 'stencil', round(d['roofline_streaming_stencil']['frac'],3))" $1 $2; }
timeout 300 python bench.py 2>/dev/null | tail -n 1 > gpurun_out/${T}_default1.json; show gpurun_out/${T}_default1.json default1
timeout 200 python bench.py --no-extras --cpu-sample 0 2>/dev/null | tail -n 1 > gpurun_out/${T}_short.json; show gpurun_out/${T}_short.json short
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/${T}_default2.json; show gpurun_out/${T}_default2.json steps20
