#!/bin/bash
# why is bench.py's d8flowdir slower than the native harness'?  host-side batch hand-over under Python: batch length, pairing
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r03w}
one() { n=$1; shift
  env "$@" timeout 200 python bench.py --no-extras --cpu-sample 0 --steps 5 --warmup 1 2>/dev/null | tail -n 1 > gpurun_out/${T}_$n.json
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, {k: round(v,2) for k,v in d['kernel_class_ms_per_step'].items()})" gpurun_out/${T}_$n.json $n; }
one default X=1
one batch32 TDX_RELAX_BATCH=32
one batch64 TDX_RELAX_BATCH=64
one batch8 TDX_RELAX_BATCH=8
one sequential TDX_FLATS_SEQUENTIAL=1
one default2 X=1
timeout 60 taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 | python3 -c "
import json,sys
d=json.load(sys.stdin); print('tdxbench', d['ms_per_step'], d['pitremove_ms'], d['d8flowdir_ms'], d['aread8_ms'], d['d8flowdir']['ms_class'])"
