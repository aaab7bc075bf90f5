#!/bin/bash
# A/B of tile-engine builds (scripts/build_variants.sh) and of the tail batch length; every variant must reproduce the CRCs
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03r}
line() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
ok = d['crc']=={'fel': 3868594109, 'p': 3675299354, 'sd8': 3615701311, 'ad8': 2167656781}
print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'pit', d['pitremove_ms'], 'd8', d['d8flowdir_ms'], 'ad8', d['aread8_ms'], 'relax', d['pitremove']['ms_class'][1], d['d8flowdir']['ms_class'][2], 'CRC_OK' if ok else 'CRC_MISMATCH')
" $1; }
run() { # name, env...
  n=$1; shift
  env "$@" timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_$n.json 2>> gpurun_out/${T}.err || { echo "$n FAILED rc $?"; return; }
  line gpurun_out/${T}_$n.json
}
V=$GRAFT_REPO_ROOT/taudem_amd/variants
run base X=1
run noskip LD_LIBRARY_PATH=$V/noskip
run addc LD_LIBRARY_PATH=$V/addc
run base2 X=1
run noskip2 LD_LIBRARY_PATH=$V/noskip
run addc2 LD_LIBRARY_PATH=$V/addc
run tail8 TDX_RELAX_TAIL_BATCH=8
run tail4 TDX_RELAX_TAIL_BATCH=4
run sweeps8 TDX_MAX_SWEEPS=8
run sweeps32 TDX_MAX_SWEEPS=32
tail -n 3 gpurun_out/${T}.err
