#!/bin/bash
# A/B: relaxation kernels compiled for 5 workgroups per CU (scripts/build_variants.sh pit5 / lvl5 / both5)
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03u}
line() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
ok = d['crc']=={'fel': 3868594109, 'p': 3675299354, 'sd8': 3615701311, 'ad8': 2167656781}
print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'pit', d['pitremove_ms'], 'd8', d['d8flowdir_ms'], 'ad8', d['aread8_ms'], 'relax', d['pitremove']['ms_class'][1], d['d8flowdir']['ms_class'][2], 'CRC_OK' if ok else 'CRC_MISMATCH')
" $1; }
run() { n=$1; shift
  env "$@" timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_$n.json 2>> gpurun_out/${T}.err || { echo "$n FAILED rc $?"; return; }
  line gpurun_out/${T}_$n.json; }
V=$GRAFT_REPO_ROOT/taudem_amd/variants
for i in 1 2; do
run base$i X=1
run pit5_$i LD_LIBRARY_PATH=$V/pit5
run lvl5_$i LD_LIBRARY_PATH=$V/lvl5
run both5_$i LD_LIBRARY_PATH=$V/both5
done
tail -n 3 gpurun_out/${T}.err
