#!/bin/bash
# after a tile-engine change: CRC canary + timing, the engine's phase clocks, AreaD8's tile-kernel phase clocks
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03s}
line() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
ok = d['crc']=={'fel': 3868594109, 'p': 3675299354, 'sd8': 3615701311, 'ad8': 2167656781}
print(sys.argv[1].split('/')[-1], 'ms', d['ms_per_step'], 'pit', d['pitremove_ms'], 'd8', d['d8flowdir_ms'], 'ad8', d['aread8_ms'], 'relax', d['pitremove']['ms_class'][1], d['d8flowdir']['ms_class'][2], 'CRC_OK' if ok else 'CRC_MISMATCH')
" $1; }
timeout 60 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8.json 2>> gpurun_out/${T}.err || { echo "FAILED rc $?"; exit 1; }
line gpurun_out/${T}_d8.json
TDX_DEBUG_ROUNDS=1 timeout 60 $B d8 -n 16384 -steps 1 -warmup 0 > /dev/null 2> gpurun_out/${T}_phase_clocks.txt
grep -A1 "tile_relax_run" gpurun_out/${T}_phase_clocks.txt | grep -o "tile_relax_run.*\|cycles per.*" 
TDX_AD8_DEBUG=1 timeout 60 $B d8 -n 16384 -steps 1 -warmup 1 > /dev/null 2> gpurun_out/${T}_ad8_phases.txt
cat gpurun_out/${T}_ad8_phases.txt | tail -n 3
