#!/bin/bash
# round 5: the D-infinity bulk rounds on 16 x 16 one-wave tiles (TDX_DINF_BULK_TILE=16) against the 32 x 32 default: tests under the verifier, CRCs and times at 16384^2 / 32768^2, phase clocks
mkdir -p gpurun_out
export TDX_DINF_BULK_TILE=16
TDX_SWEEP_VERIFY=1 timeout 900 python -m pytest tests/test_gpu_dinf.py -m gpu -q -x > gpurun_out/r05g_pytest_dinf_tile16.txt 2>&1; echo "pytest(tile16) rc=$?"; tail -3 gpurun_out/r05g_pytest_dinf_tile16.txt
unset TDX_DINF_BULK_TILE
for N in 16384 32768; do
 for T in 32 16; do
  for U in 400 6000; do
   echo "n=$N tile=$T until=$U: $(TDX_DINF_BULK_TILE=$T TDX_DINF_BULK_UNTIL=$U timeout 300 taudem_amd/bin/tdxbench dinf -n $N -steps 2 -warmup 1 -crc 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in d if k in ('ms_per_step','dinfflowdir_ms','areadinf_ms','crc_sca','crc_ang')}, 'accum', d.get('areadinf',{}).get('ms_class'), 'rounds', d.get('areadinf',{}).get('rounds'))")"
  done
 done
done | tee gpurun_out/r05g_dinf_tile16_vs_32.txt
TDX_DEBUG_ROUNDS=1 TDX_DINF_BULK_TILE=16 timeout 300 taudem_amd/bin/tdxbench dinf -n 16384 -steps 1 -warmup 0 2>&1 | grep -A2 "dinf sweep rounds\|rounds .*activations" | cut -c1-400 | head -60 > gpurun_out/r05g_dinf_phase_clocks_tile16.txt
head -30 gpurun_out/r05g_dinf_phase_clocks_tile16.txt
