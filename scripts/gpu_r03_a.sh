#!/bin/bash
# Round-3 step A: the int16 level fields, the solo-round hand-over and the sweep verifier on the box - native harness A/B runs
# (no Python start-up), then the full GPU suite.
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03a}
timeout 120 $B d8 -n 16384 -steps 3 -crc > gpurun_out/${T}_d8.json 2> gpurun_out/${T}_d8.err
TDX_SOLO_CHAIN=0 timeout 120 $B d8 -n 16384 -steps 3 -crc > gpurun_out/${T}_d8_nochain.json 2>> gpurun_out/${T}_d8.err
TDX_SWEEP_VERIFY=2 timeout 120 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf_verify.json 2> gpurun_out/${T}_dinf.err
timeout 120 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf.json 2>> gpurun_out/${T}_dinf.err
TDX_SOLO_CHAIN=0 timeout 120 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf_nochain.json 2>> gpurun_out/${T}_dinf.err
timeout 200 $B dinf -n 32768 -steps 1 -crc > gpurun_out/${T}_dinf_32768.json 2>> gpurun_out/${T}_dinf.err
TDX_SWEEP_VERIFY=2 timeout 300 $B decay -nx 65536 -ny 8192 -steps 1 -crc > gpurun_out/${T}_decay_strip.json 2> gpurun_out/${T}_decay.err
for f in gpurun_out/${T}_*.json; do echo "== $f"; cut -c1-420 $f; done
tail -5 gpurun_out/${T}_dinf.err gpurun_out/${T}_decay.err gpurun_out/${T}_d8.err
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8 > gpurun_out/${T}_pytest_gpu.txt; tail -6 gpurun_out/${T}_pytest_gpu.txt
