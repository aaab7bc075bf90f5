#!/bin/bash
# Round-3 step B: find the test that hung in step A (every command under a short timeout; pytest-timeout dumps the Python stacks)
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03b}
timeout 90 $B d8 -n 16384 -steps 3 -crc > gpurun_out/${T}_d8.json 2> gpurun_out/${T}_d8.err
TDX_SOLO_CHAIN=0 timeout 90 $B d8 -n 16384 -steps 3 -crc > gpurun_out/${T}_d8_nochain.json 2>> gpurun_out/${T}_d8.err
timeout 90 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf.json 2> gpurun_out/${T}_dinf.err
TDX_SOLO_CHAIN=0 timeout 90 $B dinf -n 16384 -steps 2 -crc > gpurun_out/${T}_dinf_nochain.json 2>> gpurun_out/${T}_dinf.err
TDX_SWEEP_VERIFY=2 timeout 90 $B dinf -n 16384 -steps 1 -warmup 0 -crc > gpurun_out/${T}_dinf_verify.json 2>> gpurun_out/${T}_dinf.err
for f in gpurun_out/${T}_*.json; do echo "== $f"; cut -c1-330 $f; done
tail -n 5 gpurun_out/${T}_dinf.err gpurun_out/${T}_d8.err
TDX_SOLO_CHAIN=0 timeout 300 python -m pytest "tests/test_flowalg.py::test_every_sweep_tool_on_both_tile_geometries" -q --no-header -p no:cacheprovider -x --timeout=200 --timeout-method=thread 2>&1 | tail -30 > gpurun_out/${T}_t17_nochain.txt; tail -n 12 gpurun_out/${T}_t17_nochain.txt
timeout 300 python -m pytest "tests/test_flowalg.py::test_every_sweep_tool_on_both_tile_geometries" -q --no-header -p no:cacheprovider -x --timeout=200 --timeout-method=thread 2>&1 | tail -60 > gpurun_out/${T}_t17.txt; tail -n 40 gpurun_out/${T}_t17.txt
