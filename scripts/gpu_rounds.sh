#!/bin/bash
# Schedule statistics of the tile engine (active tiles per round, cycles per activation) for one 16384^2 step.
export TMPDIR=/tmp
mkdir -p gpurun_out
TDX_DEBUG_ROUNDS=1 TDX_FLATS_SEQUENTIAL=1 timeout 300 python bench.py --size ${1:-16384} --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/rounds.log 2>&1
grep -c . gpurun_out/rounds.log
