#!/bin/bash
export TMPDIR=/tmp
export TDX_DINF_TILES=1
TDX_DEBUG_ROUNDS=1 timeout 600 python scripts/bench_dinf.py --size 16384 --steps 1 --warmup 0 2>&1 | grep "rounds .*activations" | head -40
