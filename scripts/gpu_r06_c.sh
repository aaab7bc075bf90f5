export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_d8.py -m gpu -q --no-header -p no:cacheprovider -x -k "codes_outside or quirks or kahn or tiled_path or golden_aread8" 2>&1 | tail -n 8 > gpurun_out/r06c_pytest_codes.txt
bash scripts/gpu_round_times.sh > /dev/null 2>&1; mv gpurun_out/round_times.txt gpurun_out/r06c_round_times.txt
(TDX_DEBUG_ROUNDS=1 TDX_FLATS_SEQUENTIAL=1 taudem_amd/bin/tdxbench d8 -n 16384 -steps 1 2>&1 | grep -v "^{" | cut -c1-400) > gpurun_out/r06c_phase_clocks.txt
cat gpurun_out/r06c_pytest_codes.txt | tail -3
