# the seeded fuzz of the strip protocol, longer than the suite's 32 seeds: 120 seeds, then 24 seeds on rasters up to 3 x larger on each side
mkdir -p gpurun_out
T=${1:-r06zzz}
TDX_FUZZ_SEEDS=120 timeout 900 python -m pytest tests/test_gpu_fuzz_strips.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/${T}_fuzz_120_seeds.txt; cat gpurun_out/${T}_fuzz_120_seeds.txt
TDX_FUZZ_SEEDS=24 TDX_FUZZ_SCALE=3 timeout 900 python -m pytest tests/test_gpu_fuzz_strips.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/${T}_fuzz_24_seeds_scale3.txt; cat gpurun_out/${T}_fuzz_24_seeds_scale3.txt
