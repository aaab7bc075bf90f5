export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06l}
for m in 0 8; do echo -n "dinf 16384 TDX_FLATS_MACRO=$m  "; TDX_FLATS_MACRO=$m taudem_amd/bin/tdxbench dinf -n 16384 -steps 3 -crc 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','dinfflowdir_ms','areadinf_ms')}, d['crc'], d['dinfflowdir']['rounds'], d['dinfflowdir']['ms_class'])"; done > gpurun_out/${T}_macro_dinf_ab.txt 2>&1
for m in 0 8; do echo -n "dinf 32768 TDX_FLATS_MACRO=$m  "; TDX_FLATS_MACRO=$m taudem_amd/bin/tdxbench dinf -n 32768 -steps 2 -crc 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','dinfflowdir_ms','areadinf_ms')}, d['crc'], d['dinfflowdir']['rounds'], d['dinfflowdir']['ms_class'])"; done >> gpurun_out/${T}_macro_dinf_ab.txt 2>&1
for m in 0 8; do echo -n "d8 16384 TDX_FLATS_MACRO=$m  "; TDX_FLATS_MACRO=$m taudem_amd/bin/tdxbench d8 -n 16384 -steps 8 -crc 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','pitremove_ms','d8flowdir_ms','aread8_ms')}, d['crc'], d['d8flowdir']['rounds'])"; done >> gpurun_out/${T}_macro_dinf_ab.txt 2>&1
cat gpurun_out/${T}_macro_dinf_ab.txt
timeout 900 python -m pytest tests/test_gpu_d8.py tests/test_gpu_dinf.py tests/test_gpu_pathological.py tests/test_gpu_fuzz_strips.py tests/test_gpu_large_golden.py tests/test_strips.py -m gpu -q --no-header -p no:cacheprovider --timeout=600 -x 2>&1 | tail -n 5
