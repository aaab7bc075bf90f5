export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r06g}
for rep in 1 2; do for w in 1 2 4 8 16; do echo -n "TDX_MACRO_WGS=$w  "; TDX_MACRO_WGS=$w taudem_amd/bin/tdxbench d8 -n 16384 -steps 8 -crc 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','pitremove_ms','d8flowdir_ms','aread8_ms')}, d['crc']['p'], d['d8flowdir']['rounds'], d['d8flowdir']['ms_class'][2])"; done; done > gpurun_out/${T}_macro_wgs.txt 2>&1
cat gpurun_out/${T}_macro_wgs.txt
