export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for tb in 0 4 6 8; do echo -n "TDX_RELAX_TAIL_BATCH=$tb  "; if [ $tb = 0 ]; then unset TDX_RELAX_TAIL_BATCH; else export TDX_RELAX_TAIL_BATCH=$tb; fi; taudem_amd/bin/tdxbench d8 -n 16384 -steps 8 -crc 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','pitremove_ms','d8flowdir_ms','aread8_ms')}, d['crc']['fel'], d['crc']['p'])"; done; done > gpurun_out/r06n_tail_batch.txt 2>&1
cat gpurun_out/r06n_tail_batch.txt
