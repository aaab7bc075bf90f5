mkdir -p gpurun_out
( for n in 4096 16384 32768; do
  for v in "TDX_FLATS_LIST=1" "TDX_FLATS_MACRO=0" "TDX_FLATS_MACRO=8"; do
    echo -n "dinf $n $v  "
    env $v taudem_amd/bin/tdxbench dinf -n $n -steps 2 -crc 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['dinfflowdir']
print({k:d[k] for k in ['ms_per_step','dinfflowdir_ms','areadinf_ms']}, d.get('crc'), f['rounds'], f['ms_class'])"
  done
done ) > gpurun_out/r06ae_dinf_stream_classify.txt 2>&1
cat gpurun_out/r06ae_dinf_stream_classify.txt
timeout 1200 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_gpu_fuzz_strips.py tests/test_gpu_cli.py tests/test_gpu_pathological.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -4
