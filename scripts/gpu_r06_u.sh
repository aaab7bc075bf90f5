mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_flowalg.py tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_gpu_pathological.py tests/test_gpu_fuzz_strips.py tests/test_gpu_cli.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r06u_tests.txt
cat gpurun_out/r06u_tests.txt
for m in "dinf 4096" "dinf 16384"; do set -- $m; taudem_amd/bin/tdxbench $1 -n $2 -steps 2 -crc 2>/dev/null | tail -n 1; done > gpurun_out/r06u_tdxbench_crc.jsonl
taudem_amd/bin/tdxbench decay -steps 1 -crc 2>/dev/null | tail -n 1 >> gpurun_out/r06u_tdxbench_crc.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r06u_tdxbench_crc.jsonl'):
    d = json.loads(l); print(d['mode'], d['nx'], d['ms_per_step'], d.get('crc'))
PY
TDX_SWEEP_VERIFY=1 timeout 600 python scripts/bench_flowalg.py --digest 2>&1 | tail -1 > gpurun_out/r06u_flowalg_verify.json
timeout 600 python scripts/bench_flowalg.py 2>&1 | tail -1 > gpurun_out/r06u_flowalg.json
cat gpurun_out/r06u_flowalg.json
