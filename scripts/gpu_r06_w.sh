mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_flowalg.py tests/test_gpu_gridnet.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -3
TDX_SWEEP_VERIFY=1 timeout 600 python scripts/bench_flowalg.py --digest 2>&1 | tail -1 > gpurun_out/r06w_flowalg_verify.json
timeout 600 python scripts/bench_flowalg.py 2>&1 | tail -1 > gpurun_out/r06w_flowalg.json
cat gpurun_out/r06w_flowalg.json
python - <<'PY'
import json
a = json.load(open('gpurun_out/r06w_flowalg_verify.json'))['ms']; b = json.load(open('profiles/r06u_flowalg_16384_verify_digests.json'))['ms']
print("digests equal:", all(a[k] == b[k] for k in a if k.endswith('_crc')), [k for k in a if k.endswith('_crc') and a[k] != b[k]])
PY
