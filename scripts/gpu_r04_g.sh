#!/bin/bash
# Round 4, call G: walker's evaluation over the contributors that exist (dinf_sweep_tile.inc: eval_walk) - parity, CRCs, timings
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 100 taudem_amd/bin/tdxbench dinf -n 4096 -steps 1 -crc > $O/canary_dinf_4096.json 2>&1; grep -o '"crc".*' $O/canary_dinf_4096.json
TDX_SWEEP_VERIFY=1 timeout 900 python -m pytest tests/test_gpu_dinf.py tests/test_flowalg.py -m gpu -q --no-header -p no:cacheprovider --timeout=600 --timeout-method=thread -x 2>&1 | tail -n 5
for i in 1 2; do timeout 200 taudem_amd/bin/tdxbench dinf -n 16384 -steps 2 -crc >> $O/tdxbench_dinf_16384.jsonl 2>&1; done
timeout 300 taudem_amd/bin/tdxbench dinf -n 32768 -steps 2 -crc > $O/tdxbench_dinf_32768.json 2>&1
timeout 300 taudem_amd/bin/tdxbench decay -steps 2 -crc > $O/tdxbench_decay.json 2>&1
python - <<'PY'
import json
for f in ("tdxbench_dinf_16384.jsonl", "tdxbench_dinf_32768.json", "tdxbench_decay.json"):
    for l in open("gpurun_out/r04g/" + f):
        try: d = json.loads(l)
        except Exception: print("unparsed", l[:300]); continue
        print(f, d["nx"], {k: round(d[k], 2) for k in d if k.endswith("_ms")}, d.get("ms_per_step"), d.get("crc"), {k: d[k]["rounds"] for k in ("areadinf", "dinfdecayaccum") if k in d})
PY
