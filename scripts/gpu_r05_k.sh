#!/bin/bash
# one-off (round 5, second session): AreaD8's three changes - flat Kahn loop, forest walk with the next node prefetched, in-binade scan of the big-cell fold
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r05k
# 1. the GPU suite on the new defaults, without the D-infinity tests (nothing of theirs changed; the closing run of the round takes all of it)
timeout 1200 python -m pytest tests -m gpu -k "not dinf and not flowalg and not decay" -q --no-header -p no:cacheprovider --timeout=900 --timeout-method=thread --durations=6 2>&1 | tail -n 30 > gpurun_out/${T}_pytest_gpu.txt; tail -n 16 gpurun_out/${T}_pytest_gpu.txt
if ! grep -q " passed" gpurun_out/${T}_pytest_gpu.txt || grep -q "failed" gpurun_out/${T}_pytest_gpu.txt; then
  for cfg in "TDX_AD8_KAHN_NESTED=1" "TDX_AD8_BIG_SCAN=0"; do
    echo "== bisect: $cfg"; env $cfg timeout 600 python -m pytest tests/test_gpu_d8.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -n 5
  done
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
# 2. A/B through the native harness: per-stage ms and CRCs
for cfg in "A=0" "TDX_AD8_KAHN_NESTED=1" "TDX_AD8_BIG_SCAN=0" "TDX_AD8_KAHN_NESTED=1 TDX_AD8_BIG_SCAN=0" "TDX_AD8_BIG_SCAN=1" "A=1"; do
  echo "== $cfg" >> gpurun_out/${T}_ab_16384.txt
  env $cfg taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 -warmup 2 -crc 2>/dev/null | tail -n 1 >> gpurun_out/${T}_ab_16384.txt
done
python - <<'PY'
import json
lines = open("gpurun_out/r05k_ab_16384.txt").read().splitlines()
for i in range(0, len(lines) - 1, 2):
    try:
        d = json.loads(lines[i + 1])
        print(lines[i], d["ms_per_step"], d["pitremove_ms"], d["d8flowdir_ms"], d["aread8_ms"], d["aread8"]["ms_class"], d["crc"], "EXPECT ad8 2167656781")
    except Exception as e:
        print(lines[i], "??", lines[i + 1][:200])
PY
# 3. kernel stats of the default (AreaD8 kernels)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -o r -- $GRAFT_REPO_ROOT/taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 -warmup 1 > /dev/null 2>&1)
find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats_tdxbench_16384.csv
rm -rf gpurun_out/prof_$T
grep -E "ad8_|forest" gpurun_out/${T}_kernel_stats_tdxbench_16384.csv | cut -d, -f1-4 | cut -c1-160
# 4. eight strips of configs[3] on this GPU: segment trace -> projection (new defaults, then the old big-cell loop)
for V in new oldfold; do
  E="A=0"; [ $V = oldfold ] && E="TDX_AD8_BIG_SCAN=0"
  env $E TDX_COMM_TRACE=1 timeout 600 python bench.py --gpus 8 --in-process --workload d8 --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/${T}_segments_8strips_d8_$V.json > gpurun_out/${T}_8strips_65536_d8_$V.log 2> /dev/null
  tail -n 1 gpurun_out/${T}_8strips_65536_d8_$V.log > gpurun_out/${T}_8strips_65536_d8_$V.json; rm -f gpurun_out/${T}_8strips_65536_d8_$V.log
  python scripts/project_8gpu.py gpurun_out/${T}_segments_8strips_d8_$V.json > gpurun_out/${T}_projection_8gpu_d8_$V.txt; echo "== 8 strips, $V"; grep -E "aread8|total|pitremove \||d8flowdir \|" gpurun_out/${T}_projection_8gpu_d8_$V.txt | cut -c1-150
done
