#!/bin/bash
# one-off (round 5, second session): blocked chains through the big-cell scan, the fold's LDS ring, both entry walks of the tile kernels in one loop, apply kernel at six tiles per CU (A/B)
export TMPDIR=/tmp
mkdir -p gpurun_out
T=r05l
timeout 1200 python -m pytest tests -m gpu -k "not dinf and not flowalg and not decay" -q --no-header -p no:cacheprovider --timeout=900 --timeout-method=thread --durations=6 2>&1 | tail -n 30 > gpurun_out/${T}_pytest_gpu.txt; tail -n 14 gpurun_out/${T}_pytest_gpu.txt
for cfg in "A=0" "TDX_AD8_APPLY_OCC6=1" "A=1" "TDX_AD8_APPLY_OCC6=1 B=1"; do
  echo "== $cfg" >> gpurun_out/${T}_ab_16384.txt
  env $cfg taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 -warmup 2 -crc 2>/dev/null | tail -n 1 >> gpurun_out/${T}_ab_16384.txt
done
python - <<'PY'
import json
lines = open("gpurun_out/r05l_ab_16384.txt").read().splitlines()
for i in range(0, len(lines) - 1, 2):
    try:
        d = json.loads(lines[i + 1])
        print(lines[i], d["ms_per_step"], d["pitremove_ms"], d["d8flowdir_ms"], d["aread8_ms"], d["aread8"]["ms_class"], d["crc"]["ad8"], "EXPECT ad8 2167656781")
    except Exception as e:
        print(lines[i], "??", lines[i + 1][:200])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -o r -- $GRAFT_REPO_ROOT/taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 -warmup 1 > /dev/null 2>&1)
find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats_tdxbench_16384.csv
rm -rf gpurun_out/prof_$T
python - <<'PY'
import csv
for r in csv.reader(open('gpurun_out/r05l_kernel_stats_tdxbench_16384.csv')):
    if 'ad8_' in r[0]: print(r[0].split('(')[0][-50:], r[1], r[3])
PY
(cd /tmp && TDX_AD8_APPLY_OCC6=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -o r -- $GRAFT_REPO_ROOT/taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 -warmup 1 > /dev/null 2>&1)
find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats_tdxbench_16384_occ6.csv
rm -rf gpurun_out/prof_$T
grep "ad8_tile_apply" gpurun_out/${T}_kernel_stats_tdxbench_16384_occ6.csv | cut -d, -f2-4
for V in new; do
  TDX_COMM_TRACE=1 timeout 600 python bench.py --gpus 8 --in-process --workload d8 --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/${T}_segments_8strips_d8_$V.json > gpurun_out/${T}_8strips_65536_d8_$V.log 2> /dev/null
  tail -n 1 gpurun_out/${T}_8strips_65536_d8_$V.log > gpurun_out/${T}_8strips_65536_d8_$V.json; rm -f gpurun_out/${T}_8strips_65536_d8_$V.log
  python scripts/project_8gpu.py gpurun_out/${T}_segments_8strips_d8_$V.json > gpurun_out/${T}_projection_8gpu_d8_$V.txt; echo "== 8 strips, $V"; grep -E "aread8|total|pitremove \||d8flowdir \|" gpurun_out/${T}_projection_8gpu_d8_$V.txt | cut -c1-150
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05l_segments_8strips_d8_new.json')); logs=d['logs']
for i in range(len(logs[0])):
    s0=logs[0][i]
    if s0[0]=='aread8' and s0[1]=='big cells' and s0[2]==0: print(i,[round(lg[i][3],2) for lg in logs])
PY
