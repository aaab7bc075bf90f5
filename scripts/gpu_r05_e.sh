#!/bin/bash
# round 5: the DEFAULT bench line with the new config4_8strips_one_gpu leg (memory: the leg closes the main context first), timed; then kernel stats of the big-cell fold (one wave vs trees)
mkdir -p gpurun_out
/usr/bin/time -v timeout 900 python bench.py > gpurun_out/r05e_bench_default.log 2> gpurun_out/r05e_bench_default.err; echo "bench rc=$?"
tail -n 1 gpurun_out/r05e_bench_default.log > gpurun_out/r05e_bench_default.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05e_bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','stage_ms_per_step')})
for k in ('config3','config4_strip','config5_strip','flowalg_16384','config4_8strips_one_gpu'):
    v=d.get(k,{})
    print(k, json.dumps(v)[:900])
PY
grep -E "Elapsed|Maximum resident" gpurun_out/r05e_bench_default.err
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in trees onewave; do
  if [ $V = onewave ]; then export TDX_AD8_BIG_ONE_WAVE=1; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$V -o r -- $R/taudem_amd/bin/tdxbench d8 -n 16384 -steps 3 -warmup 1 > $R/gpurun_out/r05e_tdxbench_$V.json 2>/dev/null)
  find gpurun_out/prof_$V -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/r05e_kernel_stats_d8_$V.csv
  rm -rf gpurun_out/prof_$V
  grep -E "ad8_big|DeviceRadix|ad8_tile|ad8_forest" gpurun_out/r05e_kernel_stats_d8_$V.csv | cut -c1-160
  tail -n 1 gpurun_out/r05e_tdxbench_$V.json | cut -c1-400
done
