#!/bin/bash
# round 5: A/B of the big-cell fold (one wave / trees) x (all eight suffix adds / only the non-zero ones), then the default bench line with the eight-strip leg
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for LIB in default alladds; do
 for V in trees onewave; do
  unset TDX_AD8_BIG_ONE_WAVE; if [ $V = onewave ]; then export TDX_AD8_BIG_ONE_WAVE=1; fi
  if [ $LIB = alladds ]; then export LD_LIBRARY_PATH=$R/taudem_amd/variants/alladds; else unset LD_LIBRARY_PATH; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -o r -- $R/taudem_amd/bin/tdxbench d8 -n 16384 -steps 3 -warmup 1 -crc > $R/gpurun_out/r05f_tdxbench_${LIB}_$V.json 2>/dev/null)
  echo "$LIB $V: $(find gpurun_out/prof_x -name '*kernel_stats.csv' | head -n 1 | xargs grep ad8_big_fold | awk -F'",' '{print $2}')  $(tail -n 1 gpurun_out/r05f_tdxbench_${LIB}_$V.json | grep -o '"aread8_ms": [0-9.]*')  $(tail -n 1 gpurun_out/r05f_tdxbench_${LIB}_$V.json | grep -o '"crc_ad8": "[0-9a-f]*"')"
  rm -rf gpurun_out/prof_x
 done
done
unset LD_LIBRARY_PATH TDX_AD8_BIG_ONE_WAVE
S=$(date +%s)
timeout 900 python bench.py > gpurun_out/r05f_bench_default.log 2> gpurun_out/r05f_bench_default.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"
tail -n 1 gpurun_out/r05f_bench_default.log > gpurun_out/r05f_bench_default.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05f_bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','stage_ms_per_step')})
for k in ('config3','config4_strip','config5_strip','flowalg_16384','config4_8strips_one_gpu'):
    v=d.get(k,{})
    print(k, json.dumps(v)[:1100])
PY
tail -3 gpurun_out/r05f_bench_default.err
