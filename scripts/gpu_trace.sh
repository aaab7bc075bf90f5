#!/bin/bash
# rocprofv3 kernel stats of one short 16384^2 bench run -> gpurun_out/trace/
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace -o t -- python $R/bench.py --size 16384 --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/trace.log 2>&1)
find gpurun_out/trace -name "*kernel_trace.csv" -delete; find gpurun_out/trace -name "*.db" -delete
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r['Name'].replace('(anonymous namespace)::','').split('(')[0][-60:].ljust(60), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
