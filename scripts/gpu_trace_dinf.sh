#!/bin/bash
# rocprofv3 kernel stats of the D-infinity configuration (DinfFlowDir + AreaDinf at 16384^2) -> gpurun_out/trace_dinf/
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_dinf -o t -- python $R/scripts/bench_dinf.py --size 16384 > $R/gpurun_out/trace_dinf.log 2>&1)
find gpurun_out/trace_dinf -name "*kernel_trace.csv" -delete; find gpurun_out/trace_dinf -name "*.db" -delete
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace_dinf/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'].replace('(anonymous namespace)::','').split('(')[0][-60:].ljust(60), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
