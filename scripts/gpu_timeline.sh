#!/bin/bash
# kernel timeline of ONE 16384^2 step (native harness): every launch in time order with its grid, per stream overlap visible from the offsets
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/timeline
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- $R/taudem_amd/bin/tdxbench d8 -n ${1:-16384} -steps 1 -warmup 1 > $O/run.log 2>&1)
python - <<'PY'
import csv, glob, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/timeline'
f = glob.glob(O + '/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
# the last step: from the last pit_seed_kernel<1> on
idx = max(i for i, r in enumerate(rows) if 'pit_seed_kernel<1>' in r['Kernel_Name'])
rows = rows[idx:]
t0 = int(rows[0]['Start_Timestamp'])
with open(O + '/timeline.txt', 'w') as out:
    for r in rows:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        out.write(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} q{r.get('Queue_Id', '?'):>3} wg{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):7d} {short(r['Kernel_Name'])}\n")
print(open(O + '/timeline.txt').read()[:200])
PY
find $O/tr -name "*.csv" -delete; find $O/tr -name "*.db" -delete
