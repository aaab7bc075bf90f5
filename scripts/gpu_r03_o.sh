#!/bin/bash
# timeline of ONE steady-state step of the D8 pipeline at 16384^2 (rocprofv3 kernel trace of the native harness) + the tile engine's phase clocks
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
T=${1:-r03o}
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$T -o t -- $R/taudem_amd/bin/tdxbench d8 -n 16384 -steps 2 -warmup 1 > $R/gpurun_out/${T}_trace.log 2>&1)
python3 - $T <<'PY'
import csv, glob, sys
T = sys.argv[1]
f = glob.glob(f'gpurun_out/tl_{T}/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seeds = [i for i, r in enumerate(rows) if 'pit_seed_kernel' in r['Kernel_Name']]
rows = rows[seeds[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    if 'relax_kernel' in n:
        return 'relax<' + ('Pit' if 'PitOp' in n else ('Level' if 'LevelOpT<1' in n else 'Reach')) + '>'
    return n.split('(')[0][:48]
with open(f'gpurun_out/{T}_timeline.txt', 'w') as out:
    out.write('# start_us dur_us queue kernel grid\n')
    for r in rows:
        s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        out.write(f"{s:10.1f} {d:8.1f} {r.get('Queue_Id','?'):>3} {short(r['Kernel_Name'])} {r.get('Grid_Size_X', r.get('Grid_Size',''))}\n")
print('rows', len(rows), 'span ms', (int(rows[-1]['End_Timestamp']) - t0) / 1e6)
PY
rm -rf gpurun_out/tl_$T
TDX_DEBUG_ROUNDS=1 timeout 60 taudem_amd/bin/tdxbench d8 -n 16384 -steps 1 -warmup 0 > /dev/null 2> gpurun_out/${T}_phase_clocks.txt
tail -n 30 gpurun_out/${T}_phase_clocks.txt | cut -c1-400
