#!/bin/bash
# Per-round duration of the tile-relaxation launches of one 16384^2 step (rocprofv3 kernel trace) next to the round's
# active tile count (TDX_DEBUG_ROUNDS=2)  ->  gpurun_out/round_times.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && TDX_DEBUG_ROUNDS=2 TDX_FLATS_SEQUENTIAL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rt -o t -- python $R/bench.py --size 16384 --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/rt.log 2>&1)
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/rt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
relax = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'relax_kernel' in r['Kernel_Name']]
log = open('gpurun_out/rt.log').read()
runs = re.findall(r"rounds\((\d+) tiles\):([ \d]*)", log)
out = open('gpurun_out/round_times.txt', 'w')
i = 0
for ntiles, counts in runs:
    counts = [int(c) for c in counts.split()]
    # launches of this run: rounds in batches 4, 8, 16, 32, 64 ... until an empty round is seen
    n, batch, launched = len(counts), 4, 0
    while launched <= n - 1 or launched == 0:
        launched += batch; batch = min(64, batch * 2)
        if launched > n: break
    mine = relax[i:i + launched]; i += launched
    name = mine[0][0].split('<')[1].split(',')[0] if mine else '?'
    tot = sum(d for _, d in mine)
    out.write(f"run {name} tiles {ntiles}: {n} rounds, {launched} launches, {tot/1e3:.2f} ms\n")
    for k, (_, d) in enumerate(mine):
        out.write(f"   {k:4d} {counts[k] if k < n else 0:7d} tiles {d:8.1f} us\n")
print(open('gpurun_out/round_times.txt').read()[:200]); print('relax launches', len(relax), 'assigned', i)
PY
rm -rf gpurun_out/rt
