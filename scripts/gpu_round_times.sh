#!/bin/bash
# Per-round duration of the tile-relaxation launches of one 16384^2 step (rocprofv3 kernel trace) next to the round's
# active tile count (TDX_DEBUG_ROUNDS=2)  ->  gpurun_out/round_times.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && TDX_DEBUG_ROUNDS=2 TDX_FLATS_SEQUENTIAL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rt -o t -- $R/taudem_amd/bin/tdxbench d8 -n ${1:-16384} -steps 1 -warmup 0 > $R/gpurun_out/rt.log 2>&1)
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/rt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a run of the round schedule = one first_list_kernel followed by its round launches (the trailing empty rounds included)
runs_k, cur = [], None
for r in rows:
    n = r['Kernel_Name']
    if 'first_list_kernel' in n:
        cur = []; runs_k.append(cur)
    elif cur is not None and ('relax_kernel' in n or 'sweep_kernel' in n):
        cur.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp'])))
log = open('gpurun_out/rt.log').read()
runs = re.findall(r"rounds\((\d+) tiles[^)]*\):([ \d]*)", log)
out = open('gpurun_out/round_times.txt', 'w')
out.write(f"{len(runs)} printed runs, {len(runs_k)} traced runs (one first_list_kernel each)\n")
for (ntiles, counts), mine in zip(runs, runs_k):
    counts = [int(c) for c in counts.split()]
    name = mine[0][0].split('<')[1].split(',')[0] if mine else '?'
    tot = sum(d for _, d, _ in mine)
    wall = (mine[-1][2] - mine[0][2]) / 1e3 + mine[-1][1] if mine else 0.0
    out.write(f"run {name} tiles {ntiles}: {len(counts)} rounds, {len(mine)} launches, kernel time {tot/1e3:.2f} ms, wall {wall/1e3:.2f} ms\n")
    for k, (_, d, _) in enumerate(mine):
        out.write(f"   {k:4d} {counts[k] if k < len(counts) else 0:7d} tiles {d:8.1f} us\n")
print(open('gpurun_out/round_times.txt').read()[:200])
PY
rm -rf gpurun_out/rt
