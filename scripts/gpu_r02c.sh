#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && TDX_DEBUG_ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rtd -o t -- python $R/scripts/bench_dinf.py --size 16384 --steps 1 --warmup 0 > $R/gpurun_out/rtd.log 2>&1)
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/rtd/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sw = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'sweep_kernel' in r['Kernel_Name']]
log = open('gpurun_out/rtd.log').read()
m = re.search(r"dinf sweep rounds\((\d+) tiles\):([ \d]*)", log)
counts = [int(c) for c in m.group(2).split()]
out = open('gpurun_out/r02c_dinf_round_times.txt', 'w')
out.write(f"dinf sweep at 16384^2: {len(counts)} rounds, {len(sw)} launches, {sum(sw)/1e3:.2f} ms, {sum(counts)} tile activations\n")
for k, d in enumerate(sw):
    out.write(f"  {k:4d} {counts[k] if k < len(counts) else 0:7d} tiles {d:8.1f} us\n")
print(open('gpurun_out/r02c_dinf_round_times.txt').read()[:3000])
PY
rm -rf gpurun_out/rtd
