export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for v in 8 4 2 1; do
  (cd /tmp && TDX_MAX_SWEEPS=$v TDX_FLATS_SEQUENTIAL=1 timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ns_$v -o t -- $R/taudem_amd/bin/tdxbench d8 -n 16384 -steps 1 -warmup 0 > $R/gpurun_out/ns_$v.log 2>&1)
  python - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob(f'gpurun_out/ns_{v}/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    if 'relax_kernel' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if d > 150: out.append(f"{'Pit' if 'PitOp' in n else 'Lvl'}:{d:.0f}")
print('max_sweeps', v, ' '.join(out))
PY
  rm -rf gpurun_out/ns_$v gpurun_out/ns_$v.log
done > gpurun_out/r06f_max_sweeps_round0.txt 2>&1
cat gpurun_out/r06f_max_sweeps_round0.txt
