export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && TDX_DEBUG_ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rv -o t -- python $R/scripts/bench_flowalg.py --only aread8_weighted > $R/gpurun_out/rv.log 2>&1)
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/rv/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sw = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows if 'sweep_kernel' in r['Kernel_Name'] and 'dsweep' not in r['Kernel_Name']]
log = open('gpurun_out/rv.log').read()
runs = re.findall(r"d8 sweep rounds\((\d+) tiles of (\d+)\):([ \d]*)", log)
out = open('gpurun_out/r06v_aread8w_round_times.txt', 'w')
out.write(f"aread8_weighted at 16384^2: {len(sw)} sweep launches traced, {len(runs)} printed runs (warm-up call + timed call)\n")
names = sorted(set(n for n, _, _ in sw))
for n in names: out.write(f"  kernel {n[:110]}: {sum(1 for a in sw if a[0]==n)} launches, {sum(a[1] for a in sw if a[0]==n)/1e3:.1f} ms\n")
# the last call = second half of the launches
half = sw[len(sw)//2:]
counts = [int(c) for r in runs[len(runs)//2:] for c in r[2].split()]
out.write(f"timed call: {len(half)} launches, {sum(d for _,d,_ in half)/1e3:.1f} ms kernel time; rounds printed {len(counts)}, tile activations {sum(counts)}\n")
step = max(1, len(half)//16)
for a in range(0, len(half), step):
    seg = half[a:a+step]; cs = counts[a:a+step]
    out.write(f"   launches {a:5d}..{a+len(seg)-1:5d}: {sum(d for _,d,_ in seg)/1e3:7.2f} ms, mean {sum(d for _,d,_ in seg)/len(seg):7.1f} us per launch, tiles per round ~{(sum(cs)//max(1,len(cs)))}, grid {seg[0][2]}\n")
print(open('gpurun_out/r06v_aread8w_round_times.txt').read())
PY
rm -rf gpurun_out/rv
