export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && TDX_FLATS_MACRO=${1:-8} timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mt -o t -- $R/taudem_amd/bin/tdxbench d8 -n 16384 -steps 1 -warmup 1 > $R/gpurun_out/mt.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/mt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = max(i for i, r in enumerate(rows) if 'pit_seed_kernel<1>' in r['Kernel_Name'])
rows = rows[idx:]
t0 = int(rows[0]['Start_Timestamp'])
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
with open('gpurun_out/r06h_timeline_macro.txt', 'w') as out:
    for r in rows:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        out.write(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} q{r.get('Queue_Id', '?'):>3} wg{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):7d} {short(r['Kernel_Name'])}\n")
PY
rm -rf gpurun_out/mt
