# AreaDinf at 32768^2: per-launch durations of the 64 x 64 tail (and the 16 x 16 bulk) in launch order
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && TDX_DEBUG_ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ad -o t -- $R/taudem_amd/bin/tdxbench dinf -n ${1:-32768} -steps 1 -warmup 0 > $R/gpurun_out/ad.log 2>&1)
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/ad/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
out = open('gpurun_out/r06ad_areadinf_tail_32768.txt', 'w')
for tag in ('dsweep16', 'dsweep32', 'dsweep64'):
    sw = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if tag + '::sweep_kernel' in r['Kernel_Name']]
    if not sw: continue
    out.write(f"{tag}: {len(sw)} launches, {sum(sw)/1e3:.2f} ms; ")
    b = [0, 10, 20, 40, 60, 80, 100, 150, 200, 400, 1000, 5000, 1e9]
    out.write("histogram us " + " ".join(f"<{int(b[i+1])}:{sum(1 for d in sw if b[i] <= d < b[i+1])}({sum(d for d in sw if b[i] <= d < b[i+1])/1e3:.1f}ms)" for i in range(len(b)-1)) + "\n")
    step = max(1, len(sw) // 24)
    for a in range(0, len(sw), step):
        seg = sw[a:a+step]
        out.write(f"   launches {a:5d}..{a+len(seg)-1:5d}: {sum(seg)/1e3:8.2f} ms, mean {sum(seg)/len(seg):8.1f} us, max {max(seg):9.1f} us\n")
out.close()
print(open('gpurun_out/r06ad_areadinf_tail_32768.txt').read())
PY
grep -o "dinf sweep rounds([^)]*):[ 0-9]\{0,2000\}" gpurun_out/ad.log | tail -1 | cut -c1-1500 >> gpurun_out/r06ad_areadinf_tail_32768.txt
rm -rf gpurun_out/ad
