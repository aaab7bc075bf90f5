#!/bin/bash
# Round 4, call M: kernel timeline of one D8 pipeline step at 16384^2 (tail rounds: duration of a round and the gap to the next launch)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m
mkdir -p $O
cd /tmp
TDX_FLATS_SEQUENTIAL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- $R/taudem_amd/bin/tdxbench d8 -n 16384 -steps 1 -warmup 1 > $O/trace.log 2>&1
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04m"
f = glob.glob(O + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
# second step only: after the last synth / first half
t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
half = len(t) // 2
t = t[half:]
t0 = t[0][0]
out = open(O + "/timeline_d8_16384.txt", "w")
prev_end = None
for s, e, k in t:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} gap {gap:6.1f}  {k}\n")
    prev_end = e
out.close()
# summary: relax launches with duration < 40 us: mean duration and mean gap
import statistics
for name in ("PitOp", "LevelOp"):
    d = [(e - s) / 1e3 for s, e, k in t if name in k]
    small = [x for x in d if x < 40]
    gaps = []
    for i in range(1, len(t)):
        if name in t[i][2] and name in t[i - 1][2] and (t[i][1] - t[i][0]) / 1e3 < 40: gaps.append((t[i][0] - t[i - 1][1]) / 1e3)
    print(name, "launches", len(d), "total ms %.2f" % (sum(d) / 1e3), "small (<40us)", len(small), "mean %.1f us" % statistics.mean(small), "mean gap before a small launch %.2f us" % statistics.mean(gaps))
print("step span ms %.2f" % ((t[-1][1] - t0) / 1e6))
PY
rm -rf $O/tr
