"""Kernel timeline of a traced run: python scripts/timeline.py <dir with *kernel_trace.csv> [substring ...]
Prints per-kernel totals, and for the kernels whose name contains one of the substrings the launches in time order (start offset us, duration us)."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]
tot, cnt = defaultdict(float), defaultdict(int)
for r in rows:
    k = short(r["Kernel_Name"]); tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[k] += 1
for k in sorted(tot, key=lambda k: -tot[k])[:25]:
    print(f"{tot[k] / 1e3:9.3f} ms {cnt[k]:6d} x  {k}")
if subs:
    t0 = None
    line = []
    for r in rows:
        k = short(r["Kernel_Name"])
        if not any(s in k for s in subs):
            continue
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None: t0 = st
        line.append(f"{(st - t0) / 1e3:.0f}:{(en - st) / 1e3:.1f}")
    print("start_us:duration_us of", subs)
    print(" ".join(line))
