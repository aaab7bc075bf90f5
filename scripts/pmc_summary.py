"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel: calls, per-counter sum and mean per dispatch.
usage: pmc_summary.py <dir with *counter_collection.csv> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(set))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                k = k.replace("(anonymous namespace)::", "").split("(")[0][-90:]
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                calls[k][c].add(row.get("Dispatch_Id", "0"))
    out = {}
    for k, cs in acc.items():
        out[k] = {c: {"sum": v, "dispatches": len(calls[k][c]), "mean_per_dispatch": v / max(1, len(calls[k][c]))} for c, v in cs.items()}
    txt = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    for k in sorted(out, key=lambda k: -max(v["sum"] for v in out[k].values()))[:12]:
        print(k)
        for c, v in sorted(out[k].items()):
            print(f"    {c:24s} sum {v['sum']:.4g}  n {v['dispatches']}  mean {v['mean_per_dispatch']:.4g}")

if __name__ == "__main__":
    main()
