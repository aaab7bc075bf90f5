#!/bin/bash
# Round 4, call L: the D8 pipeline ten times in fresh processes - spread of the stage times and of the streaming slope stencil, with the rasters' addresses
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
for i in 1 2 3 4 5 6 7 8 9 10; do TDXBENCH_ADDR=1 timeout 120 taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 >> $O/ten_processes.jsonl 2>> $O/ten_processes_addr.txt; done
python - <<'PY'
import json, statistics
rows = [json.loads(l) for l in open("gpurun_out/r04l/ten_processes.jsonl")]
addr = open("gpurun_out/r04l/ten_processes_addr.txt").read().strip().splitlines()
def col(f): return [f(d) for d in rows]
step = col(lambda d: d["ms_per_step"]); slope = col(lambda d: d["d8flowdir"]["ms_class"][0]); bfs = col(lambda d: d["d8flowdir"]["ms_class"][2]); relax = col(lambda d: d["pitremove"]["ms_class"][1])
for name, v in (("ms_per_step", step), ("slope_stencil_ms", slope), ("levels_ms", bfs), ("pit_relax_ms", relax)):
    print(name, "median %.3f min %.3f max %.3f spread %.1f %%" % (statistics.median(v), min(v), max(v), 100 * (max(v) - min(v)) / statistics.median(v)), [round(x, 3) for x in v])
print("\n".join(addr[:10]))
PY
