"""GridNet + Threshold on the D8 rasters of the headline configuration, one MI355X, HBM-resident.
usage: python scripts/bench_gridnet.py [--size 16384] [--steps 2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taudem_amd as T

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--warmup", type=int, default=1)
a = ap.parse_args()
n = a.size
ctx = T.Context(0)
dem = ctx.synth_dem(n, seed=1234)
fel = ctx.pitremove(dem, -9999.0)
del dem
p, sd8 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0)
del fel, sd8
ad8 = ctx.aread8(p, -32768)
def step():
    _, _, _, s1 = ctx.gridnet(p, -32768, 30.0, 30.0, stats=True)
    _, s2 = ctx.threshold(ad8, 100.0, -1.0, stats=True)
    return s1, s2
for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    s1, s2 = step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / a.steps
# threshold: 4 B read + 2 B written per cell
print(json.dumps({"metric": "Mcells/s (GridNet + Threshold)", "value": n * n / el / 1e6, "unit": "Mcells/s", "n_gpus": 1, "ms_per_step": el * 1e3,
                  "config": {"workload": f"{n}x{n} synthetic fractal DEM: D8 directions -> GridNet (plen, tlen, gord), AreaD8 -> Threshold, in HBM"},
                  "gridnet_ms": s1["ms_total"], "gridnet_classes": {k: s1["ms_" + k] for k in ("stencil", "accum", "misc")},
                  "threshold_ms": s2["ms_total"], "threshold_GBps": 6.0 * n * n / (s2["ms_total"] * 1e-3) / 1e9 if s2["ms_total"] > 0 else None}))
