"""D-infinity configuration (BASELINE.json configs[2]): DinfFlowDir + AreaDinf on one MI355X, HBM-resident.
usage: python scripts/bench_dinf.py [--size 16384] [--steps 2]"""
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
ang = torch.empty_like(fel); slp = torch.empty_like(fel); sca = torch.empty_like(fel)
def step():
    _, _, s1 = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, out=(ang, slp), stats=True)
    _, s2 = ctx.areadinf(ang, dx=30.0, dy=30.0, out=sca, stats=True)
    return s1, s2
for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    s1, s2 = step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": "Mcells/s (DinfFlowDir->AreaDinf)", "value": n * n / el / 1e6, "unit": "Mcells/s", "n_gpus": 1, "ms_per_step": el * 1e3,
                  "config": {"workload": f"{n}x{n} synthetic fractal DEM (pit-filled), DinfFlowDir + AreaDinf in HBM"},
                  "dinfflowdir_ms": s1["ms_total"], "areadinf_ms": s2["ms_total"],
                  "dinfflowdir_classes": {k: s1["ms_" + k] for k in ("stencil", "bfs", "flatdir", "misc")},
                  "areadinf_classes": {k: s2["ms_" + k] for k in ("stencil", "accum")}, "areadinf_rounds": s2["rounds"]}))
