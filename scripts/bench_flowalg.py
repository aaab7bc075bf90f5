"""The accumulation tools beside the headline path (SURVEY.md 8f ranks 2 and 4, config 5's kernel) on one MI355X, HBM-resident inputs:
weighted AreaD8, D8FlowPathExtremeUp, GridNet, DinfDecayAccum with weights and outlets, DinfUpDependence, DinfRevAccum,
DinfConcLimAccum, DinfTransLimAccum.  One JSON line with the ms of each (library-side HIP-event time of the call).
usage: python scripts/bench_flowalg.py [--size 16384]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import taudem_amd as T

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--only", default="", help="comma-separated tool names (default: all)")
ap.add_argument("--digest", action="store_true", help="also CRC-32 of every result raster (schedule experiments: the bits must not move)")
a = ap.parse_args()
n = a.size
ctx = T.Context(0)
dev = "cuda:0"
dem = ctx.synth_dem(n, seed=1234)
fel = ctx.pitremove(dem, -9999.0)
del dem
p, _ = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, want_slope=False)
ang, slp = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
del fel, slp
g = torch.Generator(device=dev).manual_seed(7)
w = torch.rand((n, n), device=dev, dtype=torch.float32, generator=g)
w2 = 0.9 + 0.1 * torch.rand((n, n), device=dev, dtype=torch.float32, generator=g)
dg16 = (torch.rand((n, n), device=dev, generator=g) < 0.01).to(torch.int16)
dg32 = dg16.to(torch.int32)
outl = (np.array([n // 2, n // 3], dtype=np.int32), np.array([n - 5, n // 2], dtype=np.int32))
res = {}
only = set(x for x in a.only.split(',') if x)
def timed(name, fn):
    if only and name not in only: return
    fn()                       # warm-up (scratch allocation)
    torch.cuda.synchronize()
    out = fn()
    res[name] = out[-1]["ms_total"]
    if a.digest:
        import zlib
        res[name + "_crc"] = [zlib.crc32(o.cpu().numpy().tobytes()) for o in out[:-1] if torch.is_tensor(o)]
timed("aread8_weighted", lambda: ctx.aread8(p, weights=w, stats=True))
timed("d8flowpathextremeup", lambda: ctx.d8flowpathextremeup(p, w, stats=True))
timed("gridnet", lambda: ctx.gridnet(p, -32768, 30.0, 30.0, stats=True))
timed("dinfdecayaccum_w_outlets", lambda: ctx.dinfdecayaccum(ang, w2, dx=30.0, dy=30.0, weights=w, outlets=outl, stats=True))
timed("dinfdecayaccum", lambda: ctx.dinfdecayaccum(ang, w2, dx=30.0, dy=30.0, stats=True))
timed("dinfupdependence", lambda: ctx.dinfupdependence(ang, dg32, dx=30.0, dy=30.0, stats=True))
timed("dinfrevaccum", lambda: ctx.dinfrevaccum(ang, w, dx=30.0, dy=30.0, stats=True))
timed("dinfconclimaccum", lambda: ctx.dinfconclimaccum(ang, w2, dg16, w + 0.5, dx=30.0, dy=30.0, stats=True))
timed("dinftranslimaccum_cs", lambda: ctx.dinftranslimaccum(ang, w, 50.0 * w2, cs=w2, dx=30.0, dy=30.0, stats=True))
print(json.dumps({"metric": "ms per call", "size": n, "n_gpus": 1, "ms": res,
                  "config": {"workload": f"{n}x{n} synthetic fractal DEM (pit-filled): D8 / D-infinity directions from the library, random weight / multiplier / indicator grids in HBM"}}))
