import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taudem_amd as T
ctx = T.Context(0)
for (ny, nx) in [(4096, 4096), (4096, 4160), (4096, 4096 + 8), (16384, 16384), (16384, 16384 + 64)]:
    dem = ctx.synth_dem((ny, nx), seed=5, base_wavelength=2048 if ny <= 4096 else 8192)
    fel = ctx.pitremove(dem, -9999.0)
    torch.cuda.synchronize()
    os.environ.pop("TDX_DEBUG_ROUNDS", None)
    t0 = time.perf_counter()
    fel, st = ctx.pitremove(dem, -9999.0, stats=True)
    torch.cuda.synchronize()
    print("shape", ny, nx, "pitremove ms", round(st["ms_total"], 2), "relax ms", round(st["ms_relax"], 2), "rounds", st["rounds"], "Mcells/s", round(ny * nx / st["ms_total"] / 1e3), flush=True)
    del dem, fel
