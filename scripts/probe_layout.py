import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taudem_amd as T
ctx = T.Context(0)
for (ny, nx) in [(65536, 256), (16384, 1024), (4096, 4096), (1024, 16384)]:
    dem = ctx.synth_dem((ny, nx), seed=5, base_wavelength=1024)
    print("shape", ny, nx, flush=True)
    sys.stderr.flush()
    fel = ctx.pitremove(dem, -9999.0)
    torch.cuda.synchronize()
