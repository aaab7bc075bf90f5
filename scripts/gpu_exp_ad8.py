"""GPU experiment: which inter-workgroup visibility recipe is correct/fast for the AreaD8 walk."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taudem_amd as T
from oracle import oracle as O

ctx = T.Context(0)
cases = [((1000, 777), 7), ((2048, 2048), 3)]
for shape, seed in cases:
    dem = O.synth_dem(shape, seed)
    fel = ctx.pitremove(dem, -9999.0)
    p, _ = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, want_slope=False)
    p_o, _, _ = O.d8flowdir(O.pitremove(dem, -9999.0), -3.0e38, 30.0, 30.0)
    print(shape, "p equal", np.array_equal(p, p_o), flush=True)
    a_o = {cc: O.aread8(p_o, -32768, contcheck=cc) for cc in (True, False)}
    for mode in (0, 1, 2):
        os.environ["TDX_AD8_MODE"] = str(mode)
        bad = 0; worst = 0
        t0 = time.time()
        for rep in range(6):
            for cc in (True, False):
                a = ctx.aread8(p, -32768, contcheck=cc)
                nd = int((a.view(np.uint32) != a_o[cc].view(np.uint32)).sum())
                bad += nd > 0; worst = max(worst, nd)
        print(f"  mode {mode}: failing runs {bad}/12 worst {worst} cells, {time.time()-t0:.2f}s", flush=True)

import torch
for n in (4096, 16384):
    dem = ctx.synth_dem(n, seed=1234)
    fel = ctx.pitremove(dem, -9999.0)
    p, _ = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, want_slope=False)
    ref = None
    for mode in (0, 1, 2):
        os.environ["TDX_AD8_MODE"] = str(mode)
        ts = []
        for rep in range(3):
            a, st = ctx.aread8(p, -32768, stats=True)
            ts.append(st["ms_accum"])
            if mode == 1 and ref is None:
                ref = a.clone()
        print(f"n={n} mode {mode}: accum ms {ts}", flush=True)
    for mode in (0, 2):
        os.environ["TDX_AD8_MODE"] = str(mode)
        a = ctx.aread8(p, -32768)
        print(f"n={n} mode {mode} vs mode 1: differing cells {int((a != ref).sum())}", flush=True)
