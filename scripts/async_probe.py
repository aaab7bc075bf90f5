"""Bounded probe of the asynchronous tile worklist: each case in its own process under a timeout."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import taudem_amd as T
from oracle import oracle as O
n = int(sys.argv[1])
dem = O.synth_dem((n, n), 7)
with T.Context(0) as ctx:
    fel = ctx.pitremove(dem, -9999.0)
    ok = np.array_equal(fel.view(np.uint32), O.pitremove(dem, -9999.0).view(np.uint32))
    print("n", n, "pitremove equal", ok, flush=True)
''' % ROOT
for n in (64, 128, 300, 1000):
    env = dict(os.environ, TDX_DEBUG_ROUNDS="1", TDX_RELAX_ASYNC="1", TDX_ASYNC_SPIN="200000")
    try:
        r = subprocess.run([sys.executable, "-c", CASE, str(n)], env=env, capture_output=True, text=True, timeout=40)
        print(f"--- n={n} rc={r.returncode}\n{r.stdout[-500:]}\n{r.stderr[-1500:]}", flush=True)
    except subprocess.TimeoutExpired as e:
        print(f"--- n={n} TIMEOUT\n{(e.stdout or b'')[-500:]}\n{(e.stderr or b'')[-1500:]}", flush=True)
        break
