"""Bounded probe (run under `timeout`): one PitRemove + D8FlowDir of a small synthetic DEM against the oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
import taudem_amd
from oracle import oracle as O

O.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = taudem_amd.Context(0)
dem = O.synth_dem((n, n + 57), 7)
fel = ctx.pitremove(dem, -9999.0)
fel_o = O.pitremove(dem, -9999.0)
print("pitremove", "OK" if np.array_equal(fel.view(np.uint32), fel_o.view(np.uint32)) else "MISMATCH", flush=True)
p, sd8 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0)
p_o, sd8_o, _ = O.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
print("d8flowdir", "OK" if np.array_equal(p, p_o) else "MISMATCH", flush=True)
