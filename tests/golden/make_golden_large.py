"""Config-scale goldens: runs the REAL reference tools (oracle/_ref, 8 MPI ranks) on the bench generator's DEM
(tdx_synth_dem / orc_synth_dem, seed 1234 - bit-identical on host and device) at 2048^2 and 4096^2 and commits
DIGESTS of every raster the reference wrote, not the rasters:

    python tests/golden/make_golden_large.py [sizes...]          # build container only (needs /root/reference)

tests/golden/large_digests.json then holds, per size and raster: SHA-256 of the raw little-endian bytes, CRC-32 of
every 64-row band (so a failing GPU test can say WHERE it differs) and a few summary numbers, plus the reference's own
timings (the tools' "Compute time" lines) as the CPU baseline of that size.  The GPU tests regenerate the DEM on the
device, run the HIP path and compare digests (tests/test_gpu_large_golden.py).
"""
import hashlib
import json
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import taudem_amd as T  # noqa: E402  (raster file IO only)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "large_digests.json")
BAND = 64


def digest(a):
    """SHA-256 of the raw bytes + CRC-32 per 64-row band + summary numbers of a raster."""
    a = np.ascontiguousarray(a)
    raw = a.view(np.uint8).reshape(a.shape[0], -1)
    d = {"dtype": str(a.dtype), "shape": list(a.shape), "sha256": hashlib.sha256(raw.tobytes()).hexdigest(),
         "band_rows": BAND, "band_crc32": [zlib.crc32(raw[r:r + BAND].tobytes()) for r in range(0, a.shape[0], BAND)]}
    if a.dtype == np.float32:
        fin = a[np.abs(a) < 1e37]
        d["sum_f64"] = float(fin.astype(np.float64).sum())
        d["max"] = float(fin.max()) if fin.size else None
        d["n_finite"] = int(fin.size)
    else:
        d["hist"] = {str(int(v)): int(c) for v, c in zip(*np.unique(a, return_counts=True))}
    return d


def make(n, seed=1234, ranks=8, dinf=True):
    dem = O.synth_dem(n, seed)
    res = {"n": n, "seed": seed, "ranks": ranks, "dx": 30.0, "dy": 30.0, "nodata": -9999.0, "rasters": {}, "ref_seconds": {}}
    res["rasters"]["dem"] = digest(dem)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TDX_TMP", None)) as d:
        f = lambda s: os.path.join(d, s)  # noqa: E731
        gt = (0.0, 30.0, 0.0, 30.0 * n, 0.0, -30.0)
        T.write_raster(f("dem.tif"), dem, -9999.0, geotransform=gt)
        del dem

        def run(tool, args, outs):
            t0 = time.time()
            _, err, tm = O.run_ref(tool, args, ranks, timeout=48 * 3600)
            res["ref_seconds"][tool] = dict(tm, wall=time.time() - t0)
            for key, (name, dt) in outs.items():
                a, _ = T.read_raster(f(name), dt)
                res["rasters"][key] = digest(a)
            flats = [ln.strip() for ln in err.splitlines() if "flats" in ln.lower()]
            if flats:
                res["ref_seconds"][tool]["stderr_flats"] = flats
            print(n, tool, res["ref_seconds"][tool], flush=True)

        run("pitremove", ["-z", f("dem.tif"), "-fel", f("fel.tif")], {"fel": ("fel.tif", np.float32)})
        run("d8flowdir", ["-fel", f("fel.tif"), "-p", f("p.tif"), "-sd8", f("sd8.tif")], {"p": ("p.tif", np.int16), "sd8": ("sd8.tif", np.float32)})
        run("aread8", ["-p", f("p.tif"), "-ad8", f("ad8.tif")], {"ad8": ("ad8.tif", np.float32)})
        if dinf:
            run("dinfflowdir", ["-fel", f("fel.tif"), "-ang", f("ang.tif"), "-slp", f("slp.tif")], {"ang": ("ang.tif", np.float32), "slp": ("slp.tif", np.float32)})
            run("areadinf", ["-ang", f("ang.tif"), "-sca", f("sca.tif")], {"sca": ("sca.tif", np.float32)})
    return res


if __name__ == "__main__":
    O.build()
    sizes = [int(s) for s in sys.argv[1:]] or [2048, 4096]
    allres = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for n in sizes:
        allres[str(n)] = make(n)
        json.dump(allres, open(OUT, "w"), indent=1)
        print("written", n, flush=True)
