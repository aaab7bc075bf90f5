"""Golden vectors for the reverse D-infinity flow algebra: runs the REAL reference tools (oracle/_ref/dinfupdependence,
oracle/_ref/dinfrevaccum, built from /root/reference by oracle/Makefile; 1-3 MPI ranks) on the D-infinity angles of the
committed cases.  Build container only:

    python tests/golden/make_golden_flowalg.py

case_<name>_flowalg.npz holds the extra inputs (disturbance grid dg, weight grid wg with a few nodata cells) and every raster the
reference wrote.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import taudem_amd as T  # noqa: E402  (raster file IO only)
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make(name, ranks=1):
    g = np.load(os.path.join(OUT, f"case_{name}.npz"))
    ang = g["ang"]
    ny, nx = ang.shape
    dx, dy, geographic = float(g["dx"]), float(g["dy"]), bool(g["geographic"])
    gt = (-111.9, dx, 0.0, 41.9, 0.0, -dy) if geographic else (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
    rng = np.random.default_rng(300 + nx)
    dg = (rng.random((ny, nx)) < 0.01).astype(np.int32)                      # 1 % disturbance cells
    dg[ny // 3:ny // 3 + 4, nx // 2:nx // 2 + 6] = 2
    wg = (rng.random((ny, nx), dtype=np.float32) * 10.0).astype(np.float32)
    wg[rng.random((ny, nx)) < 0.01] = -9999.0                                # nodata weights
    res = {"dg": dg, "wg": wg}
    with tempfile.TemporaryDirectory() as d:
        f = lambda s: os.path.join(d, s)  # noqa: E731
        T.write_raster(f("ang.tif"), ang, -3.402823466e38, geotransform=gt, geographic=geographic)
        T.write_raster(f("dg.tif"), dg, -1, geotransform=gt, geographic=geographic)
        T.write_raster(f("wg.tif"), wg, -9999.0, geotransform=gt, geographic=geographic)
        O.run_ref("dinfupdependence", ["-ang", f("ang.tif"), "-dg", f("dg.tif"), "-dep", f("dep.tif")], ranks)
        res["dep"], _ = T.read_raster(f("dep.tif"))
        O.run_ref("dinfrevaccum", ["-ang", f("ang.tif"), "-wg", f("wg.tif"), "-racc", f("racc.tif"), "-dmax", f("dmax.tif")], ranks)
        res["racc"], _ = T.read_raster(f("racc.tif"))
        res["dmax"], _ = T.read_raster(f("dmax.tif"))
    np.savez_compressed(os.path.join(OUT, f"case_{name}_flowalg.npz"), **res)
    print(name, ang.shape, "ranks", ranks, "dep max", float(res["dep"].max()), "racc max", float(res["racc"][res["racc"] > -1e30].max()))


if __name__ == "__main__":
    O.build()
    make("plain")
    make("holes", ranks=3)
    make("rect_dxdy", ranks=2)
    make("geographic")
    make("fourway_mask")
