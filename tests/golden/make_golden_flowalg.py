"""Golden vectors for the D-infinity flow algebra of SURVEY.md 8f rank 4: runs the REAL reference tools (oracle/_ref/dinfupdependence,
dinfrevaccum, dinfconclimaccum, dinftranslimaccum, built from /root/reference by oracle/Makefile; 1-3 MPI ranks) on the D-infinity
angles of the committed cases.  Build container only:

    python tests/golden/make_golden_flowalg.py

case_<name>_flowalg.npz holds the extra inputs (disturbance grid dg, weight grid wg with a few nodata cells, decay multiplier dm2,
specific discharge q, indicator grid dgs, supply tsup, capacity tc, concentration cs, outlet cells) and every raster the reference wrote.
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
    # concentration / transport limited accumulation: inputs with a few nodata and non-positive cells
    dm2 = (0.8 + 0.2 * rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    dm2[rng.random((ny, nx)) < 0.003] = -9999.0
    q = (0.5 + rng.random((ny, nx), dtype=np.float32) * 4.0).astype(np.float32)
    q[rng.random((ny, nx)) < 0.004] = 0.0                                    # q <= 0: the cell gets no value
    q[rng.random((ny, nx)) < 0.002] = -9999.0
    dgs = (rng.random((ny, nx)) < 0.02).astype(np.int16)                     # 2 % source cells
    tsup = (rng.random((ny, nx), dtype=np.float32) * 2.0).astype(np.float32)
    tsup[rng.random((ny, nx)) < 0.003] = -9999.0
    tc = (rng.random((ny, nx), dtype=np.float32) * 6.0).astype(np.float32)
    tc[rng.random((ny, nx)) < 0.003] = -9999.0
    cs = (rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    cs[rng.random((ny, nx)) < 0.003] = -9999.0
    order = np.argsort(g["sca_nc"], axis=None)[::-1][[5, 60, 300]]            # outlet cells (x, y) with some area upstream
    oy_, ox_ = np.unravel_index(order, ang.shape)
    oxy = np.stack([ox_, oy_], axis=1).astype(np.int32)
    res = {"dg": dg, "wg": wg, "dm2": dm2, "q": q, "dgs": dgs, "tsup": tsup, "tc": tc, "cs": cs, "outlets_xy": oxy}
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
        for nm, arr in (("dm2", dm2), ("q", q), ("tsup", tsup), ("tc", tc), ("cs", cs)):
            T.write_raster(f(nm + ".tif"), arr, -9999.0, geotransform=gt, geographic=geographic)
        T.write_raster(f("dgs.tif"), dgs, -1, geotransform=gt, geographic=geographic)
        with open(f("outlets.txt"), "w") as fo:   # cell centres in map coordinates
            for x, y in oxy:
                fo.write(f"{float(gt[0] + (int(x) + 0.5) * gt[1])!r} {float(gt[3] + (int(y) + 0.5) * gt[5])!r}\n")
        base = ["-ang", f("ang.tif"), "-dg", f("dgs.tif"), "-dm", f("dm2.tif"), "-q", f("q.tif")]
        O.run_ref("dinfconclimaccum", base + ["-ctpt", f("ctpt.tif"), "-csol", "2.5"], ranks)
        res["ctpt"], _ = T.read_raster(f("ctpt.tif"))
        O.run_ref("dinfconclimaccum", base + ["-ctpt", f("ctpt_nc.tif"), "-nc"], ranks)
        res["ctpt_nc"], _ = T.read_raster(f("ctpt_nc.tif"))
        O.run_ref("dinfconclimaccum", base + ["-ctpt", f("ctpt_o.tif"), "-nc", "-o", f("outlets.txt")], ranks)
        res["ctpt_outlets_nc"], _ = T.read_raster(f("ctpt_o.tif"))
        tb = ["-ang", f("ang.tif"), "-tsup", f("tsup.tif"), "-tc", f("tc.tif")]
        O.run_ref("dinftranslimaccum", tb + ["-tla", f("tla.tif"), "-tdep", f("tdep.tif")], ranks)
        res["tla"], _ = T.read_raster(f("tla.tif"))
        res["tdep"], _ = T.read_raster(f("tdep.tif"))
        O.run_ref("dinftranslimaccum", tb + ["-tla", f("tla2.tif"), "-tdep", f("tdep2.tif"), "-cs", f("cs.tif"), "-ctpt", f("tctpt.tif"), "-nc"], ranks)
        res["tla_cs_nc"], _ = T.read_raster(f("tla2.tif"))
        res["tdep_cs_nc"], _ = T.read_raster(f("tdep2.tif"))
        res["tctpt_cs_nc"], _ = T.read_raster(f("tctpt.tif"))
        O.run_ref("dinftranslimaccum", tb + ["-tla", f("tla3.tif"), "-tdep", f("tdep3.tif"), "-cs", f("cs.tif"), "-ctpt", f("tctpt3.tif"), "-nc", "-o", f("outlets.txt")], ranks)
        res["tla_cs_outlets_nc"], _ = T.read_raster(f("tla3.tif"))
        res["tdep_cs_outlets_nc"], _ = T.read_raster(f("tdep3.tif"))
        res["tctpt_cs_outlets_nc"], _ = T.read_raster(f("tctpt3.tif"))
    np.savez_compressed(os.path.join(OUT, f"case_{name}_flowalg.npz"), **res)
    print(name, ang.shape, "ranks", ranks, "dep max", float(res["dep"].max()), "racc max", float(res["racc"][res["racc"] > -1e30].max()))


if __name__ == "__main__":
    O.build()
    make("plain")
    make("holes", ranks=3)
    make("rect_dxdy", ranks=2)
    make("geographic")
    make("fourway_mask")
