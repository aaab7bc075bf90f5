"""Generates the committed golden vectors by running the REAL reference tools (oracle/_ref, built
from /root/reference by oracle/Makefile) on small seeded inputs.  Run in the build container only:

    python tests/golden/make_golden.py

Each case_<name>.npz holds the inputs and every raster the reference wrote.  The GPU box has no
/root/reference; tests there compare against these files (and against the C restatement, which
tests/test_oracle_vs_golden.py pins to the same files).
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


def make_case(name, ny, nx, seed, dx=30.0, dy=30.0, hole=False, ranks=1, geographic=False, fourway=False, with_mask=False):
    rng = np.random.default_rng(seed)
    dem = O.synth_dem((ny, nx), seed)
    nodata = -9999.0
    if hole:
        yy, xx = np.mgrid[0:ny, 0:nx]
        dem[(yy - ny * 0.4) ** 2 + (xx - nx * 0.6) ** 2 < (min(nx, ny) * 0.12) ** 2] = nodata
        dem[:, :2] = nodata          # nodata border strip on the west side
        dem[ny - 3:, nx // 2:] = nodata
    w = rng.random((ny, nx), dtype=np.float32) * 3.0
    w[rng.random((ny, nx)) < 0.01] = -9999.0 if hole else w[0, 0]   # a few nodata weights in the hole case
    dm = (0.9 + 0.1 * rng.random((ny, nx), dtype=np.float32)).astype(np.float32)
    mask = (rng.random((ny, nx)) < 0.02).astype(np.int16)
    res = {"dem": dem, "w": w, "dm": dm, "dx": dx, "dy": dy, "nodata": nodata, "geographic": geographic, "fourway": fourway}
    with tempfile.TemporaryDirectory() as d:
        if geographic:
            gt = (-111.9, dx, 0.0, 41.9, 0.0, -dy)
        else:
            gt = (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
        f = lambda s: os.path.join(d, s)  # noqa: E731
        T.write_raster(f("dem.tif"), dem, nodata, geotransform=gt, geographic=geographic)
        T.write_raster(f("w.tif"), w, -9999.0, geotransform=gt, geographic=geographic)
        T.write_raster(f("dm.tif"), dm, -9999.0, geotransform=gt, geographic=geographic)
        T.write_raster(f("mask.tif"), mask, -32768, geotransform=gt, geographic=geographic)
        args = ["-z", f("dem.tif"), "-fel", f("fel.tif")] + (["-4way"] if fourway else []) + (["-depmask", f("mask.tif")] if with_mask else [])
        O.run_ref("pitremove", args, ranks)
        if with_mask:
            res["mask"] = mask
        res["fel"], info = T.read_raster(f("fel.tif"))
        res["dxc"], res["dyc"] = info["dxc"], info["dyc"]
        _, err, _ = O.run_ref("d8flowdir", ["-fel", f("fel.tif"), "-p", f("p.tif"), "-sd8", f("sd8.tif")], ranks)
        res["d8_stderr"] = np.array(err)
        res["p"], _ = T.read_raster(f("p.tif"), np.int16)
        res["sd8"], _ = T.read_raster(f("sd8.tif"))
        O.run_ref("aread8", ["-p", f("p.tif"), "-ad8", f("ad8.tif")], ranks)
        res["ad8"], _ = T.read_raster(f("ad8.tif"))
        O.run_ref("aread8", ["-p", f("p.tif"), "-ad8", f("ad8nc.tif"), "-nc"], ranks)
        res["ad8_nc"], _ = T.read_raster(f("ad8nc.tif"))
        O.run_ref("aread8", ["-p", f("p.tif"), "-ad8", f("ad8w.tif"), "-wg", f("w.tif")], ranks)
        res["ad8_w"], _ = T.read_raster(f("ad8w.tif"))
        O.run_ref("aread8", ["-p", f("p.tif"), "-ad8", f("ad8wnc.tif"), "-wg", f("w.tif"), "-nc"], ranks)
        res["ad8_w_nc"], _ = T.read_raster(f("ad8wnc.tif"))
        # outlets: a handful of high-accumulation cells + one on nodata/outside
        a = res["ad8_nc"]
        order = np.argsort(a, axis=None)[::-1]
        pick = order[[3, 40, 200, 900]]
        oy, ox = np.unravel_index(pick, a.shape)
        ox = np.concatenate([ox, [0, nx + 5]]); oy = np.concatenate([oy, [0, 2]])
        xs = gt[0] + (ox + 0.5) * dx
        ys = gt[3] - (oy + 0.5) * dy
        with open(f("outlets.txt"), "w") as fh:
            for x_, y_ in zip(xs, ys):
                fh.write(f"{float(x_)!r} {float(y_)!r}\n")
        res["outlet_xy"] = np.stack([xs, ys])
        O.run_ref("aread8", ["-p", f("p.tif"), "-ad8", f("ad8o.tif"), "-o", f("outlets.txt")], ranks)
        res["ad8_outlets"], _ = T.read_raster(f("ad8o.tif"))
        O.run_ref("aread8", ["-p", f("p.tif"), "-ad8", f("ad8onc.tif"), "-o", f("outlets.txt"), "-nc"], ranks)
        res["ad8_outlets_nc"], _ = T.read_raster(f("ad8onc.tif"))
        # D-infinity
        _, err, _ = O.run_ref("dinfflowdir", ["-fel", f("fel.tif"), "-ang", f("ang.tif"), "-slp", f("slp.tif")], ranks)
        res["ang"], _ = T.read_raster(f("ang.tif"))
        res["slp"], _ = T.read_raster(f("slp.tif"))
        O.run_ref("areadinf", ["-ang", f("ang.tif"), "-sca", f("sca.tif")], ranks)
        res["sca"], _ = T.read_raster(f("sca.tif"))
        O.run_ref("areadinf", ["-ang", f("ang.tif"), "-sca", f("scanc.tif"), "-nc"], ranks)
        res["sca_nc"], _ = T.read_raster(f("scanc.tif"))
        O.run_ref("areadinf", ["-ang", f("ang.tif"), "-sca", f("scaw.tif"), "-wg", f("w.tif"), "-nc"], ranks)
        res["sca_w_nc"], _ = T.read_raster(f("scaw.tif"))
        O.run_ref("areadinf", ["-ang", f("ang.tif"), "-sca", f("scao.tif"), "-o", f("outlets.txt"), "-nc"], ranks)
        res["sca_outlets_nc"], _ = T.read_raster(f("scao.tif"))
        O.run_ref("dinfdecayaccum", ["-ang", f("ang.tif"), "-dm", f("dm.tif"), "-dsca", f("dsca.tif")], ranks)
        res["dsca"], _ = T.read_raster(f("dsca.tif"))
        O.run_ref("dinfdecayaccum", ["-ang", f("ang.tif"), "-dm", f("dm.tif"), "-dsca", f("dscaw.tif"), "-wg", f("w.tif"), "-nc"], ranks)
        res["dsca_w_nc"], _ = T.read_raster(f("dscaw.tif"))
        O.run_ref("dinfdecayaccum", ["-ang", f("ang.tif"), "-dm", f("dm.tif"), "-dsca", f("dscao.tif"), "-o", f("outlets.txt"), "-nc"], ranks)
        res["dsca_outlets_nc"], _ = T.read_raster(f("dscao.tif"))
    np.savez_compressed(os.path.join(OUT, f"case_{name}.npz"), **res)
    flats = [ln for ln in str(res["d8_stderr"]).splitlines() if "flats" in ln]
    print(name, dem.shape, "ranks", ranks, flats)


if __name__ == "__main__":
    O.build()
    make_case("plain", 96, 128, seed=11)
    make_case("holes", 120, 100, seed=12, hole=True, ranks=3)
    make_case("rect_dxdy", 80, 150, seed=13, dx=10.0, dy=25.0, ranks=2)
    make_case("geographic", 90, 110, seed=14, dx=0.0003, dy=0.0003, geographic=True)
    make_case("fourway_mask", 100, 100, seed=15, fourway=True, with_mask=True, hole=True)
