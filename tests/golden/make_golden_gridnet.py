"""Golden vectors for GridNet and Threshold: runs the REAL reference tools (oracle/_ref/gridnet, oracle/_ref/threshold,
built from /root/reference by oracle/Makefile) on the D8 rasters of the committed cases.  Build container only:

    python tests/golden/make_golden_gridnet.py

case_<name>_gridnet.npz holds the extra inputs (mask rasters, thresholds) and every raster the reference wrote.
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
    p, ad8 = g["p"], g["ad8_nc"]
    ny, nx = p.shape
    dx, dy, geographic = float(g["dx"]), float(g["dy"]), bool(g["geographic"])
    gt = (-111.9, dx, 0.0, 41.9, 0.0, -dy) if geographic else (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
    rng = np.random.default_rng(100 + ny)
    mask_i32 = np.where(ad8 < 0, -7, ad8).astype(np.int32)            # GridNet mask: evaluate cells draining >= gn_thresh cells
    gn_thresh = 4
    ssa_thresh = 25.0
    tmask = (rng.random((ny, nx), dtype=np.float32) - 0.2).astype(np.float32)   # Threshold mask: >= 0 passes
    res = {"mask_i32": mask_i32, "gn_thresh": gn_thresh, "ssa_thresh": ssa_thresh, "tmask": tmask}
    with tempfile.TemporaryDirectory() as d:
        f = lambda s: os.path.join(d, s)  # noqa: E731
        T.write_raster(f("p.tif"), p, -32768, geotransform=gt, geographic=geographic)
        T.write_raster(f("ad8.tif"), ad8, -1.0, geotransform=gt, geographic=geographic)
        T.write_raster(f("mask.tif"), mask_i32, -2147483647, geotransform=gt, geographic=geographic)
        T.write_raster(f("tmask.tif"), tmask, -9999.0, geotransform=gt, geographic=geographic)
        O.run_ref("gridnet", ["-p", f("p.tif"), "-plen", f("plen.tif"), "-tlen", f("tlen.tif"), "-gord", f("gord.tif")], ranks)
        res["plen"], _ = T.read_raster(f("plen.tif"))
        res["tlen"], _ = T.read_raster(f("tlen.tif"))
        res["gord"], _ = T.read_raster(f("gord.tif"), np.int16)
        O.run_ref("gridnet", ["-p", f("p.tif"), "-plen", f("plenm.tif"), "-tlen", f("tlenm.tif"), "-gord", f("gordm.tif"), "-mask", f("mask.tif"),
                              "-thresh", str(gn_thresh)], ranks)
        res["plen_m"], _ = T.read_raster(f("plenm.tif"))
        res["tlen_m"], _ = T.read_raster(f("tlenm.tif"))
        res["gord_m"], _ = T.read_raster(f("gordm.tif"), np.int16)
        O.run_ref("threshold", ["-ssa", f("ad8.tif"), "-src", f("src.tif"), "-thresh", str(ssa_thresh)], ranks)
        res["src"], _ = T.read_raster(f("src.tif"), np.int16)
        O.run_ref("threshold", ["-ssa", f("ad8.tif"), "-src", f("srcm.tif"), "-thresh", str(ssa_thresh), "-mask", f("tmask.tif")], ranks)
        res["src_m"], _ = T.read_raster(f("srcm.tif"), np.int16)
        # D8FlowPathExtremeUp: upstream maximum / minimum of the D8 slope grid; with -nc; restricted to outlets
        T.write_raster(f("sd8.tif"), g["sd8"], -1.0, geotransform=gt, geographic=geographic)
        O.run_ref("d8flowpathextremeup", ["-p", f("p.tif"), "-sa", f("sd8.tif"), "-ssa", f("xmax.tif")], ranks)
        res["xup_max"], _ = T.read_raster(f("xmax.tif"))
        O.run_ref("d8flowpathextremeup", ["-p", f("p.tif"), "-sa", f("sd8.tif"), "-ssa", f("xmin.tif"), "-min", "-nc"], ranks)
        res["xup_min_nc"], _ = T.read_raster(f("xmin.tif"))
        xs, ys = g["outlet_xy"]
        with open(f("outlets.txt"), "w") as fh:
            for x_, y_ in zip(xs, ys):
                fh.write(f"{float(x_)!r} {float(y_)!r}\n")
        O.run_ref("d8flowpathextremeup", ["-p", f("p.tif"), "-sa", f("sd8.tif"), "-ssa", f("xo.tif"), "-o", f("outlets.txt"), "-nc"], ranks)
        res["xup_max_outlets_nc"], _ = T.read_raster(f("xo.tif"))
        # GridNet restricted to the outlets' catchments (the last two points of the list lie on nodata / outside the raster; an outlet on a
        # nodata cell makes the reference index its neighbour tables out of range, so only the in-catchment points are used here)
        with open(f("outlets_in.txt"), "w") as fh:
            for x_, y_ in list(zip(xs, ys))[:4]:
                fh.write(f"{float(x_)!r} {float(y_)!r}\n")
        O.run_ref("gridnet", ["-p", f("p.tif"), "-plen", f("pleno.tif"), "-tlen", f("tleno.tif"), "-gord", f("gordo.tif"), "-o", f("outlets_in.txt")], ranks)
        res["plen_o"], _ = T.read_raster(f("pleno.tif"))
        res["tlen_o"], _ = T.read_raster(f("tleno.tif"))
        res["gord_o"], _ = T.read_raster(f("gordo.tif"), np.int16)
    np.savez_compressed(os.path.join(OUT, f"case_{name}_gridnet.npz"), **res)
    print(name, p.shape, "ranks", ranks, "max gord", int(res["gord"].max()), "max plen", float(res["plen"].max()), "src cells", int((res["src"] == 1).sum()))


if __name__ == "__main__":
    O.build()
    make("plain")
    make("holes", ranks=3)
    make("rect_dxdy", ranks=2)
    make("geographic")
    make("fourway_mask")
