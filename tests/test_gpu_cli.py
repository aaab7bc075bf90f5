"""The command-line tools end to end on the GPU box: GeoTIFF in, reference flag surface, GeoTIFF out,
pixels bit-identical to the rasters the real reference tools produced (tests/golden)."""
import os
import subprocess

import numpy as np
import pytest

import taudem_amd as T
from conftest import bits_equal, describe_diff, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taudem_amd", "bin")


def run(tool, *args):
    r = subprocess.run([os.path.join(BIN, tool), *args], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@pytest.mark.parametrize("case", ["plain", "rect_dxdy"])
def test_cli_d8_chain_matches_reference_outputs(tmp_path, case):
    g = load_golden(case)
    ny, nx = g["dem"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("dem.tif"), np.ascontiguousarray(g["dem"]), float(g["nodata"]), geotransform=(1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy))
    out = run("pitremove", "-z", f("dem.tif"), "-fel", f("demfel.tif"))
    assert "PitRemove version 5.4.0" in out and "Compute time" in out          # banner + timing block of src/flood.cpp:61,517-519
    run("d8flowdir", "-fel", f("demfel.tif"), "-p", f("demp.tif"), "-sd8", f("demsd8.tif"))
    run("aread8", f("dem.tif"))                                               # simple usage: names derived with nameadd()
    run("aread8", "-p", f("demp.tif"), "-ad8", f("nc.tif"), "-nc")
    fel, info = T.read_raster(f("demfel.tif"), np.float32)
    p, _ = T.read_raster(f("demp.tif"), np.int16)
    sd8, _ = T.read_raster(f("demsd8.tif"), np.float32)
    ad8, ainfo = T.read_raster(f("demad8.tif"), np.float32)
    nc, _ = T.read_raster(f("nc.tif"), np.float32)
    assert bits_equal(fel, g["fel"]), describe_diff(fel, g["fel"], "fel")
    assert bits_equal(p, g["p"]), describe_diff(p, g["p"], "p")
    assert bits_equal(sd8, g["sd8"]), describe_diff(sd8, g["sd8"], "sd8")
    assert bits_equal(ad8, g["ad8"]), describe_diff(ad8, g["ad8"], "ad8")
    assert bits_equal(nc, g["ad8_nc"]), describe_diff(nc, g["ad8_nc"], "ad8_nc")
    assert ainfo["nodata"] == -1.0 and tuple(ainfo["geotransform"]) == tuple(info["geotransform"])   # src/aread8.cpp:310-311, tiffIO.cpp:344-349


def test_cli_dinf_chain_matches_reference_outputs(tmp_path):
    g = load_golden("plain")
    ny, nx = g["dem"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("fel.tif"), np.ascontiguousarray(g["fel"]), -3.0e38, geotransform=(1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy))
    run("dinfflowdir", "-fel", f("fel.tif"), "-ang", f("ang.tif"), "-slp", f("slp.tif"))
    run("areadinf", "-ang", f("ang.tif"), "-sca", f("sca.tif"))
    ang, _ = T.read_raster(f("ang.tif"), np.float32)
    slp, _ = T.read_raster(f("slp.tif"), np.float32)
    sca, _ = T.read_raster(f("sca.tif"), np.float32)
    assert bits_equal(ang, g["ang"]), describe_diff(ang, g["ang"], "ang")
    assert bits_equal(slp, g["slp"]), describe_diff(slp, g["slp"], "slp")
    ok = np.isclose(sca, g["sca"], rtol=1e-6, atol=0) | (sca == g["sca"])
    assert ok.all(), f"sca: {(~ok).sum()} cells beyond 1e-6 relative"


REF = os.path.join(ROOT, "oracle", "_ref")


def _outlet_file(tmp_path, g, kind):
    from test_boundary_files import write_geojson, write_shp

    pts = [(float(x_), float(y_)) for x_, y_ in zip(*g["outlet_xy"])]
    path = str(tmp_path / ("outlets." + {"shp": "shp", "geojson": "geojson", "txt": "txt"}[kind]))
    if kind == "shp":
        write_shp(path, pts)
    elif kind == "geojson":
        write_geojson(path, pts)
    else:
        with open(path, "w") as fh:
            for x_, y_ in pts:
                fh.write(f"{x_!r} {y_!r}\n")
    return path


@pytest.mark.parametrize("kind", ["shp", "geojson", "txt"])
def test_cli_outlet_files(tmp_path, kind):
    """aread8 -o / areadinf -o / dinfdecayaccum -o / d8flowpathextremeup -o with outlet FILES (shapefile, GeoJSON, text): the
    readers, geoToGlobalXY and the upstream closure end to end, against the rasters the reference wrote for the same points."""
    g = load_golden("plain")
    ny, nx = g["dem"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    gt = (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("p.tif"), np.ascontiguousarray(g["p"]), -32768, geotransform=gt)
    T.write_raster(f("ang.tif"), np.ascontiguousarray(g["ang"]), -3.402823466e38, geotransform=gt)
    T.write_raster(f("dm.tif"), np.ascontiguousarray(g["dm"]), -9999.0, geotransform=gt)
    o = _outlet_file(tmp_path, g, kind)
    run("aread8", "-p", f("p.tif"), "-ad8", f("a.tif"), "-o", o)
    run("aread8", "-p", f("p.tif"), "-ad8", f("anc.tif"), "-o", o, "-nc")
    run("areadinf", "-ang", f("ang.tif"), "-sca", f("s.tif"), "-o", o, "-nc")
    run("dinfdecayaccum", "-ang", f("ang.tif"), "-dm", f("dm.tif"), "-dsca", f("d.tif"), "-o", o, "-nc")
    for name, key in (("a", "ad8_outlets"), ("anc", "ad8_outlets_nc"), ("s", "sca_outlets_nc"), ("d", "dsca_outlets_nc")):
        a, _ = T.read_raster(f(name + ".tif"), np.float32)
        assert bits_equal(a, g[key]), describe_diff(a, g[key], f"{key} with {kind} outlets")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "shim_pitremove")), reason="oracle/_ref/shim_* not built (make -C oracle shimmed)")
def test_reference_mains_on_the_shim(tmp_path):
    """The reference's own UNMODIFIED mains (src/PitRemovemn.cpp, D8FlowDirmn.cpp, aread8mn.cpp, DinfFlowDirmn.cpp, areadinfmn.cpp) linked
    against taudem_amd_shim.cpp + libtaudem_amd.so only (oracle/Makefile `shimmed`): same flags in, same pixels out."""
    g = load_golden("plain")
    ny, nx = g["dem"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("dem.tif"), np.ascontiguousarray(g["dem"]), float(g["nodata"]), geotransform=(1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy))

    def shim(tool, *args):
        r = subprocess.run([os.path.join(REF, "shim_" + tool), *args], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    out = shim("pitremove", f("dem.tif"))                                     # simple usage: the main derives demfel.tif with nameadd()
    assert "PitRemove version 5.4.0" in out
    shim("d8flowdir", "-fel", f("demfel.tif"), "-p", f("demp.tif"), "-sd8", f("demsd8.tif"))
    shim("aread8", "-p", f("demp.tif"), "-ad8", f("demad8.tif"), "-nc")
    shim("dinfflowdir", "-fel", f("demfel.tif"), "-ang", f("demang.tif"), "-slp", f("demslp.tif"))
    shim("areadinf", "-ang", f("demang.tif"), "-sca", f("demsca.tif"))
    for name, key, dt in (("fel", "fel", np.float32), ("p", "p", np.int16), ("sd8", "sd8", np.float32), ("ad8", "ad8_nc", np.float32), ("slp", "slp", np.float32),
                          ("sca", "sca", np.float32)):
        a, _ = T.read_raster(f("dem" + name + ".tif"), dt)
        if name == "sca":   # end to end from OUR angles: the stated gate (1e-6 relative)
            assert (np.isclose(a, g[key], rtol=1e-6, atol=0) | (a == g[key])).all()
        else:
            assert bits_equal(a, g[key]), describe_diff(a, g[key], name)


@pytest.mark.slow
def test_bigtiff_above_4gb_round_trip(tmp_path):
    """A raster above 4 GB goes out as BigTIFF like the reference's (src/tiffIO.cpp:322-330) and comes back pixel for pixel - through our
    reader and, when the helper builds, through libtiff."""
    import shutil

    if shutil.disk_usage(str(tmp_path)).free < 12 * 2 ** 30:
        pytest.skip("needs 12 GB of scratch disk")
    nx, ny = 36000, 30000     # 4.32e9 bytes of float32
    rng = np.random.default_rng(5)
    row = rng.random(nx, dtype=np.float32)
    a = np.empty((ny, nx), np.float32)
    a[:] = row
    a += np.arange(ny, dtype=np.float32)[:, None]
    path = str(tmp_path / "big.tif")
    T.write_raster(path, a, -9999.0, geotransform=(0.0, 1.0, 0.0, float(ny), 0.0, -1.0), lzw=False)
    assert os.path.getsize(path) > 2 ** 32
    assert open(path, "rb").read(4) in (b"II\x2b\x00", b"MM\x00\x2b")      # BigTIFF magic 43
    b, info = T.read_raster(path, np.float32)
    assert info["nx"] == nx and info["ny"] == ny and np.array_equal(a, b)
    del b
    src = os.path.join(ROOT, "tests", "tiffx", "tiffx.c")
    exe = str(tmp_path / "tiffx")
    r = subprocess.run(["gcc", "-O2", "-o", exe, src, "-I/opt/conda/include", "-L/opt/conda/lib", "-ltiff", "-Wl,-rpath,/opt/conda/lib:/usr/lib/x86_64-linux-gnu"],
                       capture_output=True)
    if r.returncode == 0:
        raw = str(tmp_path / "big.raw")
        assert subprocess.run([exe, "dump", path, raw]).returncode == 0
        c = np.fromfile(raw, dtype=np.float32).reshape(ny, nx)
        assert np.array_equal(a, c)
