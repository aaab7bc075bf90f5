"""The reverse D-infinity flow algebra (DinfUpDependence, DinfRevAccum): the C restatement against the rasters
of the real reference tools (CPU), and the HIP path against both (GPU) - bit for bit when fed the reference's own angles."""
import os
import subprocess

import numpy as np
import pytest

import taudem_amd as T
from conftest import bits_equal, describe_diff, golden_cases, load_golden, load_golden_flowalg

CASES = golden_cases()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taudem_amd", "bin")


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name, oracle):
    g, h = load_golden(name), load_golden_flowalg(name)
    dep = oracle.dinfupdependence(g["ang"], h["dg"], dx=g["dxc"], dy=g["dyc"])
    assert bits_equal(dep, h["dep"]), describe_diff(dep, h["dep"], "dep")
    racc, dmax = oracle.dinfrevaccum(g["ang"], h["wg"], dx=g["dxc"], dy=g["dyc"])
    assert bits_equal(racc, h["racc"]), describe_diff(racc, h["racc"], "racc")
    assert bits_equal(dmax, h["dmax"]), describe_diff(dmax, h["dmax"], "dmax")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_reference(name, ctx):
    g, h = load_golden(name), load_golden_flowalg(name)
    ang = np.ascontiguousarray(g["ang"])
    dep = ctx.dinfupdependence(ang, np.ascontiguousarray(h["dg"]), dx=g["dxc"], dy=g["dyc"])
    assert bits_equal(dep, h["dep"]), describe_diff(dep, h["dep"], "dep")
    racc, dmax = ctx.dinfrevaccum(ang, np.ascontiguousarray(h["wg"]), dx=g["dxc"], dy=g["dyc"])
    assert bits_equal(racc, h["racc"]), describe_diff(racc, h["racc"], "racc")
    assert bits_equal(dmax, h["dmax"]), describe_diff(dmax, h["dmax"], "dmax")


@pytest.mark.gpu
def test_gpu_vs_oracle_larger(ctx, oracle):
    rng = np.random.default_rng(12)
    dem = oracle.synth_dem((1100, 900), 31)
    dem[400:440, 200:300] = -9999.0
    ang, _, _ = oracle.dinfflowdir(oracle.pitremove(dem, -9999.0), -3.0e38, 30.0, 20.0)
    dg = (rng.random(ang.shape) < 0.002).astype(np.int32)
    w = (rng.random(ang.shape, dtype=np.float32) * 5).astype(np.float32)
    w[rng.random(ang.shape) < 0.005] = -9999.0
    dep_o = oracle.dinfupdependence(ang, dg, dx=30.0, dy=20.0)
    dep = ctx.dinfupdependence(ang, dg, dx=30.0, dy=20.0)
    assert bits_equal(dep, dep_o), describe_diff(dep, dep_o, "dep")
    racc_o, dmax_o = oracle.dinfrevaccum(ang, w, dx=30.0, dy=20.0)
    racc, dmax = ctx.dinfrevaccum(ang, w, dx=30.0, dy=20.0)
    assert bits_equal(racc, racc_o), describe_diff(racc, racc_o, "racc")
    assert bits_equal(dmax, dmax_o), describe_diff(dmax, dmax_o, "dmax")


@pytest.mark.gpu
@pytest.mark.parametrize("ngpus", [1, 3])
def test_cli(tmp_path, ngpus):
    g, h = load_golden("holes"), load_golden_flowalg("holes")
    ny, nx = g["ang"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    gt = (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("xang.tif"), np.ascontiguousarray(g["ang"]), -3.402823466e38, geotransform=gt)
    T.write_raster(f("xdg.tif"), np.ascontiguousarray(h["dg"]), -1, geotransform=gt)
    T.write_raster(f("xwg.tif"), np.ascontiguousarray(h["wg"]), -9999.0, geotransform=gt)
    N = ["--gpus", str(ngpus)] if ngpus > 1 else []

    def run(tool, *args):
        r = subprocess.run([os.path.join(BIN, tool), *args], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    out = run("dinfupdependence", *N, f("x.tif"))                     # simple usage: xang / xdg -> xdep
    assert "DinfUpDependence version 5.4.0" in out and f"Processors: {ngpus}" in out
    run("dinfrevaccum", "-ang", f("xang.tif"), "-wg", f("xwg.tif"), "-racc", f("racc.tif"), "-dmax", f("dmax.tif"), *N)
    for name, key in (("xdep", "dep"), ("racc", "racc"), ("dmax", "dmax")):
        a, info = T.read_raster(f(name + ".tif"), np.float32)
        assert bits_equal(a, h[key]), describe_diff(a, h[key], key)
    assert T.raster_info(f("xdep.tif"))["nodata"] == -1.0
