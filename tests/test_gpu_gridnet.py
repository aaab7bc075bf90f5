"""GPU parity of GridNet and Threshold through the C ABI: bit-exact against the rasters written by the
real reference tools (tests/golden/case_*_gridnet.npz) and against the pinned CPU restatement on seeded synthetic DEMs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import bits_equal, describe_diff, golden_cases, load_golden, load_golden_gridnet, outlets_to_indices

pytestmark = pytest.mark.gpu
CASES = golden_cases()
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taudem_amd", "bin")


@pytest.mark.parametrize("name", CASES)
def test_golden_gridnet(name, ctx):
    g, h = load_golden(name), load_golden_gridnet(name)
    p = np.ascontiguousarray(g["p"])
    plen, tlen, gord = ctx.gridnet(p, -32768, g["dxc"], g["dyc"])
    assert bits_equal(gord, h["gord"]), describe_diff(gord, h["gord"], "gord")
    assert bits_equal(plen, h["plen"]), describe_diff(plen, h["plen"], "plen")
    assert bits_equal(tlen, h["tlen"]), describe_diff(tlen, h["tlen"], "tlen")
    plen, tlen, gord = ctx.gridnet(p, -32768, g["dxc"], g["dyc"], mask=np.ascontiguousarray(h["mask_i32"]), thresh=int(h["gn_thresh"]))
    assert bits_equal(gord, h["gord_m"]), describe_diff(gord, h["gord_m"], "gord (mask)")
    assert bits_equal(plen, h["plen_m"]), describe_diff(plen, h["plen_m"], "plen (mask)")
    assert bits_equal(tlen, h["tlen_m"]), describe_diff(tlen, h["tlen_m"], "tlen (mask)")


@pytest.mark.parametrize("name", CASES)
def test_golden_threshold(name, ctx):
    g, h = load_golden(name), load_golden_gridnet(name)
    ssa = np.ascontiguousarray(g["ad8_nc"])
    src = ctx.threshold(ssa, float(h["ssa_thresh"]), -1.0)
    assert bits_equal(src, h["src"]), describe_diff(src, h["src"], "src")
    src = ctx.threshold(ssa, float(h["ssa_thresh"]), -1.0, mask=np.ascontiguousarray(h["tmask"]))
    assert bits_equal(src, h["src_m"]), describe_diff(src, h["src_m"], "src (mask)")


@pytest.mark.parametrize("name", CASES)
def test_golden_d8flowpathextremeup(name, ctx):
    g, h = load_golden(name), load_golden_gridnet(name)
    p, sa = np.ascontiguousarray(g["p"]), np.ascontiguousarray(g["sd8"])
    a = ctx.d8flowpathextremeup(p, sa, -32768, usemax=True, contcheck=True)
    assert bits_equal(a, h["xup_max"]), describe_diff(a, h["xup_max"], "xup_max")
    a = ctx.d8flowpathextremeup(p, sa, -32768, usemax=False, contcheck=False)
    assert bits_equal(a, h["xup_min_nc"]), describe_diff(a, h["xup_min_nc"], "xup_min_nc")
    a = ctx.d8flowpathextremeup(p, sa, -32768, usemax=True, contcheck=False, outlets=outlets_to_indices(g))
    assert bits_equal(a, h["xup_max_outlets_nc"]), describe_diff(a, h["xup_max_outlets_nc"], "xup_max_outlets_nc")


def test_d8flowpathextremeup_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(4)
    dem = oracle.synth_dem((900, 1100), 21)
    dem[100:140, 300:360] = -9999.0
    p, sd8, _ = oracle.d8flowdir(oracle.pitremove(dem, -9999.0), -3.0e38, 30.0, 30.0)
    sa = (rng.random(p.shape, dtype=np.float32) * 100.0 - 20.0).astype(np.float32)
    for usemax in (True, False):
        for cc in (True, False):
            a_o = oracle.d8flowpathextremeup(p, sa, -32768, usemax=usemax, contcheck=cc)
            a = ctx.d8flowpathextremeup(p, sa, -32768, usemax=usemax, contcheck=cc)
            assert bits_equal(a, a_o), describe_diff(a, a_o, f"usemax={usemax} contcheck={cc}")


@pytest.mark.parametrize("shape,seed", [((1, 1), 2), ((3, 3), 3), ((5, 200), 4), ((257, 301), 5), ((1000, 777), 7)])
def test_gridnet_vs_oracle(shape, seed, ctx, oracle):
    dem = oracle.synth_dem(shape, seed)
    fel = oracle.pitremove(dem, -9999.0)
    p, _, _ = oracle.d8flowdir(fel, -3.0e38, 10.0, 25.0)
    a = oracle.aread8(p, -32768, contcheck=False)
    for mask, thresh in ((None, 0), (np.where(a < 0, -3, a).astype(np.int32), 3)):
        pl_o, tl_o, go_o = oracle.gridnet(p, -32768, 10.0, 25.0, mask=mask, thresh=thresh)
        pl, tl, go = ctx.gridnet(p, -32768, 10.0, 25.0, mask=mask, thresh=thresh)
        assert bits_equal(go, go_o), describe_diff(go, go_o, f"gord thresh={thresh}")
        assert bits_equal(pl, pl_o), describe_diff(pl, pl_o, f"plen thresh={thresh}")
        assert bits_equal(tl, tl_o), describe_diff(tl, tl_o, f"tlen thresh={thresh}")


def test_gridnet_quirks(ctx, oracle):
    """p == 0 cells (counted as contributors of their south-east neighbour, never drained: src/gridnet.cpp:250-256,429-432),
    nodata holes, a two-cell cycle."""
    rng = np.random.default_rng(9)
    dem = oracle.synth_dem((300, 300), 12)
    p, _, _ = oracle.d8flowdir(oracle.pitremove(dem, -9999.0), -3.0e38, 30.0, 30.0)
    p = p.copy()
    idx = rng.integers(5, 295, size=(40, 2))
    for y, x in idx[:20]:
        p[y, x] = 0
    for y, x in idx[20:30]:
        p[y, x] = -32768
    p[150, 150], p[150, 151] = 1, 5   # a cycle
    pl_o, tl_o, go_o = oracle.gridnet(p, -32768, 30.0, 30.0)
    pl, tl, go = ctx.gridnet(p, -32768, 30.0, 30.0)
    assert bits_equal(go, go_o), describe_diff(go, go_o, "gord")
    assert bits_equal(pl, pl_o), describe_diff(pl, pl_o, "plen")
    assert bits_equal(tl, tl_o), describe_diff(tl, tl_o, "tlen")


def test_cli_gridnet_threshold(tmp_path, ctx):
    """The command-line tools on GeoTIFFs: same pixels as the reference tools wrote for the same inputs."""
    import taudem_amd as T

    g, h = load_golden("rect_dxdy"), load_golden_gridnet("rect_dxdy")
    ny, nx = g["p"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    gt = (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("p.tif"), np.ascontiguousarray(g["p"]), -32768, geotransform=gt)
    T.write_raster(f("ad8.tif"), np.ascontiguousarray(g["ad8_nc"]), -1.0, geotransform=gt)
    T.write_raster(f("mask.tif"), np.ascontiguousarray(h["mask_i32"]), -2147483647, geotransform=gt)

    def run(tool, *args):
        r = subprocess.run([os.path.join(BIN, tool), *args], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    out = run("gridnet", "-p", f("p.tif"), "-plen", f("plen.tif"), "-tlen", f("tlen.tif"), "-gord", f("gord.tif"), "-mask", f("mask.tif"), "-thresh",
              str(int(h["gn_thresh"])))
    assert "GridNet version 5.4.0" in out
    for key, name, dt in (("plen_m", "plen.tif", np.float32), ("tlen_m", "tlen.tif", np.float32), ("gord_m", "gord.tif", np.int16)):
        a, _ = T.read_raster(f(name), dt)
        assert bits_equal(a, h[key]), describe_diff(a, h[key], key)
    T.write_raster(f("sd8.tif"), np.ascontiguousarray(g["sd8"]), -1.0, geotransform=gt)
    out = run("d8flowpathextremeup", "-p", f("p.tif"), "-sa", f("sd8.tif"), "-ssa", f("xmin.tif"), "-min", "-nc")
    assert "D8FlowPathExtremeUp version 5.4.0" in out
    a, _ = T.read_raster(f("xmin.tif"), np.float32)
    assert bits_equal(a, h["xup_min_nc"]), describe_diff(a, h["xup_min_nc"], "xup_min_nc")
    out = run("threshold", "-ssa", f("ad8.tif"), "-src", f("src.tif"), "-thresh", str(float(h["ssa_thresh"])))
    assert "Threshold version 5.4.0" in out
    a, _ = T.read_raster(f("src.tif"), np.int16)
    assert bits_equal(a, h["src"]), describe_diff(a, h["src"], "src")


@pytest.mark.parametrize("name", CASES)
def test_golden_gridnet_outlets(name, ctx):
    """gridnet -o (src/gridnet.cpp:269-369): only the outlets' upstream closure is evaluated, every other cell with a direction gets order 0."""
    g, h = load_golden(name), load_golden_gridnet(name)
    if "gord_o" not in h:
        pytest.skip("no -o rasters in this golden file")
    p = np.ascontiguousarray(g["p"])
    plen, tlen, gord = ctx.gridnet(p, -32768, g["dxc"], g["dyc"], outlets=outlets_to_indices(g))
    assert bits_equal(gord, h["gord_o"]), describe_diff(gord, h["gord_o"], "gord -o")
    assert bits_equal(plen, h["plen_o"]), describe_diff(plen, h["plen_o"], "plen -o")
    assert bits_equal(tlen, h["tlen_o"]), describe_diff(tlen, h["tlen_o"], "tlen -o")


def test_gridnet_outlets_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(8)
    dem = oracle.synth_dem((700, 900), 17)
    dem[300:330, 100:180] = -9999.0
    p, _, _ = oracle.d8flowdir(oracle.pitremove(dem, -9999.0), -3.0e38, 30.0, 30.0)
    ox = rng.integers(1, 899, size=12).astype(np.int32); oy = rng.integers(1, 699, size=12).astype(np.int32)
    ox[0], oy[0] = 120, 310          # an outlet on a nodata cell: ignored
    mask = (rng.random(p.shape) * 10).astype(np.int32)
    for m, t in ((None, 0), (mask, 2)):
        o = oracle.gridnet(p, -32768, 30.0, 30.0, mask=m, thresh=t, outlets=(ox, oy))
        a = ctx.gridnet(p, -32768, 30.0, 30.0, mask=m, thresh=t, outlets=(ox, oy))
        for x, y, nm in zip(a, o, ("plen", "tlen", "gord")):
            assert bits_equal(x, y), describe_diff(x, y, f"{nm} -o mask={m is not None}")


def test_sweep_and_walk_agree(ctx, oracle, monkeypatch):
    """Two independent schedules of the same dependency sweeps (LDS tiles on the round schedule vs the atomic pull walk): GridNet,
    weighted AreaD8 and D8FlowPathExtremeUp give the same bits; unweighted AreaD8 agrees across tile contraction, sweep and walk."""
    rng = np.random.default_rng(2)
    dem = oracle.synth_dem((1500, 1300), 5)
    p, _, _ = oracle.d8flowdir(oracle.pitremove(dem, -9999.0), -3.0e38, 30.0, 30.0)
    w = (rng.random(p.shape, dtype=np.float32) * 1.0e3).astype(np.float32)
    res = {}
    for mode in ("sweep", "walk"):
        if mode == "walk":
            monkeypatch.setenv("TDX_GN_WALK", "1"); monkeypatch.setenv("TDX_AD8_WALK", "1")
        res[mode] = (ctx.gridnet(p, -32768, 30.0, 30.0), ctx.aread8(p, -32768, weights=w, contcheck=False), ctx.d8flowpathextremeup(p, w, -32768, usemax=False),
                     ctx.aread8(p, -32768))
    monkeypatch.delenv("TDX_GN_WALK"); monkeypatch.delenv("TDX_AD8_WALK")
    gn_o = oracle.gridnet(p, -32768, 30.0, 30.0)
    for a, b, o, nm in zip(res["sweep"][0], res["walk"][0], gn_o, ("plen", "tlen", "gord")):
        assert bits_equal(a, o), describe_diff(a, o, f"gridnet sweep vs oracle: {nm}")
        assert bits_equal(b, o), describe_diff(b, o, f"gridnet walk vs oracle: {nm}")
    for i in (1, 2, 3):
        assert bits_equal(res["sweep"][i], res["walk"][i])
    monkeypatch.setenv("TDX_AD8_SWEEP", "1")
    assert bits_equal(ctx.aread8(p, -32768), res["walk"][3]), "unweighted AreaD8: sweep vs walk"
    monkeypatch.delenv("TDX_AD8_SWEEP")
    assert bits_equal(res["walk"][1], oracle.aread8(p, -32768, weights=w, contcheck=False))
