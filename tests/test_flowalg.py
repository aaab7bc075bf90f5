"""The D-infinity flow algebra tools of SURVEY.md 8f rank 4 (DinfUpDependence, DinfRevAccum, DinfConcLimAccum, DinfTransLimAccum): the C restatement against the rasters
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


def _outl(h):
    return np.ascontiguousarray(h["outlets_xy"][:, 0]), np.ascontiguousarray(h["outlets_xy"][:, 1])   # (column indices, row indices)


CONC_RUNS = [("ctpt", dict(csol=2.5)), ("ctpt_nc", dict(contcheck=False)), ("ctpt_outlets_nc", dict(contcheck=False, outlets=True))]
TRANS_RUNS = [(("tla", "tdep", None), dict()), (("tla_cs_nc", "tdep_cs_nc", "tctpt_cs_nc"), dict(cs=True, contcheck=False)),
              (("tla_cs_outlets_nc", "tdep_cs_outlets_nc", "tctpt_cs_outlets_nc"), dict(cs=True, contcheck=False, outlets=True))]


def _run_limited(impl, g, h):
    """Every DinfConcLimAccum / DinfTransLimAccum run of the golden generator through `impl` (the oracle module or a Context)."""
    out = {}
    ang = np.ascontiguousarray(g["ang"])
    for key, kw in CONC_RUNS:
        kw = dict(kw)
        if kw.pop("outlets", False):
            kw["outlets"] = _outl(h)
        out[key] = impl.dinfconclimaccum(ang, h["dm2"], h["dgs"], h["q"], dx=g["dxc"], dy=g["dyc"], **kw)
    for keys, kw in TRANS_RUNS:
        kw = dict(kw)
        if kw.pop("outlets", False):
            kw["outlets"] = _outl(h)
        if kw.pop("cs", False):
            kw["cs"] = h["cs"]
        res = impl.dinftranslimaccum(ang, h["tsup"], h["tc"], dx=g["dxc"], dy=g["dyc"], **kw)
        for k, r in zip(keys, res):
            if k:
                out[k] = r
    return out


@pytest.mark.parametrize("name", CASES)
def test_oracle_limited_accumulations_match_reference(name, oracle):
    g, h = load_golden(name), load_golden_flowalg(name)
    assert int((h["ctpt_outlets_nc"] > -1e30).sum()) > 10 and int((h["tla"] > -1e30).sum()) > 1000   # the goldens are not trivially empty
    for k, r in _run_limited(oracle, g, h).items():
        assert bits_equal(r, h[k]), describe_diff(r, h[k], k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_limited_accumulations_match_reference(name, ctx):
    g, h = load_golden(name), load_golden_flowalg(name)
    for k, r in _run_limited(ctx, g, h).items():
        assert bits_equal(r, h[k]), describe_diff(r, h[k], k)


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


@pytest.mark.gpu
@pytest.mark.parametrize("ngpus", [1, 3])
def test_cli_limited_accumulations(tmp_path, ngpus):
    """bin/dinfconclimaccum and bin/dinftranslimaccum (flags of src/DinfConcLimAccummn.cpp, src/DinfTransLimAccummn.cpp) on files, one GPU and
    three row strips, against the rasters of the reference tools."""
    g, h = load_golden("holes"), load_golden_flowalg("holes")
    ny, nx = g["ang"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    gt = (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("xang.tif"), np.ascontiguousarray(g["ang"]), -3.402823466e38, geotransform=gt)
    for nm, key in (("xdm", "dm2"), ("xq", "q"), ("xtsup", "tsup"), ("xtc", "tc"), ("cs", "cs")):
        T.write_raster(f(nm + ".tif"), np.ascontiguousarray(h[key]), -9999.0, geotransform=gt)
    T.write_raster(f("xdg.tif"), np.ascontiguousarray(h["dgs"]), -1, geotransform=gt)
    with open(f("outlets.txt"), "w") as fo:
        for x, y in h["outlets_xy"]:
            fo.write(f"{float(gt[0] + (int(x) + 0.5) * gt[1])!r} {float(gt[3] + (int(y) + 0.5) * gt[5])!r}\n")
    N = ["--gpus", str(ngpus)] if ngpus > 1 else []

    def run(tool, *args):
        r = subprocess.run([os.path.join(BIN, tool), *args], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    def same(path, key):
        a, _ = T.read_raster(path, np.float32)
        assert bits_equal(a, h[key]), describe_diff(a, h[key], key)

    out = run("dinfconclimaccum", *N, f("x.tif"))                     # simple usage: xang / xdg / xdm / xq -> xctpt, csol 1
    assert "DinfConcLimAccum version 5.4.0" in out and f"Processors: {ngpus}" in out
    base = ["-ang", f("xang.tif"), "-dg", f("xdg.tif"), "-dm", f("xdm.tif"), "-q", f("xq.tif")]
    run("dinfconclimaccum", *base, "-ctpt", f("c25.tif"), "-csol", "2.5", *N)
    same(f("c25.tif"), "ctpt")
    run("dinfconclimaccum", *base, "-ctpt", f("cnc.tif"), "-nc", *N)
    same(f("cnc.tif"), "ctpt_nc")
    run("dinfconclimaccum", *base, "-ctpt", f("co.tif"), "-nc", "-o", f("outlets.txt"), *N)
    same(f("co.tif"), "ctpt_outlets_nc")
    assert T.raster_info(f("xctpt.tif"))["nodata"] == pytest.approx(-3.402823466e38)

    out = run("dinftranslimaccum", *N, f("x.tif"))                    # simple usage: xang / xtsup / xtc -> xtla, xtdep
    assert "DinfTransLimAccum version 5.4.0" in out
    same(f("xtla.tif"), "tla")
    same(f("xtdep.tif"), "tdep")
    tb = ["-ang", f("xang.tif"), "-tsup", f("xtsup.tif"), "-tc", f("xtc.tif")]
    run("dinftranslimaccum", *tb, "-tla", f("t2.tif"), "-tdep", f("d2.tif"), "-cs", f("cs.tif"), "-ctpt", f("c2.tif"), "-nc", *N)
    same(f("t2.tif"), "tla_cs_nc"); same(f("d2.tif"), "tdep_cs_nc"); same(f("c2.tif"), "tctpt_cs_nc")
    run("dinftranslimaccum", *tb, "-tla", f("t3.tif"), "-tdep", f("d3.tif"), "-cs", f("cs.tif"), "-ctpt", f("c3.tif"), "-nc", "-o", f("outlets.txt"), *N)
    same(f("t3.tif"), "tla_cs_outlets_nc"); same(f("d3.tif"), "tdep_cs_outlets_nc"); same(f("c3.tif"), "tctpt_cs_outlets_nc")
    run("dinftranslimaccum", *tb, "-tla", f("t4.tif"), "-tdep", f("d4.tif"), "-cs", f("cs.tif"), *N)   # -cs without -ctpt: no concentration
    same(f("t4.tif"), "tla")
    # the reference's own unmodified mains on the shim (oracle/Makefile `shimmed`)
    ref = os.path.join(ROOT, "oracle", "_ref")
    if ngpus == 1 and os.path.exists(os.path.join(ref, "shim_dinfconclimaccum")):
        for tool, args, path, key in (("dinfconclimaccum", [*base, "-ctpt", f("s1.tif"), "-csol", "2.5"], f("s1.tif"), "ctpt"),
                                      ("dinftranslimaccum", [*tb, "-tla", f("s2.tif"), "-tdep", f("s3.tif"), "-cs", f("cs.tif"), "-ctpt", f("s4.tif"), "-nc"], f("s4.tif"),
                                       "tctpt_cs_nc")):
            r = subprocess.run([os.path.join(ref, "shim_" + tool), *args], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stdout + r.stderr
            same(path, key)


@pytest.mark.gpu
def test_gpu_limited_vs_oracle_larger(ctx, oracle):
    """Both limited accumulations on a 1100 x 900 raster with nodata holes (several tiles in both geometries), against the restatement."""
    rng = np.random.default_rng(77)
    dem = oracle.synth_dem((1100, 900), 47)
    dem[300:330, 500:640] = -9999.0
    ang, _, _ = oracle.dinfflowdir(oracle.pitremove(dem, -9999.0), -3.0e38, 30.0, 20.0)
    shp = ang.shape
    dm = (0.9 + 0.1 * rng.random(shp, dtype=np.float32)).astype(np.float32)
    dm[rng.random(shp) < 0.001] = -9999.0
    q = (0.5 + rng.random(shp, dtype=np.float32)).astype(np.float32)
    q[rng.random(shp) < 0.002] = 0.0
    dg = (rng.random(shp) < 0.01).astype(np.int16)
    tsup = rng.random(shp, dtype=np.float32)
    tc = (rng.random(shp, dtype=np.float32) * 50).astype(np.float32)
    tc[rng.random(shp) < 0.001] = -9999.0
    cs = rng.random(shp, dtype=np.float32)
    outl = (np.array([450, 120], dtype=np.int32), np.array([1000, 700], dtype=np.int32))
    for kw in (dict(), dict(contcheck=False, outlets=outl)):
        a = ctx.dinfconclimaccum(ang, dm, dg, q, csol=3.0, dx=30.0, dy=20.0, **kw)
        b = oracle.dinfconclimaccum(ang, dm, dg, q, csol=3.0, dx=30.0, dy=20.0, **kw)
        assert bits_equal(a, b), describe_diff(a, b, "ctpt")
        for c_in in (None, cs):
            ra = ctx.dinftranslimaccum(ang, tsup, tc, cs=c_in, dx=30.0, dy=20.0, **kw)
            rb = oracle.dinftranslimaccum(ang, tsup, tc, cs=c_in, dx=30.0, dy=20.0, **kw)
            for x, y, nm in zip(ra, rb, ("tla", "tdep", "ctpt")):
                if y is not None:
                    assert bits_equal(x, y), describe_diff(x, y, nm)


@pytest.mark.gpu
def test_every_sweep_tool_on_both_tile_geometries(ctx, oracle, monkeypatch):
    """A raster large enough (3100 x 2900 = 8 827 tiles of 32 x 32 > TDX_D8_BULK_UNTIL) that the generic dependency sweep starts on 32 x 32
    tiles and hands over to 64 x 64 tiles: every tool built on it, forward and reverse, bit for bit against the restatement.  (The
    directions come from the restatement's FlowDir tools so that both sides sweep the same graph.)"""
    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")   # every sweep is re-checked cell by cell against its contributors' final records
    rng = np.random.default_rng(2024)
    shape = (3100, 2900)
    dem = oracle.synth_dem(shape, 61)
    dem[1200:1260, 800:1100] = -9999.0
    fel = ctx.pitremove(dem, -9999.0)
    p, _ = ctx.d8flowdir(fel, -3.0e38, 30.0, 25.0)
    ang, _ = ctx.dinfflowdir(fel, -3.0e38, 30.0, 25.0)
    w = (rng.random(shape, dtype=np.float32) * 10.0).astype(np.float32)
    w[rng.random(shape) < 0.001] = -9999.0
    outl = (np.array([1450, 300], dtype=np.int32), np.array([2900, 1700], dtype=np.int32))

    def same(a, b, name):
        assert bits_equal(a, b), describe_diff(a, b, name)

    same(ctx.aread8(p, -32768, weights=w, weights_nodata=-9999.0), oracle.aread8(p, -32768, weights=w, weights_nodata=-9999.0), "weighted ad8")
    same(ctx.aread8(p, -32768, weights=w, weights_nodata=-9999.0, contcheck=False, outlets=outl),
         oracle.aread8(p, -32768, weights=w, weights_nodata=-9999.0, contcheck=False, outlets=outl), "weighted ad8, outlets")
    same(ctx.aread8(p, -32768, contcheck=False, outlets=outl), oracle.aread8(p, -32768, contcheck=False, outlets=outl), "ad8, outlets (tile contraction)")
    same(ctx.d8flowpathextremeup(p, w, -32768, usemax=True), oracle.d8flowpathextremeup(p, w, -32768, usemax=True), "ssa")
    same(ctx.d8flowpathextremeup(p, w, -32768, usemax=False, contcheck=False, outlets=outl),
         oracle.d8flowpathextremeup(p, w, -32768, usemax=False, contcheck=False, outlets=outl), "ssa min, outlets")
    for a, b, nm in zip(ctx.gridnet(p, -32768, 30.0, 25.0), oracle.gridnet(p, -32768, 30.0, 25.0), ("plen", "tlen", "gord")):
        same(np.asarray(a), np.asarray(b), nm)
    mask = rng.integers(0, 10, shape).astype(np.int32)
    for a, b, nm in zip(ctx.gridnet(p, -32768, 30.0, 25.0, mask=mask, thresh=2, outlets=outl),
                        oracle.gridnet(p, -32768, 30.0, 25.0, mask=mask, thresh=2, outlets=outl), ("plen (mask, outlets)", "tlen", "gord")):
        same(np.asarray(a), np.asarray(b), nm)
    same(ctx.areadinf(ang, dx=30.0, dy=25.0, weights=np.abs(w), contcheck=False, outlets=outl),
         oracle.areadinf(ang, dx=30.0, dy=25.0, weights=np.abs(w), contcheck=False, outlets=outl), "sca, weights + outlets")
    dm = (0.9 + 0.1 * rng.random(shape, dtype=np.float32)).astype(np.float32)
    same(ctx.dinfdecayaccum(ang, dm, dx=30.0, dy=25.0, weights=np.abs(w)), oracle.dinfdecayaccum(ang, dm, dx=30.0, dy=25.0, weights=np.abs(w)), "dsca")
    dg = (rng.random(shape) < 0.003).astype(np.int32)
    same(ctx.dinfupdependence(ang, dg, dx=30.0, dy=25.0), oracle.dinfupdependence(ang, dg, dx=30.0, dy=25.0), "dep")
    for a, b, nm in zip(ctx.dinfrevaccum(ang, w, dx=30.0, dy=25.0), oracle.dinfrevaccum(ang, w, dx=30.0, dy=25.0), ("racc", "dmax")):
        same(a, b, nm)
    q = (0.5 + rng.random(shape, dtype=np.float32)).astype(np.float32)
    same(ctx.dinfconclimaccum(ang, dm, dg.astype(np.int16), q, csol=1.5, dx=30.0, dy=25.0),
         oracle.dinfconclimaccum(ang, dm, dg.astype(np.int16), q, csol=1.5, dx=30.0, dy=25.0), "ctpt")
    tc = (rng.random(shape, dtype=np.float32) * 80).astype(np.float32)
    for a, b, nm in zip(ctx.dinftranslimaccum(ang, np.abs(w), tc, cs=dm, dx=30.0, dy=25.0, contcheck=False, outlets=outl),
                        oracle.dinftranslimaccum(ang, np.abs(w), tc, cs=dm, dx=30.0, dy=25.0, contcheck=False, outlets=outl), ("tla", "tdep", "ctpt")):
        same(a, b, nm)


@pytest.mark.gpu
@pytest.mark.slow
def test_sweep_tools_three_strips_equal_one_gpu_at_size(ctx, oracle, tmp_path, monkeypatch):
    """3100 x 2900 again, through files and `--gpus 3`: the strip protocol of the generic sweep (record rows exchanged as bit patterns, tiles
    re-activated by changed halo cells, forward and reverse) must reproduce the one-GPU rasters bit for bit."""
    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")   # (inherited by the tools: the verifier runs on every strip)
    rng = np.random.default_rng(77)
    shape = (3100, 2900)
    dem = oracle.synth_dem(shape, 62)
    fel = ctx.pitremove(dem, -9999.0)
    p, _ = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    ang, _ = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    w = (rng.random(shape, dtype=np.float32) * 10.0).astype(np.float32)
    tc = (rng.random(shape, dtype=np.float32) * 80).astype(np.float32)
    gt = (0.0, 30.0, 0.0, 30.0 * shape[0], 0.0, -30.0)
    f = lambda s: str(tmp_path / s)  # noqa: E731
    T.write_raster(f("p.tif"), np.ascontiguousarray(p), -32768, geotransform=gt)
    T.write_raster(f("ang.tif"), np.ascontiguousarray(ang), -3.402823466e38, geotransform=gt)
    T.write_raster(f("w.tif"), w, -9999.0, geotransform=gt)
    T.write_raster(f("tc.tif"), tc, -9999.0, geotransform=gt)

    def run(tool, *args):
        r = subprocess.run([os.path.join(BIN, tool), "--gpus", "3", *args], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and ("Processors: 3" in r.stdout or "Processes: 3" in r.stdout), r.stdout[-2000:] + r.stderr[-2000:]

    def same(path, a, name, dt=np.float32):
        b, _ = T.read_raster(path, dt)
        assert bits_equal(b, np.asarray(a)), describe_diff(b, np.asarray(a), name)

    run("aread8", "-p", f("p.tif"), "-ad8", f("ad8w.tif"), "-wg", f("w.tif"))
    same(f("ad8w.tif"), ctx.aread8(p, -32768, weights=w, weights_nodata=-9999.0), "weighted ad8")
    run("gridnet", "-p", f("p.tif"), "-plen", f("plen.tif"), "-tlen", f("tlen.tif"), "-gord", f("gord.tif"))
    pl, tl, go = ctx.gridnet(p, -32768, 30.0, 30.0)
    same(f("plen.tif"), pl, "plen"); same(f("tlen.tif"), tl, "tlen"); same(f("gord.tif"), go, "gord", np.int16)
    run("dinfrevaccum", "-ang", f("ang.tif"), "-wg", f("w.tif"), "-racc", f("racc.tif"), "-dmax", f("dmax.tif"))
    ra, dm = ctx.dinfrevaccum(ang, w, dx=30.0, dy=30.0)
    same(f("racc.tif"), ra, "racc"); same(f("dmax.tif"), dm, "dmax")
    run("dinftranslimaccum", "-ang", f("ang.tif"), "-tsup", f("w.tif"), "-tc", f("tc.tif"), "-tla", f("tla.tif"), "-tdep", f("tdep.tif"))
    tla, tdep, _ = ctx.dinftranslimaccum(ang, w, tc, dx=30.0, dy=30.0)
    same(f("tla.tif"), tla, "tla"); same(f("tdep.tif"), tdep, "tdep")


@pytest.mark.gpu
@pytest.mark.parametrize("shape,seed", [((1, 1), 1), ((1, 9), 2), ((3, 3), 3), ((2, 70), 4), ((65, 64), 5), ((64, 129), 6), ((130, 67), 7)])
def test_flow_algebra_on_small_and_ragged_rasters(shape, seed, ctx, oracle):
    """Degenerate and tile-edge shapes (one cell, one row, exactly one tile plus a row / a column) with RANDOM angles - cycles, flow off the
    raster, nodata angles and inputs everywhere: all four flow-algebra tools and DinfDecayAccum against the restatement."""
    rng = np.random.default_rng(seed)
    ang = (rng.random(shape) * 2 * np.pi).astype(np.float32)
    ang[rng.random(shape) < 0.1] = -3.402823466e38          # no angle
    ang[rng.random(shape) < 0.05] = -1.0                      # unresolved flat (a valid angle value < 0: prop() is 0 everywhere)
    w = (rng.random(shape) * 5).astype(np.float32)
    w[rng.random(shape) < 0.1] = -9999.0
    dm = (0.5 + rng.random(shape) * 0.5).astype(np.float32)
    dm[rng.random(shape) < 0.05] = -9999.0
    q = (rng.random(shape) * 2 - 0.2).astype(np.float32)      # some q <= 0
    dg16 = (rng.random(shape) < 0.2).astype(np.int16)
    tc = (rng.random(shape) * 3).astype(np.float32)
    cs = rng.random(shape).astype(np.float32)
    cs[rng.random(shape) < 0.05] = -9999.0

    def same(a, b, name):
        assert bits_equal(a, b), describe_diff(a, b, name)

    for cc in (True, False):
        same(ctx.dinfdecayaccum(ang, dm, dx=10.0, dy=7.0, weights=np.abs(w), contcheck=cc), oracle.dinfdecayaccum(ang, dm, dx=10.0, dy=7.0, weights=np.abs(w), contcheck=cc), "dsca")
        same(ctx.dinfconclimaccum(ang, dm, dg16, q, csol=0.7, dx=10.0, dy=7.0, contcheck=cc), oracle.dinfconclimaccum(ang, dm, dg16, q, csol=0.7, dx=10.0, dy=7.0, contcheck=cc), "ctpt")
        for c_in in (None, cs):
            for x, y, nm in zip(ctx.dinftranslimaccum(ang, w, tc, cs=c_in, dx=10.0, dy=7.0, contcheck=cc), oracle.dinftranslimaccum(ang, w, tc, cs=c_in, dx=10.0, dy=7.0, contcheck=cc),
                                ("tla", "tdep", "ctpt")):
                if y is not None:
                    same(x, y, nm)
    same(ctx.dinfupdependence(ang, dg16.astype(np.int32), dx=10.0, dy=7.0), oracle.dinfupdependence(ang, dg16.astype(np.int32), dx=10.0, dy=7.0), "dep")
    for x, y, nm in zip(ctx.dinfrevaccum(ang, w, dx=10.0, dy=7.0), oracle.dinfrevaccum(ang, w, dx=10.0, dy=7.0), ("racc", "dmax")):
        same(x, y, nm)
