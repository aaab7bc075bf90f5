"""Inputs that are nothing like a fractal surface, through the C ABI against the pinned restatement, bit for bit (tests/pathological.py):
constant plane, monotone ramps along both axes and both diagonals, a checkerboard of one-cell pits, a spiral channel (dependency depth ~ n / 8
windings), 1 x N / N x 1 / 2 x N rasters, all-nodata, one data cell, NaN and +-Inf cells.  The reference's comparisons are the contract
(src/flood.cpp:307-330, src/d8.cpp:359-409, src/linearpart.h:470-483): a NaN elevation is never `> Z`, so PitRemove never visits it and it keeps
FLT_MAX (src/flood.cpp:295); differences with a NaN are never `> smax`, so the cell is a flat cell that no flat loop resolves.  Whatever the
restatement (pinned to the real tools) makes of these inputs, the tile schedules must make the same of them - and end."""
import numpy as np
import pytest

import pathological as P
from conftest import bits_equal, describe_diff

pytestmark = pytest.mark.gpu
ANG_ND = -3.402823466e38


@pytest.mark.parametrize("name", sorted(P.CASES))
def test_d8_pipeline_on_pathological_input(name, ctx, oracle):
    dem = P.CASES[name](oracle)
    fel_o = oracle.pitremove(dem, P.NODATA)
    fel = ctx.pitremove(dem, P.NODATA)
    assert bits_equal(fel, fel_o), describe_diff(fel, fel_o, f"{name}: fel")
    p_o, sd8_o, st_o = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
    p, sd8, st = ctx.d8flowdir(fel_o, -3.0e38, 30.0, 30.0, stats=True)
    assert bits_equal(sd8, sd8_o), describe_diff(sd8, sd8_o, f"{name}: sd8")
    assert bits_equal(p, p_o), describe_diff(p, p_o, f"{name}: p")
    assert (st["flats_initial"], st["flat_iterations"], st["flats_left"]) == (st_o["flats_initial"], st_o["flat_iterations"], st_o["flats_left"])
    for cc in (True, False):
        a_o = oracle.aread8(p_o, -32768, contcheck=cc)
        a = ctx.aread8(p_o, -32768, contcheck=cc)
        assert bits_equal(a, a_o), describe_diff(a, a_o, f"{name}: ad8 contcheck={cc}")
    w = (np.arange(dem.size, dtype=np.float32).reshape(dem.shape) % np.float32(7.0)) + np.float32(0.5)
    a_o = oracle.aread8(p_o, -32768, weights=w, contcheck=False)
    a = ctx.aread8(p_o, -32768, weights=w, contcheck=False)      # the generic tile dependency sweep
    assert bits_equal(a, a_o), describe_diff(a, a_o, f"{name}: weighted ad8")


@pytest.mark.parametrize("name", sorted(P.CASES))
def test_dinf_on_pathological_input(name, ctx, oracle, monkeypatch):
    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")
    dem = P.CASES[name](oracle)
    fel = oracle.pitremove(dem, P.NODATA)
    ang_o, slp_o, st_o = oracle.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    ang, slp, st = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    assert bits_equal(slp, slp_o), describe_diff(slp, slp_o, f"{name}: slp")
    assert bits_equal(ang, ang_o), describe_diff(ang, ang_o, f"{name}: ang")
    assert (st["flats_initial"], st["flats_left"]) == (st_o["flats_initial"], st_o["flats_left"])
    for cc in (True, False):
        s_o = oracle.areadinf(ang_o, ANG_ND, 30.0, 30.0, contcheck=cc)
        s = ctx.areadinf(ang_o, ANG_ND, 30.0, 30.0, contcheck=cc)
        assert bits_equal(s, s_o), describe_diff(s, s_o, f"{name}: sca contcheck={cc}")


@pytest.mark.parametrize("name", ["plane", "spiral", "checkerboard_pits", "one_row", "nan_cells"])
def test_pathological_input_in_three_strips(name, ctx, oracle):
    """the same bits when the raster is cut into three row strips (in-process rank group): a spiral crosses every strip boundary once per winding"""
    import torch

    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows

    dem = P.CASES[name](oracle)
    ny, nx = dem.shape
    world = 3 if ny >= 3 else 1
    fel_o = oracle.pitremove(dem, P.NODATA)
    p_o, sd8_o, _ = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
    a_o = oracle.aread8(p_o, -32768, contcheck=False)
    if world == 1:
        pytest.skip("fewer rows than strips")
    parts = partition_rows(ny, world)
    with StripGroup(world, nx, [0] * world) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            pipe = StripPipeline(c, comm, nx, y1 - y0)
            d = pipe.empty(torch.float32)
            d[1:y1 - y0 + 1] = torch.from_numpy(dem[y0:y1]).cuda()
            fel, _ = pipe.pitremove(d, P.NODATA)
            p, sd8, _ = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0)
            a, _ = pipe.aread8(p, -32768, contcheck=False)
            sl = slice(1, y1 - y0 + 1)
            return {"fel": fel[sl].cpu().numpy(), "p": p[sl].cpu().numpy(), "sd8": sd8[sl].cpu().numpy(), "ad8": a[sl].cpu().numpy()}
        res = grp.run(rank_main)
    for key, ref in (("fel", fel_o), ("p", p_o), ("sd8", sd8_o), ("ad8", a_o)):
        got = np.concatenate([r[key] for r in res], axis=0)
        assert bits_equal(got, ref), describe_diff(got, ref, f"{name} in {world} strips: {key}")


@pytest.mark.slow
def test_spiral_at_4096_completes(ctx, oracle, capsys):
    """The schedule's worst case as a known number: a spiral channel at 4096^2 (pitch 8: 512 windings, a flow path of ~2 M cells that crosses
    ~64 tiles per winding).  The rounds of every stage are recorded; the test only insists that the stages END, that the
    result is flood()'s (orc_pitremove_check) and aread8()'s (orc_aread8_check) on every cell, and that the whole pipeline stays under a minute."""
    import time

    import torch

    dem = torch.from_numpy(P.spiral(4096, 8)).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fel, s1 = ctx.pitremove(dem, P.NODATA, stats=True)
    p, sd8, s2 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    a, s3 = ctx.aread8(p, -32768, contcheck=False, stats=True)
    aw, s4 = ctx.aread8(p, -32768, weights=torch.ones_like(fel), contcheck=False, stats=True)
    ang, slp, s5 = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    sca, s6 = ctx.areadinf(ang, ANG_ND, 30.0, 30.0, contcheck=False, stats=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    with capsys.disabled():
        print("\nspiral 4096^2: " + ", ".join(f"{k} {s['ms_total']:.1f} ms / {s['rounds']} rounds" for k, s in
                                               (("pitremove", s1), ("d8flowdir", s2), ("aread8", s3), ("aread8 -wg", s4), ("dinfflowdir", s5), ("areadinf", s6))) + f"; wall {wall:.2f} s")
    assert wall < 60.0
    bad, first, _ = oracle.pitremove_check(dem.cpu().numpy(), fel.cpu().numpy(), P.NODATA)
    assert bad == 0, (bad, first)
    bad, first, _ = oracle.aread8_check(p.cpu().numpy(), a.cpu().numpy(), -32768, contcheck=False)
    assert bad == 0, (bad, first)
    assert torch.equal(a, aw), "tile contraction and dependency sweep disagree (unit weights, counts below 2^24)"
    assert float(a.max()) > 1e6
