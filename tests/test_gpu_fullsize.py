"""Full-size (BASELINE.json configs[1], 16384 x 16384) checks through size-independent properties - the CPU oracle
would need hours for flat resolution at this size (O(N^1.5)); the same properties are checked against the oracle at
small sizes first, so that a property failure at full size means a size-dependent defect."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def _certify_fel(oracle, dem, fel, what):
    """Every cell of `fel` against the linear-time PitRemove certificate of the restatement (oracle/taudem_oracle.c: orc_pitremove_check - the fixed-point
    equation of flood()'s relaxation per cell, src/flood.cpp:243-271,292-331, plus the flood from the seed cells that must reach every data cell; pinned
    on CPU to the REAL tool's rasters, tests/test_oracle_vs_golden.py).  dem / fel: torch tensors or numpy arrays."""
    d = dem.cpu().numpy() if hasattr(dem, "cpu") else dem
    f = fel.cpu().numpy() if hasattr(fel, "cpu") else fel
    bad, first, reached = oracle.pitremove_check(d, f, -9999.0)
    nx = d.shape[1]
    assert bad == 0, f"{what}: {bad} cells of fel fail flood()'s certificate; first at row {first // nx} column {first % nx}"
    assert reached == d.size, f"{what}: the flood from the seed cells reached {reached} of {d.size} cells"


def _props(ctx, n, seed, monkeypatch, oracle=None, certify=None):
    import torch

    dem = ctx.synth_dem(n, seed=seed)
    fel = ctx.pitremove(dem, -9999.0)
    # (1) the filled surface is a fixed point of the fill and never below the input (src/flood.cpp:307-330)
    assert bool((fel >= dem).all())
    assert torch.equal(ctx.pitremove(fel, -9999.0), fel), "pitremove is not idempotent"
    if certify is not None:   # ... and it is flood()'s surface on every cell (no over-filled cell, no under-filled basin)
        _certify_fel(certify, dem, fel, f"{n} x {n}")
    # (2) no interior pit remains: every interior cell has a neighbour that is not higher
    f = fel[1:-1, 1:-1]
    lower_eq = torch.zeros_like(f, dtype=torch.bool)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy or dx:
                lower_eq |= fel[1 + dy:n - 1 + dy, 1 + dx:n - 1 + dx] <= f
    assert bool(lower_eq.all()), "a cell of the filled surface is strictly below all its neighbours"
    p, sd8, st = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    # (3) every interior cell got a direction; the edge ring is nodata (src/d8.cpp:383-386); slopes of directed cells >= 0
    assert st["flats_left"] == 0
    inner = p[1:-1, 1:-1]
    assert bool(((inner >= 1) & (inner <= 8)).all())
    assert bool((p[0] == -32768).all() and (p[-1] == -32768).all() and (p[:, 0] == -32768).all() and (p[:, -1] == -32768).all())
    assert bool((sd8[1:-1, 1:-1] >= 0).all())
    # (4) flow never goes uphill on the filled surface
    d1 = [0, 1, 1, 0, -1, -1, -1, 0, 1]
    d2 = [0, 0, -1, -1, -1, 0, 1, 1, 1]
    for k in range(1, 9):
        m = inner == k
        zn = fel[1 + d2[k]:n - 1 + d2[k], 1 + d1[k]:n - 1 + d1[k]]
        assert bool((zn[m] <= f[m]).all()), f"direction {k} points uphill"
    # (5) AreaD8: the tile-contraction path and the dependency walk agree bit for bit (incl. the cells above 2^24)
    a_t = ctx.aread8(p, -32768, contcheck=False)
    monkeypatch.setenv("TDX_AD8_WALK", "1")
    a_w = ctx.aread8(p, -32768, contcheck=False)
    monkeypatch.delenv("TDX_AD8_WALK")
    assert torch.equal(a_t.view(torch.int32), a_w.view(torch.int32)), "tile contraction and walk differ"
    # (6) conservation: every interior cell is counted exactly once at the cells that drain into the nodata ring
    ring_sum = 0.0
    for k in range(1, 9):
        m = inner == k
        tgt_is_ring = torch.zeros_like(m)
        yy, xx = torch.nonzero(m, as_tuple=True)
        ty, tx = yy + 1 + d2[k], xx + 1 + d1[k]
        on_ring = (ty == 0) | (ty == n - 1) | (tx == 0) | (tx == n - 1)
        ring_sum += float(a_t[1:-1, 1:-1][yy[on_ring], xx[on_ring]].double().sum())
    if float(a_t.max()) <= 2 ** 24:
        assert ring_sum == float((n - 2) * (n - 2)), (ring_sum, (n - 2) ** 2)
    else:   # float32 adds above 2^24 drop small tributaries exactly like the reference does (ulp 2..16 on the trunks)
        assert abs(ring_sum - (n - 2) ** 2) <= 1e-3 * (n - 2) ** 2
    if oracle is not None:
        fel_o = oracle.pitremove(dem.cpu().numpy(), -9999.0)
        assert np.array_equal(fel.cpu().numpy().view(np.uint32), fel_o.view(np.uint32))
        p_o, _, _ = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
        assert np.array_equal(p.cpu().numpy(), p_o)
        assert np.array_equal(a_t.cpu().numpy().view(np.uint32), oracle.aread8(p_o, -32768, contcheck=False).view(np.uint32))
    return float(a_t.max())


def test_properties_hold_where_the_oracle_confirms_them(ctx, oracle, monkeypatch):
    _props(ctx, 700, 3, monkeypatch, oracle, certify=oracle)


def test_properties_at_16384(ctx, oracle, monkeypatch):
    amax = _props(ctx, 16384, 1234, monkeypatch, certify=oracle)
    assert amax > 2 ** 24      # the exact re-evaluation path was exercised


def _dinf_props(ctx, n, seed, monkeypatch, oracle=None):
    """D-infinity at full size: angles in range, every interior cell resolved, the tile dependency sweep and the atomic pull walk
    (two independent schedules) agree bit for bit on sca, and flow is conserved at the raster's rim within float32 rounding.
    Every sweep runs under TDX_SWEEP_VERIFY=1 (a cell that does not follow from its contributors' final values fails the call).
    With `oracle`: the restatement's linear-time checkers pin the rasters to the reference's per-cell expressions on the host -
    the first pass of setdir() on every cell it gives a direction (src/dinf.cpp:549-593), area()'s loop body on every cell
    (src/areadinf.cpp:187-217; by induction over the dependency order that IS the raster the restatement would produce)."""
    import math

    import torch

    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")
    dem = ctx.synth_dem(n, seed=seed)
    fel = ctx.pitremove(dem, -9999.0)
    if oracle is not None:   # the input of everything below, on every cell
        _certify_fel(oracle, dem, fel, f"{n} x {n}")
    del dem
    ang, slp, st = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    if oracle is not None:
        bad, first, flats = oracle.dinf_first_pass_check(fel.cpu().numpy(), ang.cpu().numpy(), slp.cpu().numpy(), dx=30.0, dy=30.0)
        assert bad == 0, f"{bad} cells differ from the first pass of setdir(); first at row {first // n} column {first % n}"
        assert flats == st["flats_initial"], (flats, st["flats_initial"])
        # EVERY cell of ang / slp - the flat cells' angles included (a third of the raster at 32768^2): the restatement itself, with its flat loops
        # as breadth-first searches (linear time; pinned to the real reference's rasters and 2048^2 / 4096^2 digests on CPU), run on the host cores
        import os
        monkeypatch.setenv("ORC_FLATS", "bfs")
        oracle.set_threads(os.cpu_count() or 1)
        try:
            ang_o, slp_o, st_o = oracle.dinfflowdir(fel.cpu().numpy(), -3.0e38, 30.0, 30.0)
        finally:
            oracle.set_threads(1)
            monkeypatch.delenv("ORC_FLATS")
        assert (st_o["flats_initial"], st_o["flat_iterations"], st_o["flats_left"]) == (st["flats_initial"], st["flat_iterations"], st["flats_left"])
        ang_h = ang.cpu().numpy()
        neq = int(np.count_nonzero(ang_h.view(np.uint32) != ang_o.view(np.uint32)))
        assert neq == 0, f"ang: {neq} cells differ from the restatement (flat cells included)"
        assert np.array_equal(slp.cpu().numpy().view(np.uint32), slp_o.view(np.uint32)), "slp differs from the restatement"
        del ang_o, slp_o, ang_h
        sca_c = ctx.areadinf(ang, dx=30.0, dy=30.0, contcheck=True)      # the tool's default mode
        bad, first = oracle.areadinf_check(ang.cpu().numpy(), sca_c.cpu().numpy(), dx=30.0, dy=30.0, contcheck=True)
        assert bad == 0, f"contcheck: {bad} cells do not follow from area()'s expression; first at row {first // n} column {first % n}"
        del sca_c
    del fel
    assert st["flats_left"] == 0
    inner = ang[1:-1, 1:-1]
    assert bool(((inner >= 0) & (inner <= 2 * math.pi + 1e-6)).all()), "an interior cell has no D-infinity angle"
    assert bool((slp[1:-1, 1:-1] >= 0).all())
    ring_nd = -3.402823466e38
    assert bool((ang[0] == ring_nd).all() and (ang[-1] == ring_nd).all() and (ang[:, 0] == ring_nd).all() and (ang[:, -1] == ring_nd).all())
    del slp
    sca_t = ctx.areadinf(ang, dx=30.0, dy=30.0, contcheck=False)
    monkeypatch.setenv("TDX_DINF_WALK", "1")
    sca_w = ctx.areadinf(ang, dx=30.0, dy=30.0, contcheck=False)
    monkeypatch.delenv("TDX_DINF_WALK")
    assert torch.equal(sca_t.view(torch.int32), sca_w.view(torch.int32)), "tile dependency sweep and pull walk differ"
    del sca_w
    assert bool((sca_t[1:-1, 1:-1] >= 30.0).all()), "an interior cell was never evaluated"
    if oracle is not None:
        bad, first = oracle.areadinf_check(ang.cpu().numpy(), sca_t.cpu().numpy(), dx=30.0, dy=30.0, contcheck=False)
        assert bad == 0, f"{bad} cells do not follow from area()'s expression; first at row {first // n} column {first % n}"
    # conservation: what leaves through the nodata ring = every interior cell's own dx (float32 sums along the trunks round)
    k_d1 = [0, 1, 1, 0, -1, -1, -1, 0, 1]
    k_d2 = [0, 0, -1, -1, -1, 0, 1, 1, 1]
    total = 0.0
    a64 = inner.double()
    s64 = sca_t[1:-1, 1:-1].double()
    q = math.pi / 4      # square cells: the facet directions are multiples of pi/4
    for k in range(1, 9):
        lo, mid, hi = (k - 2) * q, (k - 1) * q, k * q
        aa = torch.where((k == 1) & (a64 > math.pi), a64 - 2 * math.pi, a64) if k == 1 else a64
        p = torch.where((aa > lo) & (aa < hi), torch.where(aa > mid, (hi - aa) / (hi - mid), (aa - lo) / (mid - lo)), torch.zeros_like(aa))
        p = torch.where(p < 1e-5, torch.zeros_like(p), p)
        yy, xx = torch.nonzero(p > 0, as_tuple=True)
        ty, tx = yy + 1 + k_d2[k], xx + 1 + k_d1[k]
        on_ring = (ty == 0) | (ty == n - 1) | (tx == 0) | (tx == n - 1)
        total += float((p[yy[on_ring], xx[on_ring]] * s64[yy[on_ring], xx[on_ring]]).sum())
    want = 30.0 * (n - 2) * (n - 2)
    assert abs(total - want) <= 2e-3 * want, (total, want)


def test_dinf_properties_small(ctx, oracle, monkeypatch):
    _dinf_props(ctx, 700, 3, monkeypatch, oracle)
    # the checkers agree with the restatement itself where it runs in a moment - and they do notice a single wrong bit
    dem = oracle.synth_dem(500, 9)
    fel = oracle.pitremove(dem, -9999.0)
    ang, slp, _ = oracle.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    sca = oracle.areadinf(ang, dx=30.0, dy=30.0)
    assert oracle.areadinf_check(ang, sca, dx=30.0, dy=30.0)[0] == 0
    assert oracle.dinf_first_pass_check(fel, ang, slp, dx=30.0, dy=30.0)[0] == 0
    sca[250, 250] = np.nextafter(sca[250, 250], np.float32(1e30))
    assert oracle.areadinf_check(ang, sca, dx=30.0, dy=30.0)[0] >= 1


def test_dinf_properties_at_16384(ctx, oracle, monkeypatch):
    _dinf_props(ctx, 16384, 1234, monkeypatch, oracle)


def _host_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:   # no psutil: assume the box is as large as every MI355X host seen so far
        return 1024.0


def test_dinf_config3_at_32768(ctx, oracle, monkeypatch):
    """BASELINE.json configs[2] at its own size (1.07 G cells, 4 x the tiles and twice the rounds of 16384^2): the properties, the sweep
    verifier, and the linear-time host checks of ang / slp / sca against the restatement's expressions."""
    if _host_gb() < 96:
        pytest.skip("needs ~60 GB of host memory for the linear-time checks")
    _dinf_props(ctx, 32768, 1234, monkeypatch, oracle)


def test_dinf_config3_three_strips_equal_one(ctx, monkeypatch):
    """32768^2 as three row strips (in-process rank group, peer transport: three contexts on this GPU) reproduces the one-strip
    rasters of DinfFlowDir and AreaDinf bit for bit - the strip protocol at configs[2]'s size, under the sweep verifier."""
    import torch

    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows
    import taudem_amd as T

    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")
    n, size = 32768, 3
    dem = ctx.synth_dem(n, seed=1234)
    fel1 = ctx.pitremove(dem, -9999.0)
    del dem
    ang1, slp1 = ctx.dinfflowdir(fel1, -3.0e38, 30.0, 30.0)
    sca1 = ctx.areadinf(ang1, dx=30.0, dy=30.0)
    parts = partition_rows(n, size)
    wl = T.synth_base_wavelength(n)
    with StripGroup(size, n) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            nyl = y1 - y0
            pipe = StripPipeline(c, comm, n, nyl)
            d = pipe.empty(torch.float32)
            c.synth_dem((nyl, n), seed=1234, x0=0, y0=y0, base_wavelength=wl, out=d[1:nyl + 1])
            fel, _ = pipe.pitremove(d, -9999.0)
            del d
            ang, slp, _ = pipe.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
            sca, _ = pipe.areadinf(ang, dx=30.0, dy=30.0)
            torch.cuda.synchronize()
            ok = {k: bool(torch.equal(a[1:nyl + 1].view(torch.int32), b[y0:y1].view(torch.int32)))
                  for k, a, b in (("fel", fel, fel1), ("ang", ang, ang1), ("slp", slp, slp1), ("sca", sca, sca1))}
            return ok
        res = grp.run(rank_main)
        assert grp.transport == "peer"
    for r, ok in enumerate(res):
        assert all(ok.values()), f"strip {r} of {size} differs from the one-strip run: {ok}"


# ---- BASELINE.json configs[4] (DinfDecayAccum -wg -o) at its own strip size, and eight strips of 65536 columns on the final engine ----
def _decay_job(ctx, nx, ny, seed=1234):
    import torch

    import bench
    import taudem_amd as T

    return bench.DecayStrip(torch, ctx, None, nx, ny, 0, ny, seed, T)


def test_decay_config5_strip(ctx, oracle, monkeypatch):
    """configs[4] as one GPU of the 8-GPU run sees it - a 65536 x 8192 strip, weights, decay multipliers, 64 outlets - under the sweep
    verifier, then EVERY cell against dmarea()'s loop body on the host (oracle.dinfdecayaccum_check, src/dinfdecayaccum.cpp:204-291): evaluated
    cells follow from their contributors' final values bit for bit (by induction over the dependency order that IS the reference's raster),
    and a cell holds a value iff an outlet is reachable downstream of it (the restatement's own search of the closure, src/commonLib.cpp:285-385)."""
    if _host_gb() < 24:
        pytest.skip("needs ~12 GB of host memory for the linear-time check")
    import torch

    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")
    nx, ny = 65536, 8192
    job = _decay_job(ctx, nx, ny)
    st = job.step()
    torch.cuda.synchronize()
    ox, oy = job.outlets
    assert len(ox) == 64
    outl = (np.array(ox, dtype=np.int32), np.array(oy, dtype=np.int32) - 1)      # strip-array rows -> raster rows
    ang, dm, w, out = (t[1:ny + 1].cpu().numpy() for t in (job.ang, job.dm, job.w, job.out))
    evaluated = int((out != np.float32(-3.402823466e38)).sum())
    # (DecayStrip.step() passes the call's default cell size 1 x 1: square cells, and with -wg the cell size only enters through prop())
    bad, first, queued = oracle.dinfdecayaccum_check(ang, dm, out, dx=1.0, dy=1.0, weights=w, contcheck=True, outlets=outl)
    assert bad == 0, f"{bad} cells do not follow from dmarea()'s expression / closure; first at row {first // nx} column {first % nx}"
    assert 0 < evaluated <= queued, (evaluated, queued)
    assert evaluated == job.evaluated_cells(torch)
    # -nc: every cell of the closure ends with a value, so the closure itself is pinned cell by cell
    _, _ = job.pipe.dinfdecayaccum(job.ang, job.dm, weights=job.w, outlets=job.outlets, contcheck=False, out=job.out)
    torch.cuda.synchronize()
    out = job.out[1:ny + 1].cpu().numpy()
    bad, first, queued_nc = oracle.dinfdecayaccum_check(ang, dm, out, dx=1.0, dy=1.0, weights=w, contcheck=False, outlets=outl)
    assert bad == 0, f"-nc: {bad} cells differ; first at row {first // nx} column {first % nx}"
    assert queued_nc == queued == int((out != np.float32(-3.402823466e38)).sum())
    assert st["rounds"] > 0


def test_eight_strips_equal_one_at_65536_columns(ctx, monkeypatch):
    """65536 columns x 16384 rows as EIGHT row strips (in-process rank group: eight contexts and rank threads, peer transport on a one-GPU box)
    against the one-strip run: fel, p, sd8, ad8 of the D8 pipeline and ang / dsca of DinfDecayAccum -wg -o (64 outlets, found per strip) are
    bit-equal - the strip protocol of configs[3] / [4] (full-width rows, eight ranks, src/linearpart.h:133-134,194-219) on the engine that is
    benchmarked, every sweep under the verifier."""
    import torch

    import bench
    import taudem_amd as T
    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows

    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")
    nx, ny, size, seed = 65536, 16384, 8, 1234
    wl = T.synth_base_wavelength(max(nx, ny))
    # ---- one strip: the plain single-GPU entry points ----
    dem = ctx.synth_dem((ny, nx), seed=seed, base_wavelength=wl)
    fel1 = ctx.pitremove(dem, -9999.0)
    del dem
    p1, sd81, st1 = ctx.d8flowdir(fel1, -3.0e38, 30.0, 30.0, stats=True)
    ad81 = ctx.aread8(p1, -32768, contcheck=False)
    one = bench.DecayStrip(torch, ctx, None, nx, ny, 0, ny, seed, T)
    one.step()
    torch.cuda.synchronize()
    ang1, dsca1 = one.ang[1:ny + 1], one.out[1:ny + 1]
    outlets1 = sorted(zip(one.outlets[0], [r - 1 for r in one.outlets[1]]))
    parts = partition_rows(ny, size)
    with StripGroup(size, nx) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            nyl = y1 - y0
            pipe = StripPipeline(c, comm, nx, nyl)
            d = pipe.empty(torch.float32)
            c.synth_dem((nyl, nx), seed=seed, x0=0, y0=y0, base_wavelength=wl, out=d[1:nyl + 1])
            fel, _ = pipe.pitremove(d, -9999.0)
            del d
            p, sd8, s2 = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0)
            ad8, _ = pipe.aread8(p, -32768, contcheck=False)
            torch.cuda.synchronize()
            ok = {k: bool(torch.equal(a[1:nyl + 1].view(torch.int16 if a.dtype == torch.int16 else torch.int32),
                                      b[y0:y1].view(torch.int16 if b.dtype == torch.int16 else torch.int32)))
                  for k, a, b in (("fel", fel, fel1), ("p", p, p1), ("sd8", sd8, sd81), ("ad8", ad8, ad81))}
            del fel, p, sd8, ad8
            job = bench.DecayStrip(torch, c, comm, nx, ny, y0, nyl, seed, T)
            job.step()
            torch.cuda.synchronize()
            ok["ang"] = bool(torch.equal(job.ang[1:nyl + 1].view(torch.int32), ang1[y0:y1].view(torch.int32)))
            ok["dsca"] = bool(torch.equal(job.out[1:nyl + 1].view(torch.int32), dsca1[y0:y1].view(torch.int32)))
            return ok, [(x, y0 + row - 1) for x, row in zip(*job.outlets)], (s2["levels_fall_max"], s2["levels_rise_max"])
        res = grp.run(rank_main)
        assert grp.transport == "peer"
    for r, (ok, _, _) in enumerate(res):
        assert all(ok.values()), f"strip {r} of {size} differs from the one-strip run: {ok}"
    assert sorted(o for _, outl, _ in res for o in outl) == outlets1 and len(outlets1) == 64
    assert all(lv == (st1["levels_fall_max"], st1["levels_rise_max"]) for _, _, lv in res), "the level statistics are global"
    assert 0 < st1["levels_fall_max"] < 32766


def test_d8_config4_strip_vs_restatement(ctx, oracle, monkeypatch):
    """BASELINE.json configs[3] as one GPU of the 8-GPU run sees it - the pipeline on a 65536 x 8192 strip of the 65536^2 DEM - against the restatement
    on the host: `p` and `sd8` of D8FlowDir on every cell (the restatement with its flat loops as breadth-first searches: linear time, pinned to the
    real reference on CPU; the first pass and setFlow2 on the host threads), `ad8` of AreaD8 on every cell through aread8()'s loop body
    (orc_aread8_check), `fel` of PitRemove on every cell through flood()'s certificate (orc_pitremove_check: per-cell fixed-point equation + the flood from the seed cells)."""
    if _host_gb() < 48:
        pytest.skip("needs ~30 GB of host memory")
    import os

    import torch

    import taudem_amd as T

    nx, ny = 65536, 8192
    dem = ctx.synth_dem((ny, nx), seed=1234, base_wavelength=T.synth_base_wavelength(65536))
    fel = ctx.pitremove(dem, -9999.0)
    assert bool((fel >= dem).all()) and torch.equal(ctx.pitremove(fel, -9999.0), fel)
    _certify_fel(oracle, dem, fel, "65536 x 8192 strip")
    del dem
    p, sd8, st = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    ad8 = ctx.aread8(p, -32768)
    fel_h = fel.cpu().numpy()
    del fel
    monkeypatch.setenv("ORC_FLATS", "bfs")
    oracle.set_threads(os.cpu_count() or 1)
    try:
        p_o, sd8_o, st_o = oracle.d8flowdir(fel_h, -3.0e38, 30.0, 30.0)
    finally:
        oracle.set_threads(1)
    del fel_h
    assert (st_o["flats_initial"], st_o["flat_iterations"], st_o["flats_left"]) == (st["flats_initial"], st["flat_iterations"], st["flats_left"])
    p_h = p.cpu().numpy()
    neq = int(np.count_nonzero(p_h != p_o))
    assert neq == 0, f"p: {neq} cells differ from the restatement"
    assert np.array_equal(sd8.cpu().numpy().view(np.uint32), sd8_o.view(np.uint32)), "sd8 differs from the restatement"
    del p_o, sd8_o, sd8
    ad8_h = ad8.cpu().numpy()
    bad, first, queued = oracle.aread8_check(p_h, ad8_h, -32768, contcheck=True)
    assert bad == 0, f"{bad} cells of ad8 do not follow from aread8()'s expression; first at row {first // nx} column {first % nx}"
    assert queued == int(((p_h >= 0) & (p_h <= 8)).sum()) and float(ad8_h.max()) > 1e7


# ---- BASELINE.json configs[3] and configs[4] THEMSELVES: 65536 x 65536 in eight strips of 65536 x 8192, every cell on the host -------------------------
# The regimes that exist only at full height (630 542 cells above 2^24 folded tree by tree with blocked contributors across strips, ad8 up to 2.75e9,
# flat levels up to 32 002 of the int16 range's 32 766, a fourth flat iteration, coarse PitRemove levels relaxed across eight strips) are checked on EVERY
# cell: the eight strips' owned rows are assembled on the host (the MI355X boxes have > 2 TB of host memory) and the WHOLE rasters go through the same
# linear-time certificates as the lone strips above - nothing strip-local is assumed by the checkers, so a defect of the strip protocol (a halo row that
# arrived late, a vote taken too early) shows up as a cell that does not follow from its neighbours in the other strip.
def _eight_strips_to_host(nx, ny, size, rank_job, names_dtypes):
    """Runs rank_job(r, c, comm, y0, nyl) -> ({name: strip tensor with halo rows}, extra) on an in-process group of `size` strips; returns the whole rasters on
    the host (numpy, owned rows of every strip in place) and the per-rank extras."""
    import torch

    from taudem_amd.distributed import StripGroup, partition_rows

    host = {k: np.empty((ny, nx), dtype=dt) for k, dt in names_dtypes.items()}
    parts = partition_rows(ny, size)
    with StripGroup(size, nx) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            nyl = y1 - y0
            tensors, extra = rank_job(r, c, comm, y0, nyl)
            torch.cuda.synchronize()
            for k, t in tensors.items():
                torch.from_numpy(host[k][y0:y1]).copy_(t[1:nyl + 1])
            return extra
        extras = grp.run(rank_main)
        assert grp.transport == "peer"
    torch.cuda.empty_cache()
    return host, extras


def test_config4_full_height_every_cell(ctx, oracle, monkeypatch):
    """BASELINE.json configs[3] itself - PitRemove -> D8FlowDir -> AreaD8 on the 65536 x 65536 synthetic DEM, row-partitioned into EIGHT strips of 65536 x 8192
    (src/linearpart.h:133-134; eight rank threads and contexts on the library's rank group, every halo exchange, vote and cross-strip dependency of the 8-GPU
    protocol) - with every one of the 4 294 967 296 cells of fel, p, sd8 and ad8 checked on the host against the restatement: fel through flood()'s certificate
    (orc_pitremove_check: src/flood.cpp:243-271,292-331 per cell + the flood from the seed cells), p and sd8 against the restatement's own rasters
    (src/d8.cpp:359-409,459-680; flat loops as breadth-first searches: linear time, pinned to the real tools on CPU), ad8 through aread8()'s loop body on every
    cell (orc_aread8_check: src/aread8.cpp:231-256, k-ordered float32 adds - all cells above 2^24 included)."""
    if _host_gb() < 400:
        pytest.skip("needs ~300 GB of host memory (the whole 65536 x 65536 rasters and the restatement's work arrays)")
    import os
    import time

    import torch

    import taudem_amd as T
    from taudem_amd.distributed import StripPipeline

    ctx.release_scratch()      # (the session context's arena - sized by the 32768^2 tests - goes first: eight more contexts share this GPU)
    torch.cuda.empty_cache()
    nx = ny = 65536
    size, seed = 8, 1234
    wl = T.synth_base_wavelength(nx)
    t0 = time.time()

    def rank_job(r, c, comm, y0, nyl):
        pipe = StripPipeline(c, comm, nx, nyl)
        dem = pipe.empty(torch.float32)
        c.synth_dem((nyl, nx), seed=seed, x0=0, y0=y0, base_wavelength=wl, out=dem[1:nyl + 1])
        fel, _ = pipe.pitremove(dem, -9999.0)
        p, sd8, s2 = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0)
        ad8, _ = pipe.aread8(p, -32768, contcheck=False)   # (no edge contamination: every cell with a direction gets its count, the main stems reach 2.75e9)
        return {"dem": dem, "fel": fel, "p": p, "sd8": sd8, "ad8": ad8}, s2

    h, extras = _eight_strips_to_host(nx, ny, size, rank_job, {"dem": np.float32, "fel": np.float32, "p": np.int16, "sd8": np.float32, "ad8": np.float32})
    st = extras[0]
    t1 = time.time()
    # ---- fel: every cell ----
    _certify_fel(oracle, h["dem"], h["fel"], "65536 x 65536 in eight strips")
    del h["dem"]
    t2 = time.time()
    # ---- ad8: every cell from its contributors (before p is compared: the two checks are independent) ----
    bad, first, queued = oracle.aread8_check(h["p"], h["ad8"], -32768, contcheck=False)
    assert bad == 0, f"{bad} cells of ad8 do not follow from aread8()'s expression; first at row {first // nx} column {first % nx}"
    big = int(np.count_nonzero(h["ad8"] > np.float32(2 ** 24)))
    amax = float(h["ad8"].max())
    assert queued == int(np.count_nonzero((h["p"] >= 0) & (h["p"] <= 8)))
    assert big > 500000 and amax > 2.0 ** 31, (big, amax)      # the regime this test exists for
    del h["ad8"]
    t3 = time.time()
    # ---- p, sd8: every cell against the restatement's rasters ----
    monkeypatch.setenv("ORC_FLATS", "bfs")
    oracle.set_threads(os.cpu_count() or 1)
    try:
        p_o, sd8_o, st_o = oracle.d8flowdir(h["fel"], -3.0e38, 30.0, 30.0)
    finally:
        oracle.set_threads(1)
    assert (st_o["flats_initial"], st_o["flat_iterations"], st_o["flats_left"]) == (st["flats_initial"], st["flat_iterations"], st["flats_left"])
    neq = int(np.count_nonzero(h["p"] != p_o))
    assert neq == 0, f"p: {neq} cells differ from the restatement"
    assert np.array_equal(h["sd8"].view(np.uint32), sd8_o.view(np.uint32)), "sd8 differs from the restatement"
    assert all(e["levels_fall_max"] == st["levels_fall_max"] for e in extras)
    assert 30000 < st["levels_fall_max"] < 32766 and st["flat_iterations"] >= 4, (st["levels_fall_max"], st["flat_iterations"])
    print(f"\nconfigs[3] at full height: {queued} directed cells, {big} cells above 2^24, ad8 max {amax:.4g}, deepest level {st['levels_fall_max']}, "
          f"{st['flat_iterations']} flat iterations; GPU + download {t1 - t0:.0f} s, fel {t2 - t1:.0f} s, ad8 {t3 - t2:.0f} s, p / sd8 {time.time() - t3:.0f} s")


def test_config5_full_height_every_cell(ctx, oracle, monkeypatch):
    """BASELINE.json configs[4] itself - DinfDecayAccum with weights, decay multipliers and 64 outlets on the 65536 x 65536 DEM in EIGHT strips of 65536 x 8192,
    under the sweep verifier - with every cell of dsca checked on the host: evaluated cells follow from their contributors' final values through dmarea()'s loop
    body bit for bit (src/dinfdecayaccum.cpp:204-291), and a cell holds a value iff an outlet is reachable downstream of it (the restatement's own search of the
    closure across the whole raster, src/commonLib.cpp:285-385).  The angles the strips computed (DinfFlowDir across eight strips) are the checker's input."""
    if _host_gb() < 400:
        pytest.skip("needs ~150 GB of host memory")
    import time

    import torch

    import bench
    import taudem_amd as T

    ctx.release_scratch()
    torch.cuda.empty_cache()
    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")
    nx = ny = 65536
    size, seed = 8, 1234
    t0 = time.time()

    def rank_job(r, c, comm, y0, nyl):
        job = bench.DecayStrip(torch, c, comm, nx, ny, y0, nyl, seed, T)
        st = job.step()
        torch.cuda.synchronize()
        return {"ang": job.ang, "dm": job.dm, "w": job.w, "dsca": job.out}, ([(x, y0 + row - 1) for x, row in zip(*job.outlets)], job.evaluated_cells(torch), st["rounds"])

    h, extras = _eight_strips_to_host(nx, ny, size, rank_job, {"ang": np.float32, "dm": np.float32, "w": np.float32, "dsca": np.float32})
    t1 = time.time()
    outl = sorted(o for e in extras for o in e[0])
    assert len(outl) == 64
    ox, oy = np.array([o[0] for o in outl], dtype=np.int32), np.array([o[1] for o in outl], dtype=np.int32)
    evaluated = int(np.count_nonzero(h["dsca"] != np.float32(-3.402823466e38)))
    assert evaluated == sum(e[1] for e in extras)
    bad, first, queued = oracle.dinfdecayaccum_check(h["ang"], h["dm"], h["dsca"], dx=1.0, dy=1.0, weights=h["w"], contcheck=True, outlets=(ox, oy))
    assert bad == 0, f"{bad} cells do not follow from dmarea()'s expression / closure; first at row {first // nx} column {first % nx}"
    assert 0 < evaluated <= queued, (evaluated, queued)
    print(f"\nconfigs[4] at full height: {queued} cells in the 64 outlets' closure, {evaluated} evaluated (the rest contaminated); GPU + download {t1 - t0:.0f} s, host check {time.time() - t1:.0f} s")
