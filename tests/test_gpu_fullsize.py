"""Full-size (BASELINE.json configs[1], 16384 x 16384) checks through size-independent properties - the CPU oracle
would need hours for flat resolution at this size (O(N^1.5)); the same properties are checked against the oracle at
small sizes first, so that a property failure at full size means a size-dependent defect."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def _props(ctx, n, seed, monkeypatch, oracle=None):
    import torch

    dem = ctx.synth_dem(n, seed=seed)
    fel = ctx.pitremove(dem, -9999.0)
    # (1) the filled surface is a fixed point of the fill and never below the input (src/flood.cpp:307-330)
    assert bool((fel >= dem).all())
    assert torch.equal(ctx.pitremove(fel, -9999.0), fel), "pitremove is not idempotent"
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
    _props(ctx, 700, 3, monkeypatch, oracle)


def test_properties_at_16384(ctx, monkeypatch):
    amax = _props(ctx, 16384, 1234, monkeypatch)
    assert amax > 2 ** 24      # the exact re-evaluation path was exercised


def _dinf_props(ctx, n, seed, monkeypatch):
    """D-infinity at full size: angles in range, every interior cell resolved, the tile dependency sweep and the atomic pull walk
    (two independent schedules) agree bit for bit on sca, and flow is conserved at the raster's rim within float32 rounding."""
    import math

    import torch

    dem = ctx.synth_dem(n, seed=seed)
    fel = ctx.pitremove(dem, -9999.0)
    del dem
    ang, slp, st = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
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
    assert bool((sca_t[1:-1, 1:-1] >= 30.0).all()), "an interior cell was never evaluated"
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


def test_dinf_properties_small(ctx, monkeypatch):
    _dinf_props(ctx, 700, 3, monkeypatch)


def test_dinf_properties_at_16384(ctx, monkeypatch):
    _dinf_props(ctx, 16384, 1234, monkeypatch)
