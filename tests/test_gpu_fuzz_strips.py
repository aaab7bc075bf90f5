"""Seeded fuzz of the strip protocol (round 5: coarse levels across strips, tree-wise big-cell folds on pending lists, bounded sweep / closure rounds between
two exchanges, both level fields side by side): random raster shapes, rank counts (2 ... 8, strips shorter than a tile included), nodata holes, weights,
outlets and a lowered big-cell threshold - every raster of every tool must equal the pinned restatement's, bit for bit, whatever the cut."""
import os

import numpy as np
import pytest

from conftest import bits_equal, describe_diff

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ANG_ND = -3.402823466e38


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    scale = int(os.environ.get("TDX_FUZZ_SCALE", "1"))
    ny = int(rng.integers(40, 700 * scale))
    nx = int(rng.integers(40, 900 * scale))
    world = int(rng.integers(2, 9))
    world = max(2, min(world, ny // 3))
    holes = int(rng.integers(0, 4))
    thr = int(rng.choice([3, 8, 40, 1 << 24]))
    eager = int(rng.choice([1, 2, 8]))
    return ny, nx, world, holes, thr, eager, rng


# (TDX_FUZZ_SEEDS=n: a longer one-off sweep, e.g. after a change of the strip protocol; TDX_FUZZ_SCALE=k: rasters up to k x larger on each side)
@pytest.mark.parametrize("seed", range(int(os.environ.get("TDX_FUZZ_SEEDS", "32"))))
def test_random_cut_random_raster(seed, oracle, monkeypatch):
    import torch

    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows

    ny, nx, world, holes, thr, eager, rng = _case(seed)
    dem = oracle.synth_dem((ny, nx), 500 + seed)
    for _ in range(holes):
        y0, x0 = int(rng.integers(0, ny - 5)), int(rng.integers(0, nx - 5))
        dem[y0:y0 + int(rng.integers(2, ny // 3 + 3)), x0:x0 + int(rng.integers(2, nx // 3 + 3))] = -9999.0
    w = rng.random((ny, nx), dtype=np.float32) + np.float32(0.25)
    dm = (rng.random((ny, nx), dtype=np.float32) * np.float32(0.1) + np.float32(0.9)).astype(np.float32)
    fel_o = oracle.pitremove(dem, -9999.0)
    p_o, sd8_o, _ = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 25.0)
    ang_o, slp_o, _ = oracle.dinfflowdir(fel_o, -3.0e38, 30.0, 25.0)
    a_o = oracle.aread8(p_o, -32768, contcheck=False)
    aw_o = oracle.aread8(p_o, -32768, weights=w, contcheck=True)
    sca_o = oracle.areadinf(ang_o, ANG_ND, 30.0, 25.0, contcheck=False)
    # outlets: a handful of cells with large D8 area (inside the raster, wherever they fall relative to the cut)
    flat = np.argsort(a_o, axis=None)[-400:]
    pick = rng.choice(flat, size=5, replace=False)
    oy, ox = np.unravel_index(pick, a_o.shape)
    outl = (ox.astype(np.int32), oy.astype(np.int32))
    ao_o = oracle.aread8(p_o, -32768, contcheck=False, outlets=outl)
    d_o = oracle.dinfdecayaccum(ang_o, dm, dx=30.0, dy=25.0, weights=w, contcheck=False, outlets=outl)
    # the reverse sweeps (round 6: a tile routine of their own): a disturbance grid of scattered cells, the weights as the accumulated quantity
    dg = (rng.random((ny, nx)) < 0.02).astype(np.int32)
    dep_o = oracle.dinfupdependence(ang_o, dg, dx=30.0, dy=25.0)
    racc_o, dmax_o = oracle.dinfrevaccum(ang_o, w, dx=30.0, dy=25.0)
    monkeypatch.setenv("TDX_AD8_BIG_THRESHOLD", str(thr))
    monkeypatch.setenv("TDX_SWEEP_EAGER_ROUNDS", str(eager))
    monkeypatch.setenv("TDX_REACH_EAGER_ROUNDS", str(eager))
    monkeypatch.setenv("TDX_SWEEP_VERIFY", "1")
    parts = partition_rows(ny, world)
    with StripGroup(world, nx, [0] * world) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            nyl = y1 - y0
            pipe = StripPipeline(c, comm, nx, nyl)
            sl = slice(1, nyl + 1)

            def put(a, dt):
                t = pipe.empty(dt)
                t[sl] = torch.from_numpy(np.ascontiguousarray(a[y0:y1])).cuda()
                return t
            d = put(dem, torch.float32)
            fel, _ = pipe.pitremove(d, -9999.0)
            p, sd8, _ = pipe.d8flowdir(fel, -3.0e38, 30.0, 25.0)
            ang, slp, _ = pipe.dinfflowdir(fel, -3.0e38, 30.0, 25.0)
            wt, dmt = put(w, torch.float32), put(dm, torch.float32)
            a, _ = pipe.aread8(p, -32768, contcheck=False)
            aw, _ = pipe.aread8(p, -32768, weights=wt, contcheck=True)
            sca, _ = pipe.areadinf(ang, ANG_ND, 30.0, 25.0, contcheck=False)
            lo = pipe.local_outlets(outl[0], outl[1], y0)
            ao, _ = pipe.aread8(p, -32768, contcheck=False, outlets=lo)
            dd, _ = pipe.dinfdecayaccum(ang, dmt, dx=30.0, dy=25.0, weights=wt, contcheck=False, outlets=lo)
            dep, _ = pipe.dinfupdependence(ang, put(dg, torch.int32), dx=30.0, dy=25.0)
            racc, dmax, _ = pipe.dinfrevaccum(ang, wt, dx=30.0, dy=25.0)
            return {k: v[sl].cpu().numpy() for k, v in (("fel", fel), ("p", p), ("sd8", sd8), ("ang", ang), ("slp", slp), ("ad8", a), ("ad8_w", aw), ("sca", sca),
                                                        ("ad8_o", ao), ("dsca_o", dd), ("dep", dep), ("racc", racc), ("dmax", dmax))}
        res = grp.run(rank_main)
    what = f"seed {seed}: {ny} x {nx} in {world} strips, {holes} holes, big-cell threshold {thr}, {eager} rounds between exchanges"
    for key, ref in (("fel", fel_o), ("p", p_o), ("sd8", sd8_o), ("ang", ang_o), ("slp", slp_o), ("ad8", a_o), ("ad8_w", aw_o), ("sca", sca_o), ("ad8_o", ao_o),
                     ("dsca_o", d_o), ("dep", dep_o), ("racc", racc_o), ("dmax", dmax_o)):
        got = np.concatenate([r[key] for r in res], axis=0)
        assert bits_equal(got, ref), describe_diff(got, ref, f"{what}: {key}")
