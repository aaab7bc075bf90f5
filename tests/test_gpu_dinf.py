"""GPU parity of the D-infinity path (DinfFlowDir -> AreaDinf -> DinfDecayAccum) through the C ABI.

Tolerances (BASELINE.json north_star): slope, facet choice and flat handling are exact; the angle may
differ by one float32 ulp where the device atan2 and glibc atan2 round a double differently; areas are
compared bit-exactly when fed the reference's own angles and within 1e-6 relative end to end."""
import numpy as np
import pytest

from conftest import bits_equal, describe_diff, golden_cases, load_golden, outlets_to_indices

pytestmark = pytest.mark.gpu
CASES = golden_cases()
ANG_ND = np.float32(-3.402823466e38)


@pytest.fixture(scope="module", params=CASES)
def g(request):
    return load_golden(request.param)


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def check_angles(ang, ang_ref, what):
    special = (ang_ref == ANG_ND) | (ang_ref == -1.0)
    assert np.array_equal(ang[special].view(np.uint32), ang_ref[special].view(np.uint32)), f"{what}: nodata/flat pattern differs"
    assert np.array_equal((ang == ANG_ND) | (ang == -1.0), special), f"{what}: nodata/flat pattern differs"
    d = ulp_diff(ang[~special], ang_ref[~special])
    assert d.max(initial=0) <= 1, f"{what}: angle differs by {d.max()} ulp"
    return int((d > 0).sum())


def test_golden_dinfflowdir(g, ctx):
    ang, slp, st = ctx.dinfflowdir(np.ascontiguousarray(g["fel"]), -3.0e38, g["dxc"], g["dyc"], stats=True)
    assert bits_equal(slp, g["slp"]), describe_diff(slp, g["slp"], "slp")
    nd = check_angles(ang, g["ang"], "ang")
    assert nd <= max(2, ang.size // 10000), f"{nd} angles differ by 1 ulp"


@pytest.mark.parametrize("key,kw", [("sca", {}), ("sca_nc", {"contcheck": False}), ("sca_w_nc", {"w": True, "contcheck": False}),
                                    ("sca_outlets_nc", {"o": True, "contcheck": False})])
def test_golden_areadinf(g, ctx, key, kw):
    s = ctx.areadinf(np.ascontiguousarray(g["ang"]), float(ANG_ND), g["dxc"], g["dyc"], weights=np.ascontiguousarray(g["w"]) if kw.get("w") else None,
                     contcheck=kw.get("contcheck", True), outlets=outlets_to_indices(g) if kw.get("o") else None)
    assert bits_equal(s, g[key]), describe_diff(s, g[key], key)


@pytest.mark.parametrize("key,kw", [("dsca", {}), ("dsca_w_nc", {"w": True, "contcheck": False}), ("dsca_outlets_nc", {"o": True, "contcheck": False})])
def test_golden_dinfdecayaccum(g, ctx, key, kw):
    s = ctx.dinfdecayaccum(np.ascontiguousarray(g["ang"]), np.ascontiguousarray(g["dm"]), float(ANG_ND), -9999.0, g["dxc"], g["dyc"],
                           weights=np.ascontiguousarray(g["w"]) if kw.get("w") else None, contcheck=kw.get("contcheck", True),
                           outlets=outlets_to_indices(g) if kw.get("o") else None)
    assert bits_equal(s, g[key]), describe_diff(s, g[key], key)


@pytest.mark.parametrize("shape,seed", [((64, 64), 1), ((3, 3), 2), ((257, 301), 5), ((1000, 777), 7)])
def test_dinf_pipeline_vs_oracle(shape, seed, ctx, oracle):
    dem = oracle.synth_dem(shape, seed)
    fel = oracle.pitremove(dem, -9999.0)
    ang_o, slp_o, st_o = oracle.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    ang, slp, st = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    assert st["flats_initial"] == st_o["flats_initial"] and st["flats_left"] == st_o["flats_left"]
    assert bits_equal(slp, slp_o), describe_diff(slp, slp_o, "slp")
    check_angles(ang, ang_o, "ang")
    # areas from the ORACLE's angles: bit-exact
    for cc in (True, False):
        s_o = oracle.areadinf(ang_o, float(ANG_ND), 30.0, 30.0, contcheck=cc)
        s = ctx.areadinf(ang_o, float(ANG_ND), 30.0, 30.0, contcheck=cc)
        assert bits_equal(s, s_o), describe_diff(s, s_o, f"sca contcheck={cc}")
    rng = np.random.default_rng(seed)
    dm = (0.9 + 0.1 * rng.random(shape, dtype=np.float32)).astype(np.float32)
    d_o = oracle.dinfdecayaccum(ang_o, dm, float(ANG_ND), -9999.0, 30.0, 30.0, contcheck=False)
    d = ctx.dinfdecayaccum(ang_o, dm, float(ANG_ND), -9999.0, 30.0, 30.0, contcheck=False)
    assert bits_equal(d, d_o), describe_diff(d, d_o, "dsca")
    # end to end (device angles -> device areas): 1e-6 relative
    s_o = oracle.areadinf(ang_o, float(ANG_ND), 30.0, 30.0, contcheck=False)
    s = ctx.areadinf(ang, float(ANG_ND), 30.0, 30.0, contcheck=False)
    ok = (s_o != -1.0)
    assert np.array_equal(s != -1.0, ok)
    rel = np.abs(s[ok].astype(np.float64) - s_o[ok]) / np.maximum(np.abs(s_o[ok]), 1e-30)
    assert rel.max(initial=0.0) <= 1e-6, f"end-to-end sca relative error {rel.max()}"


def test_dinf_repeatability(ctx, oracle):
    dem = oracle.synth_dem((500, 500), 4)
    fel = oracle.pitremove(dem, -9999.0)
    ang, _ = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    s0 = ctx.areadinf(ang, float(ANG_ND), 30.0, 30.0)
    for _ in range(3):
        assert bits_equal(ctx.areadinf(ang, float(ANG_ND), 30.0, 30.0), s0)


def test_outlet_on_a_cell_without_angle(ctx, oracle, monkeypatch):
    """An outlet placed on a cell that has no angle (here: on the rim of a nodata hole, where streams end) takes part as a pure sink, and
    its neighbours are still contaminated by it - the reference's contamination tests read the ORIGINAL angle raster (src/areadinf.cpp:196-199,
    src/DinfConcLimAccum.cpp:243, src/DinfTransLimAccum.cpp:246).  Sweep, walk and restatement agree bit for bit with and without -nc."""
    rng = np.random.default_rng(31)
    shape = (420, 380)
    dem = oracle.synth_dem(shape, 23)
    dem[150:190, 120:200] = -9999.0
    fel = oracle.pitremove(dem, -9999.0)
    ang, _, _ = oracle.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    nd = ang == ANG_ND
    # cells without angle that have a neighbour draining into them: the first few along the hole's rim, plus ordinary outlets
    cand = []
    for y in range(149, 192):
        for x in range(119, 202):
            if nd[y, x] and not nd[y - 1:y + 2, x - 1:x + 2].all():
                cand.append((x, y))
    assert len(cand) > 20
    pick = [cand[i] for i in (0, 7, 19, len(cand) // 2, len(cand) - 3)]
    ox = np.array([p[0] for p in pick] + [300, 50], dtype=np.int32)
    oy = np.array([p[1] for p in pick] + [400, 60], dtype=np.int32)
    w = rng.random(shape, dtype=np.float32)
    dm = (0.9 + 0.1 * rng.random(shape, dtype=np.float32)).astype(np.float32)
    for cc in (True, False):
        want = oracle.areadinf(ang, float(ANG_ND), 30.0, 30.0, weights=w, contcheck=cc, outlets=(ox, oy))
        got = ctx.areadinf(ang, float(ANG_ND), 30.0, 30.0, weights=w, contcheck=cc, outlets=(ox, oy))
        assert bits_equal(got, want), describe_diff(got, want, f"sca -o on cells without angle, contcheck={cc}")
        monkeypatch.setenv("TDX_DINF_WALK", "1")
        got = ctx.areadinf(ang, float(ANG_ND), 30.0, 30.0, weights=w, contcheck=cc, outlets=(ox, oy))
        monkeypatch.delenv("TDX_DINF_WALK")
        assert bits_equal(got, want), describe_diff(got, want, f"sca (walk) -o on cells without angle, contcheck={cc}")
        want = oracle.dinfdecayaccum(ang, dm, float(ANG_ND), -9999.0, 30.0, 30.0, contcheck=cc, outlets=(ox, oy))
        got = ctx.dinfdecayaccum(ang, dm, float(ANG_ND), -9999.0, 30.0, 30.0, contcheck=cc, outlets=(ox, oy))
        assert bits_equal(got, want), describe_diff(got, want, f"dsca -o on cells without angle, contcheck={cc}")
        q = (0.5 + rng.random(shape, dtype=np.float32)).astype(np.float32)
        dg = (rng.random(shape) < 0.01).astype(np.int16)
        want = oracle.dinfconclimaccum(ang, dm, dg, q, csol=1.5, dx=30.0, dy=30.0, contcheck=cc, outlets=(ox, oy))
        got = ctx.dinfconclimaccum(ang, dm, dg, q, csol=1.5, dx=30.0, dy=30.0, contcheck=cc, outlets=(ox, oy))
        assert bits_equal(got, want), describe_diff(got, want, f"ctpt -o on cells without angle, contcheck={cc}")


def test_outlets_on_tile_rims(ctx, oracle):
    """Outlets on the first / last row or column of a 64 x 64 tile whose catchment lies in the NEIGHBOURING tile: the seed of the upstream closure must
    activate that tile too (the relaxation only reports rim cells that move).  Round 4's host check of configs[4] found whole catchments missing
    behind such outlets; here every accumulation tool with -o is compared with the restatement, with the outlets on the largest streams that
    cross a tile rim."""
    shape = (400, 460)
    dem = oracle.synth_dem(shape, 41)
    fel = oracle.pitremove(dem, -9999.0)
    ang, _, _ = oracle.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    p, _, _ = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    sca = oracle.areadinf(ang, dx=30.0, dy=30.0, contcheck=False)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    rim = ((yy % 64 == 0) | (yy % 64 == 63) | (xx % 64 == 0) | (xx % 64 == 63)) & (yy > 2) & (yy < shape[0] - 3) & (xx > 2) & (xx < shape[1] - 3)
    order = np.argsort(np.where(rim, sca, -1.0), axis=None)[::-1][:48]
    ox, oy = (order % shape[1]).astype(np.int32), (order // shape[1]).astype(np.int32)
    rng = np.random.default_rng(3)
    w = rng.random(shape, dtype=np.float32)
    dm = (0.9 + 0.1 * rng.random(shape, dtype=np.float32)).astype(np.float32)
    for cc in (True, False):
        want = oracle.areadinf(ang, float(ANG_ND), 30.0, 30.0, weights=w, contcheck=cc, outlets=(ox, oy))
        got = ctx.areadinf(ang, float(ANG_ND), 30.0, 30.0, weights=w, contcheck=cc, outlets=(ox, oy))
        assert bits_equal(got, want), describe_diff(got, want, f"sca -o on tile rims, contcheck={cc}")
        want = oracle.dinfdecayaccum(ang, dm, float(ANG_ND), -9999.0, 30.0, 30.0, weights=w, contcheck=cc, outlets=(ox, oy))
        got = ctx.dinfdecayaccum(ang, dm, float(ANG_ND), -9999.0, 30.0, 30.0, weights=w, contcheck=cc, outlets=(ox, oy))
        assert bits_equal(got, want), describe_diff(got, want, f"dsca -o on tile rims, contcheck={cc}")
        want = oracle.aread8(p, -32768, contcheck=cc, outlets=(ox, oy))
        got = ctx.aread8(p, -32768, contcheck=cc, outlets=(ox, oy))
        assert bits_equal(got, want), describe_diff(got, want, f"ad8 -o on tile rims, contcheck={cc}")
        want = oracle.aread8(p, -32768, weights=w, contcheck=cc, outlets=(ox, oy))
        got = ctx.aread8(p, -32768, weights=w, contcheck=cc, outlets=(ox, oy))
        assert bits_equal(got, want), describe_diff(got, want, f"ad8 -wg -o on tile rims, contcheck={cc}")
    for a, b, name in zip(ctx.gridnet(p, -32768, 30.0, 30.0, outlets=(ox, oy)), oracle.gridnet(p, -32768, 30.0, 30.0, outlets=(ox, oy)), ("plen", "tlen", "gord")):
        assert bits_equal(a, b), describe_diff(a, b, f"gridnet -o on tile rims: {name}")


def test_ramps_along_the_facet_diagonals(ctx, oracle):
    """Planar ramps whose gradient points exactly along a facet's diagonal edge with dx != dy: S2 / S1 == D2 / D1, so VSLOPE's branch `A > AD`
    (src/dinf.cpp:299-311) compares two equal angles.  The device decides that branch by the cross product S2 * D1 vs S1 * D2 and, inside a 1e-9
    band around equality, by the rounded angles themselves (dinfflowdir.hip: vslope_s) - these rasters land in the band on every cell, in every
    facet orientation, for power-of-two and other scales."""
    dx, dy = 30.0, 25.0
    n = 96
    jj, ii = np.mgrid[0:n, 0:n].astype(np.float64)
    for t in (2.0 ** -10, 3.0 * 2.0 ** -12, 5.0 * 2.0 ** -9):
        for sx in (1.0, -1.0):
            for sy in (1.0, -1.0):
                # z falls by dx^2 t per column towards sx and by dy^2 t per row towards sy: the steepest descent runs along the cell diagonal
                z = (1000.0 - sx * ii * dx * dx * t - sy * jj * dy * dy * t).astype(np.float32)
                assert np.array_equal(z.astype(np.float64), 1000.0 - sx * ii * dx * dx * t - sy * jj * dy * dy * t), "the ramp must be exact in float32"
                ang_o, slp_o, _ = oracle.dinfflowdir(z, -3.0e38, dx, dy)
                ang, slp = ctx.dinfflowdir(z, -3.0e38, dx, dy)
                assert bits_equal(slp, slp_o), describe_diff(slp, slp_o, f"slp on the diagonal ramp t={t} sx={sx} sy={sy}")
                assert bits_equal(ang, ang_o), describe_diff(ang, ang_o, f"ang on the diagonal ramp t={t} sx={sx} sy={sy}")


@pytest.mark.parametrize("form", ["list", "stream", "stream+blocks"])
def test_dinf_dense_flats_streaming_classification_and_open_water(ctx, oracle, monkeypatch, form):
    """DinfFlowDir's flat resolution on a raster whose first queue is dense (a plateau of 1200 x 1100 cells with islands, an outlet channel - more than 1/16 of
    the raster): the list classification (TDX_FLATS_LIST=1, flatk::classify_kernel<DinfTraits>), the streaming classification it now shares with D8FlowDir
    (flatk::classify_stream_kernel<LV, DinfCodes>: the angles as one-hot codes, dontCross from the angles that ARE 2, 4, 6, 8 - src/dinf.cpp:58-105) without
    and with the open-water blocks, one strip and three, against the restatement (src/dinf.cpp:598-833) - and the blocks must have been used."""
    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows
    import torch

    rng = np.random.default_rng(12)
    ny, nx = 1500, 1400
    yy, xx = np.mgrid[0:ny, 0:nx]
    z = (300.0 + 0.05 * xx + 0.02 * yy + rng.random((ny, nx)) * 0.01).astype(np.float32)
    z[150:1350, 150:1250] = np.float32(100.0)
    for cy, cx, r in ((400, 500, 40), (900, 800, 70), (700, 300, 9), (1100, 1000, 25)):
        z[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = np.float32(400.0)
    z[640:660, 0:160] = np.float32(90.0) - 0.01 * np.arange(160, dtype=np.float32)[::-1]
    fel = oracle.pitremove(z, -9999.0)
    ang_o, slp_o, st_o = oracle.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    if form == "list":
        monkeypatch.setenv("TDX_FLATS_LIST", "1")
    monkeypatch.setenv("TDX_FLATS_MACRO", "8" if form == "stream+blocks" else "0")
    ang, slp, st = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    check_angles(ang, ang_o, f"ang, {form}")   # (the module's tolerance: one float32 ulp where the device atan2 and glibc's round a double differently)
    assert bits_equal(slp, slp_o)
    assert st["flats_initial"] == st_o["flats_initial"] > 1000000 and st["flats_initial"] > ny * nx // 16
    me = test_dinf_dense_flats_streaming_classification_and_open_water
    if form == "list":
        me.ang_list = ang.copy()
    elif hasattr(me, "ang_list"):
        assert bits_equal(ang, me.ang_list), describe_diff(ang, me.ang_list, f"ang, {form} against the list classification")   # the forms among themselves: every bit
    if form == "stream":
        me.rounds_plain = st["rounds"]
    elif form == "stream+blocks" and hasattr(me, "rounds_plain"):
        assert st["rounds"] < me.rounds_plain, "the blocks were not used"
    size = 3
    parts = partition_rows(ny, size)
    fel_t = torch.from_numpy(fel)
    with StripGroup(size, nx) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            pipe = StripPipeline(c, comm, nx, y1 - y0)
            f = pipe.empty(torch.float32)
            f[1:y1 - y0 + 1].copy_(fel_t[y0:y1])
            aa, ss, _ = pipe.dinfflowdir(f, -3.0e38, 30.0, 30.0)
            torch.cuda.synchronize()
            return aa[1:y1 - y0 + 1].cpu().numpy(), ss[1:y1 - y0 + 1].cpu().numpy()
        res = grp.run(rank_main)
    ang3 = np.concatenate([r[0] for r in res])
    assert bits_equal(ang3, ang), describe_diff(ang3, ang, f"ang in three strips, {form}")
    assert bits_equal(np.concatenate([r[1] for r in res]), slp_o)
