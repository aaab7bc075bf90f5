"""GPU parity of the D8 path (PitRemove -> D8FlowDir -> AreaD8) through the C ABI: bit-exact against
(1) the golden rasters written by the real reference tools and (2) the pinned CPU restatement on
seeded synthetic DEMs.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest

from conftest import bits_equal, describe_diff, golden_cases, load_golden, outlets_to_indices

pytestmark = pytest.mark.gpu
CASES = golden_cases()


@pytest.fixture(scope="module", params=CASES)
def g(request):
    return load_golden(request.param)


# ---- golden (real reference outputs) ---------------------------------------------------------
def test_golden_pitremove(g, ctx):
    mask = np.ascontiguousarray(g["mask"]) if "mask" in g else None
    fel = ctx.pitremove(np.ascontiguousarray(g["dem"]), float(g["nodata"]), mask=mask, fourway=bool(g["fourway"]))
    assert bits_equal(fel, g["fel"]), describe_diff(fel, g["fel"], "fel")


def test_golden_d8flowdir(g, ctx):
    p, sd8, st = ctx.d8flowdir(np.ascontiguousarray(g["fel"]), -3.0e38, g["dxc"], g["dyc"], stats=True)
    assert bits_equal(sd8, g["sd8"]), describe_diff(sd8, g["sd8"], "sd8")
    assert bits_equal(p, g["p"]), describe_diff(p, g["p"], "p")
    assert f"All slopes evaluated. {st['flats_initial']} flats to resolve." in str(g["d8_stderr"])


@pytest.mark.parametrize("key,kw", [("ad8", {}), ("ad8_nc", {"contcheck": False}), ("ad8_w", {"w": True}), ("ad8_w_nc", {"w": True, "contcheck": False}),
                                    ("ad8_outlets", {"o": True}), ("ad8_outlets_nc", {"o": True, "contcheck": False})])
def test_golden_aread8(g, ctx, key, kw):
    a = ctx.aread8(np.ascontiguousarray(g["p"]), -32768, weights=np.ascontiguousarray(g["w"]) if kw.get("w") else None, weights_nodata=-9999.0,
                   contcheck=kw.get("contcheck", True), outlets=outlets_to_indices(g) if kw.get("o") else None)
    assert bits_equal(a, g[key]), describe_diff(a, g[key], key)


# ---- oracle on seeded synthetic DEMs -----------------------------------------------------------
SHAPES = [((64, 64), 1), ((1, 1), 2), ((3, 3), 3), ((5, 200), 4), ((257, 301), 5), ((512, 512), 6), ((1000, 777), 7)]


@pytest.mark.parametrize("shape,seed", SHAPES)
def test_pipeline_vs_oracle(shape, seed, ctx, oracle):
    dem = oracle.synth_dem(shape, seed)
    fel_o = oracle.pitremove(dem, -9999.0)
    fel = ctx.pitremove(dem, -9999.0)
    assert bits_equal(fel, fel_o), describe_diff(fel, fel_o, "fel")
    p_o, sd8_o, st_o = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
    p, sd8, st = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    assert st["flats_initial"] == st_o["flats_initial"]
    assert bits_equal(sd8, sd8_o), describe_diff(sd8, sd8_o, "sd8")
    assert bits_equal(p, p_o), describe_diff(p, p_o, "p")
    assert st["flat_iterations"] == st_o["flat_iterations"] and st["flats_left"] == st_o["flats_left"]
    for cc in (True, False):
        a_o = oracle.aread8(p_o, -32768, contcheck=cc)
        a = ctx.aread8(p, -32768, contcheck=cc)
        assert bits_equal(a, a_o), describe_diff(a, a_o, f"ad8 contcheck={cc}")


@pytest.mark.parametrize("shape,seed", [((257, 301), 5), ((1000, 777), 7), ((96, 4100), 9)])
def test_first_flat_queue_as_a_list(shape, seed, ctx, oracle, monkeypatch):
    """A dense first flat queue never exists as a list (bit masks + streaming passes); TDX_FLATS_LIST=1 builds it anyway (flat_list_kernel: the path of
    rasters with few flats) - same directions, same counts."""
    dem = oracle.synth_dem(shape, seed)
    fel_o = oracle.pitremove(dem, -9999.0)
    p_o, sd8_o, st_o = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
    monkeypatch.setenv("TDX_FLATS_LIST", "1")
    p, sd8, st = ctx.d8flowdir(fel_o, -3.0e38, 30.0, 30.0, stats=True)
    assert st["flats_initial"] == st_o["flats_initial"] and st["flat_iterations"] == st_o["flat_iterations"]
    assert bits_equal(sd8, sd8_o), describe_diff(sd8, sd8_o, "sd8")
    assert bits_equal(p, p_o), describe_diff(p, p_o, "p")


def test_few_flats_take_the_list_kernels(ctx, oracle):
    """A raster whose flats are a few patches on a slope (well under 1/32 of the cells): the first queue is a list, setFlow2 and the statistics run on it."""
    ny, nx = 300, 420
    yy, xx = np.mgrid[0:ny, 0:nx]
    dem = (1000.0 - 0.5 * xx - 0.31 * yy + 0.2 * np.sin(xx / 7.0) * np.cos(yy / 5.0)).astype(np.float32)
    for (y0, x0, h, w) in ((20, 30, 9, 14), (100, 200, 17, 6), (180, 90, 5, 40), (250, 300, 12, 12), (60, 350, 3, 3)):
        dem[y0:y0 + h, x0:x0 + w] = dem[y0:y0 + h, x0:x0 + w].min()
    fel_o = oracle.pitremove(dem, -9999.0)
    p_o, sd8_o, st_o = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
    assert 0 < st_o["flats_initial"] < dem.size // 32
    p, sd8, st = ctx.d8flowdir(fel_o, -3.0e38, 30.0, 30.0, stats=True)
    assert st["flats_initial"] == st_o["flats_initial"] and st["flat_iterations"] == st_o["flat_iterations"] and st["flats_left"] == st_o["flats_left"]
    assert bits_equal(sd8, sd8_o), describe_diff(sd8, sd8_o, "sd8")
    assert bits_equal(p, p_o), describe_diff(p, p_o, "p")


def test_distance_table_follows_the_cell_sizes(ctx, oracle):
    """The D8 distance table stays on the device between calls with the same cell sizes (tdx_build_fact_table): other sizes, other rows, the old sizes
    again - each call matches the restatement."""
    dem = oracle.synth_dem((180, 240), 11)
    fel = oracle.pitremove(dem, -9999.0)
    rows = np.linspace(20.0, 35.0, 180)
    for dx, dy in ((30.0, 30.0), (10.0, 25.0), (30.0, 30.0), (rows, 28.0), (rows[::-1].copy(), 28.0), (10.0, 25.0)):
        p_o, sd8_o, _ = oracle.d8flowdir(fel, -3.0e38, dx, dy)
        p, sd8 = ctx.d8flowdir(fel, -3.0e38, dx, dy)
        assert bits_equal(sd8, sd8_o), describe_diff(sd8, sd8_o, "sd8")
        assert bits_equal(p, p_o), describe_diff(p, p_o, "p")
    fel2 = oracle.pitremove(oracle.synth_dem((90, 240), 12), -9999.0)   # fewer rows with the same leading cell sizes
    p_o, sd8_o, _ = oracle.d8flowdir(fel2, -3.0e38, 10.0, 25.0)
    p, sd8 = ctx.d8flowdir(fel2, -3.0e38, 10.0, 25.0)
    assert bits_equal(sd8, sd8_o) and bits_equal(p, p_o)


def test_round_schedule_count_ring_wraps(ctx, oracle, monkeypatch):
    """The per-round count ring of the tile engine wraps (tiny ring, no coarse start: many rounds) without losing the pending round."""
    dem = oracle.synth_dem((700, 900), 31)
    fel_o = oracle.pitremove(dem, -9999.0)
    p_o, sd8_o, _ = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
    monkeypatch.setenv("TDX_PIT_NO_COARSE", "1")
    for ring in ("3", "4", "7"):
        monkeypatch.setenv("TDX_RELAX_RING", ring)
        fel = ctx.pitremove(dem, -9999.0)
        assert bits_equal(fel, fel_o), describe_diff(fel, fel_o, f"fel ring={ring}")
        p, sd8 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0)
        assert bits_equal(p, p_o), describe_diff(p, p_o, f"p ring={ring}")


@pytest.mark.parametrize("env", [{"TDX_RELAX_LDS": "1"}, {"TDX_RELAX_PULL": "1"}, {"TDX_RELAX_PULL": "64"}, {"TDX_FLATS_MASKED": "1"}, {"TDX_FLATS_MASKED": "2"}, {"TDX_FLATS_FUSED": "1"}])
def test_tile_engine_variants_reach_the_same_bits(ctx, oracle, monkeypatch, env):
    """Schedule knobs of the tile engine (LDS-resident tile kernel, list entries per cursor pull, masked / plain form of the level
    operator per tile: all tiles masked, or half of them; both level fields in one launch per round) do not change results."""
    dem = oracle.synth_dem((900, 1100), 33)
    fel_o = oracle.pitremove(dem, -9999.0)
    p_o, sd8_o, _ = oracle.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fel = ctx.pitremove(dem, -9999.0)
    assert bits_equal(fel, fel_o), describe_diff(fel, fel_o, f"fel {env}")
    p, sd8 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    assert bits_equal(p, p_o), describe_diff(p, p_o, f"p {env}")
    if "TDX_FLATS_MASKED" in env or "TDX_FLATS_FUSED" in env:   # the D-infinity flats run on the same level fields
        ang_o, slp_o = oracle.dinfflowdir(fel_o, -3.0e38, 30.0, 30.0)[:2]
        ang, slp = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0)[:2]
        assert bits_equal(ang, ang_o), describe_diff(ang, ang_o, f"ang {env}")


def test_nodata_holes_and_weights(ctx, oracle):
    rng = np.random.default_rng(5)
    dem = oracle.synth_dem((300, 400), 21)
    yy, xx = np.mgrid[0:300, 0:400]
    dem[(yy - 120) ** 2 + (xx - 250) ** 2 < 900] = -9999.0
    dem[:4, :] = -9999.0
    fel_o = oracle.pitremove(dem, -9999.0)
    fel = ctx.pitremove(dem, -9999.0)
    assert bits_equal(fel, fel_o), describe_diff(fel, fel_o, "fel")
    p_o, sd8_o, _ = oracle.d8flowdir(fel_o, -3.0e38, 10.0, 20.0)
    p, sd8 = ctx.d8flowdir(fel, -3.0e38, 10.0, 20.0)
    assert bits_equal(p, p_o), describe_diff(p, p_o, "p")
    assert bits_equal(sd8, sd8_o), describe_diff(sd8, sd8_o, "sd8")
    w = (rng.random(dem.shape, dtype=np.float32) * 1.0e5).astype(np.float32)   # pushes sums far above 2^24: k-order matters
    w[rng.random(dem.shape) < 0.005] = -9999.0
    for cc in (True, False):
        a_o = oracle.aread8(p_o, -32768, weights=w, weights_nodata=-9999.0, contcheck=cc)
        a = ctx.aread8(p, -32768, weights=w, weights_nodata=-9999.0, contcheck=cc)
        assert bits_equal(a, a_o), describe_diff(a, a_o, f"weighted ad8 contcheck={cc}")
        assert a_o.max() > 2 ** 24


def test_device_resident_path_matches_host_path(ctx, oracle):
    import torch

    dem = oracle.synth_dem((700, 900), 33)
    d_dem = torch.from_numpy(dem).to("cuda:0")
    fel_h = ctx.pitremove(dem, -9999.0)
    d_fel = ctx.pitremove(d_dem, -9999.0)
    assert bits_equal(d_fel.cpu().numpy(), fel_h)
    p_h, sd8_h = ctx.d8flowdir(fel_h, -3.0e38, 30.0, 30.0)
    d_p, d_sd8 = ctx.d8flowdir(d_fel, -3.0e38, 30.0, 30.0)
    assert bits_equal(d_p.cpu().numpy(), p_h) and bits_equal(d_sd8.cpu().numpy(), sd8_h)
    a_h = ctx.aread8(p_h, -32768)
    d_a = ctx.aread8(d_p, -32768)
    assert bits_equal(d_a.cpu().numpy(), a_h)


def test_synth_dem_device_equals_host(ctx, oracle):
    for shape, seed, x0, y0 in [((128, 192), 1, 0, 0), ((64, 64), 99, 1000, 77)]:
        host = oracle.synth_dem(shape, seed, x0, y0, base_wavelength=256)
        dev = ctx.synth_dem(shape, seed, x0, y0, base_wavelength=256).cpu().numpy()
        assert bits_equal(dev, host), describe_diff(dev, host, "synth")


def test_repeatability(ctx, oracle):
    """The accumulation walk is schedule-free: repeated runs must give identical bits."""
    dem = oracle.synth_dem((600, 600), 8)
    fel = ctx.pitremove(dem, -9999.0)
    p, _ = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, want_slope=False)
    a0 = ctx.aread8(p, -32768)
    for _ in range(3):
        assert bits_equal(ctx.aread8(p, -32768), a0)


# ---- AreaD8 tile-contraction path (unweighted, no outlets) ---------------------------------------
def _p_field(oracle, shape, seed):
    dem = oracle.synth_dem(shape, seed)
    fel = oracle.pitremove(dem, -9999.0)
    p, _, _ = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    return p


@pytest.mark.parametrize("shape,seed", [((64, 64), 1), ((65, 130), 2), ((200, 333), 3), ((777, 1000), 4)])
def test_aread8_tiles_equal_walk_and_oracle(shape, seed, ctx, oracle, monkeypatch):
    p = _p_field(oracle, shape, seed)
    for cc in (True, False):
        a_o = oracle.aread8(p, -32768, contcheck=cc)
        monkeypatch.delenv("TDX_AD8_WALK", raising=False)
        a_t = ctx.aread8(p, -32768, contcheck=cc)
        monkeypatch.setenv("TDX_AD8_WALK", "1")
        a_w = ctx.aread8(p, -32768, contcheck=cc)
        monkeypatch.delenv("TDX_AD8_WALK", raising=False)
        assert bits_equal(a_t, a_o), describe_diff(a_t, a_o, f"tiles contcheck={cc}")
        assert bits_equal(a_w, a_o), describe_diff(a_w, a_o, f"walk contcheck={cc}")


def test_aread8_tiles_big_cell_reevaluation(ctx, oracle, monkeypatch):
    """Forces the exact k-ordered re-evaluation (cells above the float-exact limit) onto ordinary cells."""
    p = _p_field(oracle, (500, 700), 11)
    for thr in ("0", "5", "1000"):
        monkeypatch.setenv("TDX_AD8_BIG_THRESHOLD", thr)
        for cc in (True, False):
            a_o = oracle.aread8(p, -32768, contcheck=cc)
            a = ctx.aread8(p, -32768, contcheck=cc)
            assert bits_equal(a, a_o), describe_diff(a, a_o, f"threshold {thr} contcheck={cc}")


def test_aread8_tiles_quirks(ctx, oracle):
    """p == 0 cells (north-west quirk of initNeighborD8up), nodata holes, a two-cell cycle, flow into nodata."""
    rng = np.random.default_rng(3)
    p = _p_field(oracle, (300, 300), 12).copy()
    idx = rng.integers(5, 295, size=(40, 2))
    for y, x in idx[:20]:
        p[y, x] = 0
    for y, x in idx[20:30]:
        p[y, x] = -32768
    y, x = idx[30]
    p[y, x] = 1; p[y, x + 1] = 5          # 2-cycle: never evaluated, nor anything downstream
    for cc in (True, False):
        a_o = oracle.aread8(p, -32768, contcheck=cc)
        a = ctx.aread8(p, -32768, contcheck=cc)
        assert bits_equal(a, a_o), describe_diff(a, a_o, f"quirks contcheck={cc}")


@pytest.mark.parametrize("kmax", ["0", "2", "4", "8"])
def test_open_water_blocks_change_the_schedule_never_a_value(ctx, oracle, monkeypatch, kmax):
    """A raster with lakes wide enough for blocks of full tiles of every size (a plateau of 1200 x 1100 cells with islands, an outlet on one side, a pit
    basin without one): D8FlowDir with the open-water blocks off / up to 2, 4, 8 tiles on edge, one strip and three, against the restatement
    (src/d8.cpp:523-558,606-638: the level fields are what they are, whatever computes them) - and the blocks must have been used."""
    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows
    import torch

    rng = np.random.default_rng(11)
    ny, nx = 1500, 1400
    yy, xx = np.mgrid[0:ny, 0:nx]
    z = (300.0 + 0.05 * xx + 0.02 * yy + rng.random((ny, nx)) * 0.01).astype(np.float32)
    z[150:1350, 150:1250] = np.float32(100.0)                       # the lake floor: one big flat after filling
    for cy, cx, r in ((400, 500, 40), (900, 800, 70), (700, 300, 9), (1100, 1000, 25)):   # islands
        z[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = np.float32(400.0)
    z[640:660, 0:160] = np.float32(90.0) - 0.01 * np.arange(160, dtype=np.float32)[::-1]   # an outlet channel to the west edge
    fel = oracle.pitremove(z, -9999.0)
    p_o, sd8_o, st_o = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    monkeypatch.setenv("TDX_FLATS_MACRO", kmax)
    p, sd8, st = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    assert bits_equal(p, p_o), describe_diff(p, p_o, f"p, blocks <= {kmax}")
    assert bits_equal(sd8, sd8_o)
    assert st["flats_initial"] == st_o["flats_initial"] > 1000000 and st["levels_fall_max"] > 500
    if kmax == "0":
        test_open_water_blocks_change_the_schedule_never_a_value.rounds0 = st["rounds"]
    elif hasattr(test_open_water_blocks_change_the_schedule_never_a_value, "rounds0"):
        assert st["rounds"] < test_open_water_blocks_change_the_schedule_never_a_value.rounds0, "the blocks were not used"
    # three strips: blocks never touch the tile rows at a strip boundary; the halo exchange wakes them like any tile
    size = 3
    parts = partition_rows(ny, size)
    fel_t = torch.from_numpy(fel)
    with StripGroup(size, nx) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            pipe = StripPipeline(c, comm, nx, y1 - y0)
            f = pipe.empty(torch.float32)
            f[1:y1 - y0 + 1].copy_(fel_t[y0:y1])
            pp, ss, _ = pipe.d8flowdir(f, -3.0e38, 30.0, 30.0)
            torch.cuda.synchronize()
            return pp[1:y1 - y0 + 1].cpu().numpy(), ss[1:y1 - y0 + 1].cpu().numpy()
        res = grp.run(rank_main)
    p3 = np.concatenate([r[0] for r in res])
    assert bits_equal(p3, p_o), describe_diff(p3, p_o, f"p in three strips, blocks <= {kmax}")
    assert bits_equal(np.concatenate([r[1] for r in res]), sd8_o)


def test_aread8_direction_codes_outside_0_to_8(ctx, oracle, monkeypatch):
    """A p grid somebody else wrote may hold codes that are neither 0 .. 8 nor the nodata value (13, 14, 15, 20, 100, -5 ...): such a cell does not take part
    (initNeighborD8up, src/commonLib.cpp:257-266), is no nodata cell either and contaminates nobody - every path of the product (tile contraction, dependency
    sweep, walk) against the restatement.  (14 and 15 used to land on the one-hot bits of the outlets mode's sink and of nodata: ADVICE r05.)
    NOT covered, a documented deviation: the codes 9 .. 12 and -3 .. -1, for which the reference's contributor test `p[n] - k == +-4`
    (src/aread8.cpp:246) fires although the cell never takes part - there the reference contaminates ONE neighbour (with contamination checking on), the
    product treats the cell as inert like every other invalid code; and 16 .. 24 / 32, which the product's outlets mode uses
    internally (aread8.hip: P_OUTSIDE, P_SINK).  D8FlowDir never writes any of them."""
    rng = np.random.default_rng(17)
    p = _p_field(oracle, (300, 330), 21).copy()
    idx = rng.integers(3, 297, size=(120, 2))
    codes = [13, 14, 15, 31, 33, 100, 1000, -4, -5, -32767]
    for i, (y, x) in enumerate(idx):
        p[y, x] = codes[i % len(codes)]
    w = rng.random(p.shape, dtype=np.float32)
    for env in ({}, {"TDX_AD8_SWEEP": "1"}, {"TDX_AD8_WALK": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for cc in (True, False):
            a_o = oracle.aread8(p, -32768, contcheck=cc)
            a = ctx.aread8(p, -32768, contcheck=cc)
            assert bits_equal(a, a_o), describe_diff(a, a_o, f"codes outside 0..8, {env} contcheck={cc}")
        for k in env:
            monkeypatch.delenv(k)
    a_o = oracle.aread8(p, -32768, weights=w, contcheck=True)
    a = ctx.aread8(p, -32768, weights=w, contcheck=True)
    assert bits_equal(a, a_o), describe_diff(a, a_o, "codes outside 0..8, weighted")


def test_aread8_kahn_schedules_agree(ctx, oracle, monkeypatch):
    """The in-tile Kahn walk of ad8_tile_local_kernel as ONE loop per lane (default) and as a walk loop nested in the loop over the lane's sources
    (TDX_AD8_KAHN_NESTED=1): the same counts - the schedule is free (src/aread8.cpp:220-304) -, incl. the p == 0 quirk, holes and a cycle."""
    rng = np.random.default_rng(5)
    p = _p_field(oracle, (700, 900), 21).copy()
    idx = rng.integers(5, 690, size=(30, 2))
    for y, x in idx[:12]:
        p[y, x] = 0
    for y, x in idx[12:24]:
        p[y, x] = -32768
    y, x = idx[24]
    p[y, x] = 1; p[y, x + 1] = 5
    for cc in (True, False):
        a_o = oracle.aread8(p, -32768, contcheck=cc)
        for nested in (False, True):
            if nested:
                monkeypatch.setenv("TDX_AD8_KAHN_NESTED", "1")
            a = ctx.aread8(p, -32768, contcheck=cc)
            if nested:
                monkeypatch.delenv("TDX_AD8_KAHN_NESTED")
            assert bits_equal(a, a_o), describe_diff(a, a_o, f"kahn nested={nested} contcheck={cc}")


@pytest.mark.slow
def test_aread8_above_2_24_rounds_like_the_reference(ctx, oracle):
    """A comb-shaped direction field on 4200 x 4200 cells: every row drains east into the last interior
    column, which drains south - counts pass 2^24 on the trunk, where float32 adds round and the k
    order of src/aread8.cpp:239-256 decides the bits."""
    n = 4200
    p = np.full((n, n), 1, dtype=np.int16)
    p[:, n - 2] = 7
    p[:, n - 1] = 5
    p[0, :] = -32768; p[n - 1, :] = -32768; p[:, 0] = -32768
    a_o = oracle.aread8(p, -32768, contcheck=False)
    a = ctx.aread8(p, -32768, contcheck=False)
    assert a_o.max() > 2 ** 24
    assert bits_equal(a, a_o), describe_diff(a, a_o, "comb")


def _comb_expected_trunk(n):
    """Trunk column n-2 of the comb field below, folded like src/aread8.cpp:231-256: a = 1; a += E (=1); a += N (trunk above); a += W (= n-3)."""
    t = np.zeros(n, dtype=np.float32)
    prev = np.float32(0.0)
    w = np.float32(n - 3)
    for y in range(1, n - 1):
        a = np.float32(1.0) + np.float32(1.0)
        if y > 1:
            a = np.float32(a + prev)
        a = np.float32(a + w)
        t[y] = a
        prev = a
    return t


def _comb_check(ctx, n, oracle=None):
    import torch

    dev = f"cuda:{ctx.device}"
    p = torch.full((n, n), 1, dtype=torch.int16, device=dev)
    p[:, n - 2] = 7
    p[:, n - 1] = 5
    p[0, :] = -32768; p[n - 1, :] = -32768; p[:, 0] = -32768
    a = ctx.aread8(p, -32768, contcheck=False)
    exp = torch.arange(n, dtype=torch.float32, device=dev).repeat(n, 1)          # a(x, y) = x on the teeth
    exp[:, n - 1] = 1.0
    exp[:, n - 2] = torch.from_numpy(_comb_expected_trunk(n)).to(dev)
    exp[0, :] = -1.0; exp[n - 1, :] = -1.0; exp[:, 0] = -1.0
    if oracle is not None:
        assert bits_equal(exp.cpu().numpy(), oracle.aread8(p.cpu().numpy(), -32768, contcheck=False)), "the analytic comb differs from the oracle"
    neq = a.view(torch.int32) != exp.view(torch.int32)
    assert not bool(neq.any()), f"comb {n}: {int(neq.sum())} cells differ, first {torch.nonzero(neq)[:5].tolist()}"
    return float(a.max())


def test_comb_analytic_matches_oracle(ctx, oracle):
    assert _comb_check(ctx, 4200, oracle) > 2 ** 24


def _trunks_field(rows, lens):
    """Teeth that flow east into several trunks (south); the trunks end in the bottom interior row, which flows east and collects them one after the other."""
    nx = 1 + sum(lens) + 1
    p = np.full((rows, nx), 1, dtype=np.int16)
    x = 0
    for length in lens:
        x += length
        p[:, x] = 7
    p[rows - 2, :] = 1
    p[0, :] = -32768; p[rows - 1, :] = -32768; p[:, 0] = -32768; p[:, nx - 1] = -32768
    return p


@pytest.mark.slow
def test_aread8_interleaved_trunks_above_2_24(ctx, oracle, monkeypatch):
    """Three trunks of slightly different tooth lengths pass 2^24 side by side and join in the bottom row (3000 x 18006 cells, 12 621 of them above 2^24,
    up to 5.4e7: two powers of two are crossed): in count order the trunks' cells INTERLEAVE, so a 64-entry chunk of the big-cell fold holds several chains
    whose float32 additions round (odd addends at ulp 2 and 4: ties) - what the in-binade scan of ad8_big_fold_kernel resolves by pointer jumping.  The scan,
    the cell-after-cell loop it replaces (TDX_AD8_BIG_SCAN=0) and the one-wave fold of the whole list must all give the restatement's bits
    (src/aread8.cpp:231-256)."""
    p = _trunks_field(3000, (5990, 6003, 6011))
    a_o = oracle.aread8(p, -32768, contcheck=False)
    assert int((a_o > 2 ** 24).sum()) > 12000 and a_o.max() > 2 ** 25
    for env in ({}, {"TDX_AD8_BIG_SCAN": "0"}, {"TDX_AD8_BIG_SCAN": "1"}, {"TDX_AD8_BIG_ONE_WAVE": "1"}, {"TDX_AD8_BIG_RING": "256"}, {"TDX_AD8_BIG_RING": "64", "TDX_AD8_BIG_ONE_WAVE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        a = ctx.aread8(p, -32768, contcheck=False)
        for k in env:
            monkeypatch.delenv(k)
        assert bits_equal(a, a_o), describe_diff(a, a_o, f"interleaved trunks {env}")
    a_c = oracle.aread8(p, -32768, contcheck=True)
    a = ctx.aread8(p, -32768, contcheck=True)
    assert bits_equal(a, a_c), describe_diff(a, a_c, "interleaved trunks with contamination")


@pytest.mark.slow
def test_aread8_counts_above_2_30(ctx):
    """33000 x 33000 comb: 1.09e9 cells on one GPU through the TILE-CONTRACTION path; the trunk's exact count passes 2^30 (the
    per-cell words of the local pass hold 30 bits: only in-tile counts live there; crossing counts are 32-bit).  Expected
    values are analytic (checked against the oracle at 4200^2 above): float32 k-ordered adds on the trunk."""
    amax = _comb_check(ctx, 33000)
    assert amax > 2 ** 30


def test_aread8_walk_beyond_count_limit(ctx, oracle, monkeypatch):
    """Rasters with >= 2^32 cells (all strips together) cannot use 32-bit exact counts: they take the pull walk.  The switch is
    exercised here through its test hook."""
    monkeypatch.setenv("TDX_AD8_COUNT_LIMIT", "1000")
    assert _comb_check(ctx, 4200, None) > 2 ** 24


def test_aread8_tiled_path_counts_the_participating_cells(ctx, oracle, monkeypatch):
    """A raster of more cells than the 32-bit exact counts hold (65536 x 65536 is exactly 2^32 cells) keeps the tile contraction as long as its PARTICIPATING
    cells stay below 2^32 (4 294 700 699 at BASELINE.json configs[3]): the limit hook is put between the two numbers of a small raster with a nodata hole."""
    dem = oracle.synth_dem((300, 400), 21)
    dem[40:220, 60:330] = -9999.0
    fel = oracle.pitremove(dem, -9999.0)
    p, _, _ = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    npart = int(((p >= 0) & (p <= 8)).sum())
    assert npart < p.size - 1000
    a_o = oracle.aread8(p, -32768, contcheck=False)
    monkeypatch.setenv("TDX_AD8_COUNT_LIMIT", str(npart + 10))          # cells > limit >= participating cells: tile contraction
    a, st = ctx.aread8(p, -32768, contcheck=False, stats=True)
    assert bits_equal(a, a_o), describe_diff(a, a_o, "ad8 (tile contraction below the participating-cell limit)")
    assert st["launches_accum"] <= 3, "the tile dependency sweep ran instead of the tile contraction"
    monkeypatch.setenv("TDX_AD8_COUNT_LIMIT", str(npart - 10))          # participating cells above the limit: the dependency sweep
    a, st = ctx.aread8(p, -32768, contcheck=False, stats=True)
    assert bits_equal(a, a_o), describe_diff(a, a_o, "ad8 (dependency sweep above the limit)")
    assert st["launches_accum"] > 3


def _canal(n):
    """one flat canal of n cells inside walls, draining at its west end: a flat whose deepest incfall level is n"""
    dem = np.full((5, n + 2), 100.0, np.float32)
    dem[2, 1:n + 1] = 10.0
    dem[2, 0] = 5.0
    return dem


@pytest.mark.parametrize("env", [{"TDX_LEVELS_INT32": "1"}, {"TDX_LEVELS_LIMIT": "40"}])
def test_int32_level_fields_give_the_reference_bits(g, ctx, monkeypatch, env):
    """The int32 level fields - the fallback for flats deeper than the reference's short counters hold - on inputs the reference can do: forced from the
    start (TDX_LEVELS_INT32) and entered through the fallback itself (TDX_LEVELS_LIMIT=40: the int16 pass gives up at level 40 and the call starts over),
    both must reproduce the REAL tools' p / sd8 / ang / slp."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p, sd8, st = ctx.d8flowdir(np.ascontiguousarray(g["fel"]), -3.0e38, g["dxc"], g["dyc"], stats=True)
    assert bits_equal(sd8, g["sd8"]), describe_diff(sd8, g["sd8"], "sd8")
    assert bits_equal(p, g["p"]), describe_diff(p, g["p"], "p")
    assert f"All slopes evaluated. {st['flats_initial']} flats to resolve." in str(g["d8_stderr"])
    ang, slp = ctx.dinfflowdir(np.ascontiguousarray(g["fel"]), -3.0e38, g["dxc"], g["dyc"])
    assert bits_equal(slp, g["slp"]), describe_diff(slp, g["slp"], "slp")
    assert bits_equal(ang, g["ang"]), describe_diff(ang, g["ang"], "ang")


@pytest.mark.slow
def test_flat_deeper_than_int16(ctx, oracle, monkeypatch):
    """A flat of 70 000 levels.  The reference's `short` elev2 / dn partitions wrap at 32 767 (src/d8.cpp:483-486: the real algorithm leaves 37 233 of
    the canal's cells without a direction - the restatement with 16-bit counters shows it), so there is nothing to be bit-equal WITH; the product
    starts the call over on int32 level fields and must give what the algorithm means - the restatement built with 32-bit counters
    (oracle/libtaudem_oracle_l32.so, ORC_LVL_T): every canal cell drains west.  A superset of the reference, not a parity claim.  One strip and two."""
    import torch

    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows

    n = 70000
    dem = _canal(n)
    fel = oracle.pitremove(dem, -9999.0)
    monkeypatch.setenv("ORC_FLATS", "bfs")
    p_o, sd8_o, st_o = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0, levels32=True)
    ang_o, slp_o, _ = oracle.dinfflowdir(fel, -3.0e38, 30.0, 30.0, levels32=True)
    monkeypatch.delenv("ORC_FLATS")
    assert int((p_o[2, 1:n + 1] == 5).sum()) == n
    p, sd8, st = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, stats=True)
    assert st["levels_fall_max"] >= n - 1 and st["flats_left"] == 0
    assert bits_equal(p, p_o), describe_diff(p, p_o, "p (70 000 levels)")
    assert bits_equal(sd8, sd8_o), describe_diff(sd8, sd8_o, "sd8")
    ang, slp = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    assert bits_equal(ang, ang_o), describe_diff(ang, ang_o, "ang (70 000 levels)")
    assert bits_equal(slp, slp_o), describe_diff(slp, slp_o, "slp")
    # two strips: the canal's row belongs to the second one, its walls to both - every rank must take the same turn to the int32 fields
    ny, nx = fel.shape
    parts = partition_rows(ny, 2)
    with StripGroup(2, nx, [0, 0]) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            pipe = StripPipeline(c, comm, nx, y1 - y0)
            f = pipe.empty(torch.float32)
            f[1:y1 - y0 + 1] = torch.from_numpy(fel[y0:y1]).cuda()
            pp, ss, _ = pipe.d8flowdir(f, -3.0e38, 30.0, 30.0)
            return pp[1:y1 - y0 + 1].cpu().numpy(), ss[1:y1 - y0 + 1].cpu().numpy()
        res = grp.run(rank_main)
    p2 = np.concatenate([r[0] for r in res], axis=0)
    s2 = np.concatenate([r[1] for r in res], axis=0)
    assert bits_equal(p2, p_o), describe_diff(p2, p_o, "p in two strips")
    assert bits_equal(s2, sd8_o), describe_diff(s2, sd8_o, "sd8 in two strips")


@pytest.mark.slow
def test_many_big_cells_in_one_three_and_eight_strips(ctx, oracle, monkeypatch):
    """The exact k-ordered re-evaluation of the cells above the float32-exact range (src/aread8.cpp:231-256) as the eight strips of BASELINE.json configs[3]
    see it - ~10^5 such cells per strip, main stems that cross the strip boundaries again and again, outer rounds that only re-visit what the
    neighbouring strips still block (ad8_big_compact_kernel) - at a size the restatement does in seconds: 2048^2 with the threshold lowered to 6
    (TDX_AD8_BIG_THRESHOLD; below 2^24 the float32 adds are exact whatever their order, so the restatement's raster is the answer).  >= 10^5 cells take
    the path; one strip, three and eight must give the same bits."""
    import os

    import torch

    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows

    n = 2048
    dem = oracle.synth_dem((n, n), 77)
    fel = oracle.pitremove(dem, -9999.0)
    monkeypatch.setenv("ORC_FLATS", "bfs")
    oracle.set_threads(os.cpu_count() or 1)
    try:
        p_o, _, _ = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    finally:
        oracle.set_threads(1)
        monkeypatch.delenv("ORC_FLATS")
    a_o = oracle.aread8(p_o, -32768, contcheck=False)
    monkeypatch.setenv("TDX_AD8_BIG_THRESHOLD", "6")
    assert int((a_o > 6).sum()) >= 100000
    a, st = ctx.aread8(p_o, -32768, contcheck=False, stats=True)
    assert st["cells_evaluated"] >= 100000, "the big-cell path was not taken"
    assert bits_equal(a, a_o), describe_diff(a, a_o, "ad8, one strip")
    for ring in ("64", "192", "1024"):   # the fold's LDS ring shortened: trees longer than the ring keep a copy in global memory and read far contributors back
        monkeypatch.setenv("TDX_AD8_BIG_RING", ring)
        a = ctx.aread8(p_o, -32768, contcheck=False)
        assert bits_equal(a, a_o), describe_diff(a, a_o, f"ad8, one strip, ring of {ring}")
    monkeypatch.delenv("TDX_AD8_BIG_RING")
    a_c = oracle.aread8(p_o, -32768, contcheck=True)
    a = ctx.aread8(p_o, -32768, contcheck=True)
    assert bits_equal(a, a_c), describe_diff(a, a_c, "ad8 with contamination, one strip")
    for world in (3, 8):
        if world == 3:
            monkeypatch.setenv("TDX_AD8_BIG_RING", "128")   # (three strips with a shortened ring, eight with the whole one)
        else:
            monkeypatch.delenv("TDX_AD8_BIG_RING", raising=False)
        parts = partition_rows(n, world)
        with StripGroup(world, n, [0] * world) as grp:
            def rank_main(r, c, comm):
                y0, y1 = parts[r]
                pipe = StripPipeline(c, comm, n, y1 - y0)
                pp = pipe.empty(torch.int16)
                pp[1:y1 - y0 + 1] = torch.from_numpy(p_o[y0:y1]).cuda()
                out = []
                for cc in (False, True):
                    aa, s = pipe.aread8(pp, -32768, contcheck=cc)
                    out.append(aa[1:y1 - y0 + 1].cpu().numpy())
                return out + [s["rounds"]]
            res = grp.run(rank_main)
        for i, ref in ((0, a_o), (1, a_c)):
            got = np.concatenate([r[i] for r in res], axis=0)
            assert bits_equal(got, ref), describe_diff(got, ref, f"ad8 in {world} strips (contcheck={bool(i)})")
        assert res[0][2] > 2, "no outer rounds: the strips never exchanged a big cell"


@pytest.mark.parametrize("after", [1, 3, 8])
def test_pitremove_coarse_correction_on_one_strip(ctx, oracle, monkeypatch, after):
    """PitRemove's coarse correction (restrict the fine surface after `after` rounds, relax the coarse level again, W <- min(W, Wc[block])) is only taken on
    one strip when the first coarse level needed >= 16 rounds - never at test sizes.  Forced here (TDX_PIT_VCYCLE_MIN=0) on rasters with lakes, nodata holes
    and a spiral channel: any upper bound converges to flood()'s surface, so the bits must not move."""
    import pathological as P

    monkeypatch.setenv("TDX_PIT_VCYCLE_MIN", "0")
    monkeypatch.setenv("TDX_PIT_VCYCLE_AFTER", str(after))
    dems = [oracle.synth_dem((700, 900), 31), oracle.synth_dem((1100, 640), 32), P.spiral(640, 8), P.checkerboard_pits(600, 700)]
    holes = oracle.synth_dem((800, 800), 33)
    holes[100:300, 200:650] = -9999.0
    holes[500:520, :] = -9999.0
    dems.append(holes)
    for i, dem in enumerate(dems):
        fel_o = oracle.pitremove(dem, -9999.0)
        fel, st = ctx.pitremove(dem, -9999.0, stats=True)
        assert bits_equal(fel, fel_o), describe_diff(fel, fel_o, f"fel (raster {i}, correction after {after} rounds)")
