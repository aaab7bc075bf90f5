"""Pins the CPU restatement (oracle/taudem_oracle.c) to rasters written by the REAL reference tools
(tests/golden/*.npz, produced by tests/golden/make_golden.py from oracle/_ref).  CPU only."""
import numpy as np
import pytest

from conftest import bits_equal, describe_diff, golden_cases, load_golden, load_golden_gridnet, outlets_to_indices

CASES = golden_cases()


@pytest.fixture(scope="module", params=CASES)
def g(request):
    return load_golden(request.param)


def test_have_cases():
    assert len(CASES) >= 5


def test_pitremove(g, oracle):
    mask = g["mask"] if "mask" in g else None
    fel = oracle.pitremove(g["dem"], float(g["nodata"]), mask=mask, fourway=bool(g["fourway"]))
    assert bits_equal(fel, g["fel"]), describe_diff(fel, g["fel"], "fel")


def test_d8flowdir(g, oracle):
    p, sd8, st = oracle.d8flowdir(g["fel"], -3.0e38, g["dxc"], g["dyc"])
    assert bits_equal(p, g["p"]), describe_diff(p, g["p"], "p")
    assert bits_equal(sd8, g["sd8"]), describe_diff(sd8, g["sd8"], "sd8")
    assert f"All slopes evaluated. {st['flats_initial']} flats to resolve." in str(g["d8_stderr"])


def test_flat_resolution_literal_loops(g, oracle, monkeypatch):
    """ORC_PLAIN=1: the reference's literal whole-queue sweeps of resolveflats (src/d8.cpp:523-558,606-638; src/dinf.cpp:650-787) instead of
    the active-list form the restatement uses by default (oracle/taudem_oracle.c: fast_incfall / fast_incrise) - both must give the
    reference's rasters, so the fast form that produced the 8192^2 / 16384^2 digests is pinned to the literal one."""
    monkeypatch.setenv("ORC_PLAIN", "1")
    p, sd8, _ = oracle.d8flowdir(g["fel"], -3.0e38, g["dxc"], g["dyc"])
    assert bits_equal(p, g["p"]), describe_diff(p, g["p"], "p (literal loops)")
    assert bits_equal(sd8, g["sd8"]), describe_diff(sd8, g["sd8"], "sd8 (literal loops)")
    ang, slp, _ = oracle.dinfflowdir(g["fel"], -3.0e38, g["dxc"], g["dyc"])
    assert bits_equal(ang, g["ang"]), describe_diff(ang, g["ang"], "ang (literal loops)")
    assert bits_equal(slp, g["slp"]), describe_diff(slp, g["slp"], "slp (literal loops)")


@pytest.mark.parametrize("threads", [1, 5])
def test_flat_resolution_breadth_first_form(g, oracle, monkeypatch, threads):
    """ORC_FLATS=bfs: the two flat loops as breadth-first searches (oracle/taudem_oracle.c: bfs_incfall / bfs_incrise - linear in the number of
    flat cells, what makes the restatement a checker at 32768^2) give the REAL reference's p / sd8 / ang / slp like the literal and the
    active-list forms, with one host thread and with several (the threaded loops write disjoint cells)."""
    monkeypatch.setenv("ORC_FLATS", "bfs")
    oracle.set_threads(threads)
    try:
        p, sd8, _ = oracle.d8flowdir(g["fel"], -3.0e38, g["dxc"], g["dyc"])
        ang, slp, _ = oracle.dinfflowdir(g["fel"], -3.0e38, g["dxc"], g["dyc"])
    finally:
        oracle.set_threads(1)
    assert bits_equal(p, g["p"]), describe_diff(p, g["p"], "p (bfs)")
    assert bits_equal(sd8, g["sd8"]), describe_diff(sd8, g["sd8"], "sd8 (bfs)")
    assert bits_equal(ang, g["ang"]), describe_diff(ang, g["ang"], "ang (bfs)")
    assert bits_equal(slp, g["slp"]), describe_diff(slp, g["slp"], "slp (bfs)")


@pytest.mark.parametrize("key,kw", [("ad8", {}), ("ad8_nc", {"contcheck": False}), ("ad8_w", {"w": True}), ("ad8_w_nc", {"w": True, "contcheck": False}),
                                    ("ad8_outlets", {"o": True}), ("ad8_outlets_nc", {"o": True, "contcheck": False})])
def test_aread8(g, oracle, key, kw):
    a = oracle.aread8(g["p"], -32768, weights=g["w"] if kw.get("w") else None, weights_nodata=-9999.0, contcheck=kw.get("contcheck", True),
                      outlets=outlets_to_indices(g) if kw.get("o") else None)
    assert bits_equal(a, g[key]), describe_diff(a, g[key], key)


def test_dinfflowdir(g, oracle):
    ang, slp, st = oracle.dinfflowdir(g["fel"], -3.0e38, g["dxc"], g["dyc"])
    assert bits_equal(ang, g["ang"]), describe_diff(ang, g["ang"], "ang")
    assert bits_equal(slp, g["slp"]), describe_diff(slp, g["slp"], "slp")


@pytest.mark.parametrize("key,kw", [("sca", {}), ("sca_nc", {"contcheck": False}), ("sca_w_nc", {"w": True, "contcheck": False}),
                                    ("sca_outlets_nc", {"o": True, "contcheck": False})])
def test_areadinf(g, oracle, key, kw):
    s = oracle.areadinf(g["ang"], -3.402823466e38, g["dxc"], g["dyc"], weights=g["w"] if kw.get("w") else None, contcheck=kw.get("contcheck", True),
                        outlets=outlets_to_indices(g) if kw.get("o") else None)
    assert bits_equal(s, g[key]), describe_diff(s, g[key], key)


@pytest.mark.parametrize("key,kw", [("dsca", {}), ("dsca_w_nc", {"w": True, "contcheck": False}), ("dsca_outlets_nc", {"o": True, "contcheck": False})])
def test_dinfdecayaccum(g, oracle, key, kw):
    s = oracle.dinfdecayaccum(g["ang"], g["dm"], -3.402823466e38, -9999.0, g["dxc"], g["dyc"], weights=g["w"] if kw.get("w") else None,
                              contcheck=kw.get("contcheck", True), outlets=outlets_to_indices(g) if kw.get("o") else None)
    assert bits_equal(s, g[key]), describe_diff(s, g[key], key)


@pytest.mark.parametrize("name", CASES)
def test_gridnet_and_threshold(name, oracle):
    """GridNet (plain and with -mask/-thresh) and Threshold (plain and with -mask) against the reference's rasters."""
    g, h = load_golden(name), load_golden_gridnet(name)
    plen, tlen, gord = oracle.gridnet(g["p"], -32768, g["dxc"], g["dyc"])
    assert bits_equal(plen, h["plen"]), describe_diff(plen, h["plen"], "plen")
    assert bits_equal(tlen, h["tlen"]), describe_diff(tlen, h["tlen"], "tlen")
    assert bits_equal(gord, h["gord"]), describe_diff(gord, h["gord"], "gord")
    plen, tlen, gord = oracle.gridnet(g["p"], -32768, g["dxc"], g["dyc"], mask=h["mask_i32"], thresh=int(h["gn_thresh"]))
    assert bits_equal(plen, h["plen_m"]), describe_diff(plen, h["plen_m"], "plen (mask)")
    assert bits_equal(tlen, h["tlen_m"]), describe_diff(tlen, h["tlen_m"], "tlen (mask)")
    assert bits_equal(gord, h["gord_m"]), describe_diff(gord, h["gord_m"], "gord (mask)")
    ox, oy = outlets_to_indices(g)
    plen, tlen, gord = oracle.gridnet(g["p"], -32768, g["dxc"], g["dyc"], outlets=(ox[:4], oy[:4]))   # -o: the four in-catchment points
    assert bits_equal(plen, h["plen_o"]), describe_diff(plen, h["plen_o"], "plen (outlets)")
    assert bits_equal(tlen, h["tlen_o"]), describe_diff(tlen, h["tlen_o"], "tlen (outlets)")
    assert bits_equal(gord, h["gord_o"]), describe_diff(gord, h["gord_o"], "gord (outlets)")
    src = oracle.threshold(g["ad8_nc"], float(h["ssa_thresh"]), -1.0)
    assert bits_equal(src, h["src"]), describe_diff(src, h["src"], "src")
    src = oracle.threshold(g["ad8_nc"], float(h["ssa_thresh"]), -1.0, mask=h["tmask"])
    assert bits_equal(src, h["src_m"]), describe_diff(src, h["src_m"], "src (mask)")


@pytest.mark.parametrize("name", CASES)
def test_d8flowpathextremeup(name, oracle):
    """Upstream maximum (with contamination check), minimum (-min -nc) and maximum restricted to outlets (-o -nc) of the D8 slope grid."""
    g, h = load_golden(name), load_golden_gridnet(name)
    a = oracle.d8flowpathextremeup(g["p"], g["sd8"], -32768, usemax=True, contcheck=True)
    assert bits_equal(a, h["xup_max"]), describe_diff(a, h["xup_max"], "xup_max")
    a = oracle.d8flowpathextremeup(g["p"], g["sd8"], -32768, usemax=False, contcheck=False)
    assert bits_equal(a, h["xup_min_nc"]), describe_diff(a, h["xup_min_nc"], "xup_min_nc")
    a = oracle.d8flowpathextremeup(g["p"], g["sd8"], -32768, usemax=True, contcheck=False, outlets=outlets_to_indices(g))
    assert bits_equal(a, h["xup_max_outlets_nc"]), describe_diff(a, h["xup_max_outlets_nc"], "xup_max_outlets_nc")


# ---- the linear-time DinfDecayAccum checker (oracle/taudem_oracle.c: orc_dinfdecayaccum_check) pinned to the reference's own rasters ----
@pytest.mark.parametrize("key,kw", [("dsca", {}), ("dsca_w_nc", {"w": True, "contcheck": False}), ("dsca_outlets_nc", {"o": True, "contcheck": False})])
def test_decay_checker_accepts_the_reference_rasters_and_nothing_else(g, oracle, key, kw):
    """dmarea()'s loop body applied to every cell of the raster the REAL tool wrote reproduces that raster (0 mismatches), the set of queued cells
    is the set of cells that got a value in -nc mode, and one flipped bit, one value outside the outlets' catchments or one missing value is noticed."""
    w = g["w"] if kw.get("w") else None
    outl = outlets_to_indices(g) if kw.get("o") else None
    cc = kw.get("contcheck", True)
    args = dict(nodata=-3.402823466e38, dm_nodata=-9999.0, dx=g["dxc"], dy=g["dyc"], weights=w, contcheck=cc, outlets=outl)
    ref = np.array(g[key], copy=True)
    bad, first, queued = oracle.dinfdecayaccum_check(g["ang"], g["dm"], ref, **args)
    assert (bad, first) == (0, -1), f"{key}: {bad} cells of the reference's raster fail the check, first at {first}"
    have = ref != np.float32(-3.402823466e38)
    if not cc:
        assert queued == int(have.sum())     # without contamination every queued cell ends with a value
    for threads in (1, 3):                   # the closure search and the row bands give the same answer for any thread count
        assert oracle.dinfdecayaccum_check(g["ang"], g["dm"], ref, threads=threads, **args) == (0, -1, queued)
    ys, xs = np.nonzero(have)
    y, x = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
    t = ref.copy()
    t[y, x] = np.nextafter(t[y, x], np.float32(1e30))
    assert oracle.dinfdecayaccum_check(g["ang"], g["dm"], t, **args)[0] >= 1
    t = ref.copy()
    t[y, x] = np.float32(-3.402823466e38)    # a cell that was never evaluated although its contributors were
    assert oracle.dinfdecayaccum_check(g["ang"], g["dm"], t, **args)[0] >= 1
    ys, xs = np.nonzero(~have & (g["ang"] != np.float32(-3.402823466e38)))
    if outl is not None and len(ys):
        t = ref.copy()
        t[ys[0], xs[0]] = np.float32(1.0)    # a value outside the outlets' upstream closure
        assert oracle.dinfdecayaccum_check(g["ang"], g["dm"], t, **args)[0] >= 1


def test_decay_checker_vs_restatement_with_weights_outlets_and_contamination(oracle):
    """The combination BASELINE.json configs[4] runs (-wg, -o, contamination check on, decay multipliers with nodata holes) at a size the restatement's
    queue loop finishes: checker and restatement agree, and the closure is exactly the set the restatement evaluates in -nc mode."""
    rng = np.random.default_rng(5)
    n = 500
    dem = oracle.synth_dem(n, 9)
    fel = oracle.pitremove(dem, -9999.0)
    ang, _, _ = oracle.dinfflowdir(fel, -3.0e38, 30.0, 25.0)
    w = rng.random((n, n), dtype=np.float32)
    dm = (0.9 + 0.1 * rng.random((n, n), dtype=np.float32)).astype(np.float32)
    dm[100:110, 200:230] = -9999.0
    sca = oracle.areadinf(ang, dx=30.0, dy=25.0, contcheck=False)
    order = np.argsort(sca, axis=None)[::-1]
    outl = (np.array([order[3] % n, order[400] % n, 250, 0], dtype=np.int32), np.array([order[3] // n, order[400] // n, 250, 17], dtype=np.int32))
    for cc in (True, False):
        d = oracle.dinfdecayaccum(ang, dm, dx=30.0, dy=25.0, weights=w, contcheck=cc, outlets=outl)
        bad, first, queued = oracle.dinfdecayaccum_check(ang, dm, d, dx=30.0, dy=25.0, weights=w, contcheck=cc, outlets=outl)
        assert (bad, first) == (0, -1)
        if not cc:
            mark = oracle.dinf_outlet_closure(ang, outl, dx=30.0, dy=25.0)
            assert np.array_equal(mark != 0, d != np.float32(-3.402823466e38)) and queued == int(mark.sum())
    d = oracle.dinfdecayaccum(ang, dm, dx=30.0, dy=25.0, contcheck=True)      # no outlets, no weights
    assert oracle.dinfdecayaccum_check(ang, dm, d, dx=30.0, dy=25.0, contcheck=True)[:2] == (0, -1)


@pytest.mark.parametrize("key,kw", [("ad8", {}), ("ad8_nc", {"contcheck": False}), ("ad8_w", {"w": True}), ("ad8_w_nc", {"w": True, "contcheck": False})])
def test_aread8_checker_accepts_the_reference_rasters_and_nothing_else(g, oracle, key, kw):
    """The linear-time AreaD8 checker (aread8()'s loop body applied to every cell of a given raster: oracle/taudem_oracle.c: orc_aread8_check) passes the rasters
    the REAL tool wrote and notices a wrong count, a missing value and a value on a cell without direction."""
    w = g["w"] if kw.get("w") else None
    cc = kw.get("contcheck", True)
    ref = np.array(g[key], copy=True)
    bad, first, queued = oracle.aread8_check(g["p"], ref, -32768, weights=w, contcheck=cc)
    assert (bad, first) == (0, -1), f"{key}: {bad} cells of the reference's raster fail the check, first at {first}"
    assert queued == int(((g["p"] >= 0) & (g["p"] <= 8)).sum())
    assert oracle.aread8_check(g["p"], ref, -32768, weights=w, contcheck=cc, threads=3)[:2] == (0, -1)
    ys, xs = np.nonzero(ref > 0)
    y, x = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
    t = ref.copy(); t[y, x] += np.float32(1.0) if ref[y, x] < 2 ** 23 else np.float32(8.0)
    assert oracle.aread8_check(g["p"], t, -32768, weights=w, contcheck=cc)[0] >= 1
    t = ref.copy(); t[y, x] = np.float32(-1.0)
    assert oracle.aread8_check(g["p"], t, -32768, weights=w, contcheck=cc)[0] >= 1
    ys, xs = np.nonzero(g["p"] == -32768)
    if len(ys):
        t = ref.copy(); t[ys[0], xs[0]] = np.float32(1.0)
        assert oracle.aread8_check(g["p"], t, -32768, weights=w, contcheck=cc)[0] >= 1


def _fill_level(fel, dem, nodata):
    """cells the reference raised (fel > dem, both data)"""
    return (fel > dem) & (dem != np.float32(nodata))


def test_pitremove_checker_accepts_the_reference_rasters_and_nothing_else(g, oracle):
    """The linear-time PitRemove certificate (oracle/taudem_oracle.c: orc_pitremove_check - every cell against the fixed-point equation of flood()'s
    relaxation, src/flood.cpp:243-271,292-331, plus a flood from the seed cells that must reach every data cell) passes the `fel` the REAL tool wrote
    (all five cases: holes, -4way + depression mask among them), with one thread and with several, and notices (a) one raised cell, (b) one lowered
    filled cell, (c) a value on a nodata cell and (d) a closed basin that was left UN-FILLED - the defect idempotence under the product's own
    operator cannot see, and the reason for the flood: every cell of such a basin satisfies its local equation."""
    dem, ref, nodata = g["dem"], np.array(g["fel"], copy=True), float(g["nodata"])
    mask = g["mask"] if "mask" in g else None
    fw = bool(g["fourway"])
    ndata = int((np.abs(dem - np.float32(nodata)) >= 1e-5).sum())
    for th in (1, 4):
        bad, first, reached = oracle.pitremove_check(dem, ref, nodata, mask=mask, fourway=fw, threads=th)
        assert (bad, first) == (0, -1), f"{bad} cells of the reference's fel fail the certificate, first at {first}"
        assert reached == ndata
    raised = _fill_level(ref, dem, nodata)
    ys, xs = np.nonzero(raised)
    assert len(ys) > 0, "the golden case has no filled cell"
    y, x = int(ys[len(ys) // 2]), int(xs[len(xs) // 2])
    t = ref.copy(); t[y, x] = np.nextafter(t[y, x], np.float32(np.inf))                       # (a) one ulp too high
    assert oracle.pitremove_check(dem, t, nodata, mask=mask, fourway=fw)[0] >= 1
    t = ref.copy(); t[y, x] = np.nextafter(t[y, x], np.float32(-np.inf))                      # (b) one ulp too low
    assert oracle.pitremove_check(dem, t, nodata, mask=mask, fourway=fw)[0] >= 1
    ys0, xs0 = np.nonzero(np.abs(dem - np.float32(nodata)) < 1e-5)
    if len(ys0):
        t = ref.copy(); t[ys0[0], xs0[0]] = np.float32(1.0)                                     # (c)
        assert oracle.pitremove_check(dem, t, nodata, mask=mask, fourway=fw)[0] >= 1
    # (d) the largest filled basin (cells of one fill level, 4-connected through the raised set) put back to an under-filled level: its cells to
    # max(dem, level - d) with level - d above the basin's floor - water cells hold their neighbours' minimum, dry cells their elevation: every
    # equation of (i) holds inside; only the rim (which now has a lower neighbour but keeps its level) and the flood can tell
    from scipy import ndimage
    lab, n = ndimage.label(raised)
    sizes = ndimage.sum(raised, lab, index=np.arange(1, n + 1))
    big = int(np.argmax(sizes)) + 1
    basin = lab == big
    level = np.float32(ref[basin].max())
    floor = np.float32(dem[basin].min())
    if level > floor:
        low = np.float32(floor + (level - floor) * np.float32(0.5))
        t = ref.copy()
        t[basin] = np.maximum(dem[basin], low)
        bad, first, reached = oracle.pitremove_check(dem, t, nodata, mask=mask, fourway=fw)
        assert bad >= 1 and reached < ndata, "an under-filled closed basin passed the certificate"


def test_pitremove_checker_on_a_closed_two_cell_basin(oracle):
    """The smallest closed basin: two cells at 1 inside a rim at 5 on a plane at 2 that drains to the edge.  flood() fills them to 5.  Left at 1 - or at
    3, both cells equal, each holding its neighbours' minimum - every cell satisfies its own equation except that nothing drains: the flood
    from the seed cells never enters (it would have to step down from the rim)."""
    dem = np.full((9, 10), 2.0, np.float32)
    dem[3:6, 3:7] = 5.0
    dem[4, 4:6] = 1.0
    fel = oracle.pitremove(dem, -9999.0)
    assert fel[4, 4] == 5.0 and fel[4, 5] == 5.0
    assert oracle.pitremove_check(dem, fel, -9999.0)[:2] == (0, -1)
    for lvl in (1.0, 3.0):
        t = fel.copy(); t[4, 4:6] = np.float32(lvl)
        bad, first, reached = oracle.pitremove_check(dem, t, -9999.0)
        assert bad >= 1 and reached < dem.size, f"basin left at {lvl}"
    t = fel.copy(); t[4, 4:6] = np.float32(6.0)     # over-filled: above the rim
    assert oracle.pitremove_check(dem, t, -9999.0)[0] >= 1
