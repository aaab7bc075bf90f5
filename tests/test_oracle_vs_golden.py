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
