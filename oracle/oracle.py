"""TEST INFRASTRUCTURE - ctypes wrapper of the CPU restatement (oracle/taudem_oracle.c) and a runner
for the real reference tools built into oracle/_ref/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under taudem_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtaudem_oracle.so")
LIB_L32 = os.path.join(HERE, "libtaudem_oracle_l32.so")   # the same source with 32-bit level counters in resolveflats() (ORC_LVL_T)
REF_DIR = os.path.join(HERE, "_ref")
MPIEXEC = "/opt/conda/bin/mpiexec"

_lib = None


def build(force=False):
    """Compile the C restatement (and, when /root/reference is present, the reference tools)."""
    src_t = os.path.getmtime(os.path.join(HERE, "taudem_oracle.c"))
    if force or any(not os.path.exists(f) or os.path.getmtime(f) < src_t for f in (LIB, LIB_L32)):
        subprocess.run(["make", "-C", HERE, "restatement"], check=True, capture_output=True)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", HERE, "ref", "-j8"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.orc_prop.restype = C.c_double
        _lib.orc_prop.argtypes = [C.c_float, C.c_int, C.c_double, C.c_double]
    return _lib


_lib32 = None


def lib32():
    """The restatement with 32-bit level counters (flats deeper than the reference's `short` partitions hold): a superset of the reference, not a parity claim."""
    global _lib32
    if _lib32 is None:
        build()
        _lib32 = C.CDLL(LIB_L32)
    return _lib32


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def set_threads(n):
    """Host threads for the restatement's loops with independent iterations (default 1; results do not depend on n)."""
    lib().orc_set_threads(C.c_int(int(n)))


def _f64(a, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (n,)))


def _outl(outlets):
    if outlets is None:
        return None, None, 0, 0, ()
    ox = np.ascontiguousarray(np.asarray(outlets[0], dtype=np.int32))
    oy = np.ascontiguousarray(np.asarray(outlets[1], dtype=np.int32))
    return _p(ox), _p(oy), int(ox.size), 1, (ox, oy)


def synth_dem(n_or_shape, seed=1234, x0=0, y0=0, base_wavelength=None):
    if isinstance(n_or_shape, int):
        ny = nx = n_or_shape
    else:
        ny, nx = n_or_shape
    if base_wavelength is None:
        base_wavelength = 2
        while base_wavelength * 2 < max(nx, ny):
            base_wavelength *= 2
    out = np.empty((ny, nx), dtype=np.float32)
    lib().orc_synth_dem(C.c_uint64(seed), C.c_long(nx), C.c_long(ny), C.c_long(x0), C.c_long(y0), C.c_long(base_wavelength), _p(out))
    return out


def pitremove(dem, nodata=-9999.0, mask=None, fourway=False):
    dem = np.ascontiguousarray(dem, dtype=np.float32)
    ny, nx = dem.shape
    fel = np.empty_like(dem)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.int16)
    lib().orc_pitremove(_p(dem), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(mask), C.c_int(int(fourway)), _p(fel))
    return fel


def d8flowdir(fel, nodata=-3.0e38, dx=1.0, dy=1.0, levels32=False):
    fel = np.ascontiguousarray(fel, dtype=np.float32)
    ny, nx = fel.shape
    p = np.empty((ny, nx), dtype=np.int16)
    sd8 = np.empty((ny, nx), dtype=np.float32)
    st = (C.c_long * 8)()
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    (lib32() if levels32 else lib()).orc_d8flowdir(_p(fel), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(p), _p(sd8), st)
    stats = {"flats_initial": st[0], "flat_iterations": st[1], "flats_left": st[2], "sweeps_fall": st[3], "sweeps_rise": st[4]}
    return p, sd8, stats


def aread8(p, nodata=-32768, weights=None, weights_nodata=-9999.0, contcheck=True, outlets=None):
    p = np.ascontiguousarray(p, dtype=np.int16)
    ny, nx = p.shape
    ad8 = np.empty((ny, nx), dtype=np.float32)
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float32)
    ox, oy, no, use, keep = _outl(outlets)
    lib().orc_aread8(_p(p), C.c_long(nx), C.c_long(ny), C.c_int16(nodata), _p(weights), C.c_float(weights_nodata), C.c_int(int(contcheck)),
                     ox, oy, C.c_int(no), C.c_int(use), _p(ad8))
    return ad8


def pitremove_check(dem, fel, nodata=-9999.0, mask=None, fourway=False, threads=None):
    """Linear-time certificate of a PitRemove result (flood(), src/flood.cpp:243-479): every cell against the fixed-point equation of the relaxation
    (seed rule; W = Z if Z >= min of the neighbours' W, else that minimum) AND a flood from the seed cells through neighbours that are not lower,
    which must reach every data cell (an under-filled closed basin satisfies every equation and is only found by the flood).
    Returns (offending cells, index of the first one or -1, cells the flood reached)."""
    dem = np.ascontiguousarray(dem, dtype=np.float32)
    fel = np.ascontiguousarray(fel, dtype=np.float32)
    ny, nx = dem.shape
    assert fel.shape == dem.shape
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.int16)
    first, reached = C.c_long(-1), C.c_long(0)
    f = lib().orc_pitremove_check
    f.restype = C.c_long
    bad = f(_p(dem), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(mask), C.c_int(int(fourway)), _p(fel), C.c_int(int(threads or os.cpu_count() or 1)),
            C.byref(first), C.byref(reached))
    return int(bad), int(first.value), int(reached.value)


def aread8_check(p, ad8, nodata=-32768, weights=None, weights_nodata=-9999.0, contcheck=True, threads=None):
    """Linear-time pin of aread8()'s loop body (src/aread8.cpp:231-256) to a given result (no outlets): (cells of `ad8` that are not what their
    contributors' values in `ad8` give, index of the first one or -1, cells with a direction code)."""
    p = np.ascontiguousarray(p, dtype=np.int16)
    ad8 = np.ascontiguousarray(ad8, dtype=np.float32)
    ny, nx = p.shape
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float32)
    first, queued = C.c_long(-1), C.c_long(0)
    f = lib().orc_aread8_check
    f.restype = C.c_long
    bad = f(_p(p), C.c_long(nx), C.c_long(ny), C.c_int16(nodata), _p(weights), C.c_float(weights_nodata), C.c_int(int(contcheck)), _p(ad8),
            C.c_int(int(threads or os.cpu_count() or 1)), C.byref(first), C.byref(queued))
    return int(bad), int(first.value), int(queued.value)


def d8flowpathextremeup(p, sa, nodata=-32768, usemax=True, contcheck=True, outlets=None):
    """ssa of src/D8flowpathextremeup.cpp: max / min of `sa` over everything upstream of a cell (nodata -FLT_MAX)."""
    p = np.ascontiguousarray(p, dtype=np.int16)
    sa = np.ascontiguousarray(sa, dtype=np.float32)
    ny, nx = p.shape
    ssa = np.empty((ny, nx), dtype=np.float32)
    ox, oy, no, use, keep = _outl(outlets)
    lib().orc_d8flowpathextremeup(_p(p), C.c_long(nx), C.c_long(ny), C.c_int16(nodata), _p(sa), C.c_int(int(usemax)), C.c_int(int(contcheck)),
                                  ox, oy, C.c_int(no), C.c_int(use), _p(ssa))
    return ssa


def gridnet(p, nodata=-32768, dx=1.0, dy=1.0, mask=None, thresh=0, outlets=None):
    """(plen, tlen, gord) of src/gridnet.cpp; mask: int32 raster (cells with mask >= thresh are evaluated); outlets: (columns, rows)."""
    p = np.ascontiguousarray(p, dtype=np.int16)
    ny, nx = p.shape
    plen = np.empty((ny, nx), dtype=np.float32)
    tlen = np.empty((ny, nx), dtype=np.float32)
    gord = np.empty((ny, nx), dtype=np.int16)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.int32)
    ox, oy, no, use, keep = _outl(outlets)
    lib().orc_gridnet(_p(p), C.c_long(nx), C.c_long(ny), C.c_int16(nodata), _p(dxc), _p(dyc), _p(mask), C.c_int(int(thresh)), ox, oy, C.c_int(no),
                      C.c_int(use), _p(plen), _p(tlen), _p(gord))
    return plen, tlen, gord


def threshold(ssa, thresh, nodata=-1.0, mask=None):
    """src of src/Threshold.cpp: 1 where ssa >= thresh (and mask >= 0), 0 elsewhere, -32768 where ssa is nodata."""
    ssa = np.ascontiguousarray(ssa, dtype=np.float32)
    ny, nx = ssa.shape
    src = np.empty((ny, nx), dtype=np.int16)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.float32)
    lib().orc_threshold(_p(ssa), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(mask), C.c_float(thresh), _p(src))
    return src


def dinfflowdir(fel, nodata=-3.0e38, dx=1.0, dy=1.0, levels32=False):
    fel = np.ascontiguousarray(fel, dtype=np.float32)
    ny, nx = fel.shape
    ang = np.empty((ny, nx), dtype=np.float32)
    slp = np.empty((ny, nx), dtype=np.float32)
    st = (C.c_long * 8)()
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    (lib32() if levels32 else lib()).orc_dinfflowdir(_p(fel), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(ang), _p(slp), st)
    return ang, slp, {"flats_initial": st[0], "flat_iterations": st[1], "flats_left": st[2]}


def areadinf(ang, nodata=-3.402823466e38, dx=1.0, dy=1.0, weights=None, contcheck=True, outlets=None):
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    ny, nx = ang.shape
    sca = np.empty((ny, nx), dtype=np.float32)
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float32)
    ox, oy, no, use, keep = _outl(outlets)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    lib().orc_areadinf(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(weights),
                       C.c_int(int(contcheck)), ox, oy, C.c_int(no), C.c_int(use), _p(sca))
    return sca


def areadinf_check(ang, sca, nodata=-3.402823466e38, dx=1.0, dy=1.0, weights=None, contcheck=True, threads=None):
    """Linear-time pin of area()'s per-cell expression (src/areadinf.cpp:187-217) to a given result: returns (number of cells of `sca` that are
    not what their contributors' values in `sca` give, index of the first one or -1).  0 means `sca` is the raster areadinf(ang) produces."""
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    sca = np.ascontiguousarray(sca, dtype=np.float32)
    ny, nx = ang.shape
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float32)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    first = C.c_long(-1)
    f = lib().orc_areadinf_check
    f.restype = C.c_long
    bad = f(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(weights), C.c_int(int(contcheck)), _p(sca),
            C.c_int(int(threads or os.cpu_count() or 1)), C.byref(first))
    return int(bad), int(first.value)


def dinf_first_pass_check(fel, ang, slp, nodata=-3.0e38, dx=1.0, dy=1.0, threads=None):
    """Linear-time pin of setdir()'s first pass (src/dinf.cpp:549-593, SET2 :317-373): returns (mismatching cells, first index or -1, cells
    the pass leaves flat).  On non-flat cells ang / slp must be bit-exact; on flat cells only the slope and the range of the angle are checked."""
    fel = np.ascontiguousarray(fel, dtype=np.float32)
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    slp = np.ascontiguousarray(slp, dtype=np.float32)
    ny, nx = fel.shape
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    first, flats = C.c_long(-1), C.c_long(0)
    f = lib().orc_dinf_first_pass_check
    f.restype = C.c_long
    bad = f(_p(fel), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(ang), _p(slp), C.c_int(int(threads or os.cpu_count() or 1)),
            C.byref(first), C.byref(flats))
    return int(bad), int(first.value), int(flats.value)


def dinfdecayaccum(ang, dm, nodata=-3.402823466e38, dm_nodata=-9999.0, dx=1.0, dy=1.0, weights=None, contcheck=True, outlets=None):
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    dm = np.ascontiguousarray(dm, dtype=np.float32)
    ny, nx = ang.shape
    out = np.empty((ny, nx), dtype=np.float32)
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float32)
    ox, oy, no, use, keep = _outl(outlets)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    lib().orc_dinfdecayaccum(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(dm),
                             C.c_float(dm_nodata), _p(weights), C.c_int(int(contcheck)), ox, oy, C.c_int(no), C.c_int(use), _p(out))
    return out


def dinf_outlet_closure(ang, outlets, nodata=-3.402823466e38, dx=1.0, dy=1.0, threads=None):
    """uint8 raster: 1 on the upstream closure of the outlet cells over prop() > 0 edges (the cells initNeighborDinfup puts to work with
    useOutlets, src/commonLib.cpp:165-233), found by a parallel search."""
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    ny, nx = ang.shape
    mark = np.empty((ny, nx), dtype=np.uint8)
    ox, oy, no, use, keep = _outl(outlets)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    f = lib().orc_dinf_outlet_closure
    f.restype = C.c_long
    f(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), ox, oy, C.c_int(no), C.c_int(int(threads or os.cpu_count() or 1)), _p(mark))
    return mark


def dinfdecayaccum_check(ang, dm, dsca, nodata=-3.402823466e38, dm_nodata=-9999.0, dx=1.0, dy=1.0, weights=None, contcheck=True, outlets=None, threads=None):
    """Linear-time pin of dmarea() (src/dinfdecayaccum.cpp:204-291) to a given result: returns (cells of `dsca` that are not what the loop body gives
    for their contributors' values in `dsca` - or that hold a value although dmarea() never queues them -, index of the first one or -1, cells
    dmarea() queues: every cell with an angle, or the outlets' upstream closure).  0 mismatches: `dsca` is the raster dinfdecayaccum() produces."""
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    dm = np.ascontiguousarray(dm, dtype=np.float32)
    dsca = np.ascontiguousarray(dsca, dtype=np.float32)
    ny, nx = ang.shape
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float32)
    ox, oy, no, use, keep = _outl(outlets)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    first, queued = C.c_long(-1), C.c_long(0)
    f = lib().orc_dinfdecayaccum_check
    f.restype = C.c_long
    bad = f(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(dm), C.c_float(dm_nodata), _p(weights), C.c_int(int(contcheck)),
            ox, oy, C.c_int(no), C.c_int(use), _p(dsca), C.c_int(int(threads or os.cpu_count() or 1)), C.byref(first), C.byref(queued))
    return int(bad), int(first.value), int(queued.value)


def dinfupdependence(ang, dg, nodata=-3.402823466e38, dx=1.0, dy=1.0):
    """dep of src/DinfUpDependence.cpp (nodata -1)."""
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    dg = np.ascontiguousarray(dg, dtype=np.int32)
    ny, nx = ang.shape
    dep = np.empty((ny, nx), dtype=np.float32)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    lib().orc_dinfupdependence(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(dg), _p(dep))
    return dep


def dinfrevaccum(ang, w, nodata=-3.402823466e38, w_nodata=-9999.0, dx=1.0, dy=1.0):
    """(racc, dmax) of src/DinfRevAccum.cpp (nodata -FLT_MAX)."""
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    ny, nx = ang.shape
    racc = np.empty((ny, nx), dtype=np.float32)
    dmax = np.empty((ny, nx), dtype=np.float32)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    lib().orc_dinfrevaccum(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(w), C.c_float(w_nodata), _p(racc), _p(dmax))
    return racc, dmax


def dinfconclimaccum(ang, dm, dg, q, csol=1.0, nodata=-3.402823466e38, dm_nodata=-9999.0, q_nodata=-9999.0, dx=1.0, dy=1.0, contcheck=True, outlets=None):
    """ctpt of src/DinfConcLimAccum.cpp (nodata -FLT_MAX); dg is the int16 indicator grid."""
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    dm = np.ascontiguousarray(dm, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    dg = np.ascontiguousarray(dg, dtype=np.int16)
    ny, nx = ang.shape
    out = np.empty((ny, nx), dtype=np.float32)
    ox, oy, no, use, keep = _outl(outlets)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    lib().orc_dinfconclimaccum(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(dm), C.c_float(dm_nodata), _p(dg), _p(q),
                               C.c_float(q_nodata), C.c_float(csol), C.c_int(int(contcheck)), ox, oy, C.c_int(no), C.c_int(use), _p(out))
    return out


def dinftranslimaccum(ang, tsup, tc, cs=None, nodata=-3.402823466e38, tsup_nodata=-9999.0, tc_nodata=-9999.0, cs_nodata=-9999.0, dx=1.0, dy=1.0,
                      contcheck=True, outlets=None):
    """(tla, tdep, ctpt or None) of src/DinfTransLimAccum.cpp (nodata -FLT_MAX)."""
    ang = np.ascontiguousarray(ang, dtype=np.float32)
    tsup = np.ascontiguousarray(tsup, dtype=np.float32)
    tc = np.ascontiguousarray(tc, dtype=np.float32)
    ny, nx = ang.shape
    tla = np.empty((ny, nx), dtype=np.float32)
    dep = np.empty((ny, nx), dtype=np.float32)
    usec = cs is not None
    csa = np.ascontiguousarray(cs, dtype=np.float32) if usec else None
    cso = np.empty((ny, nx), dtype=np.float32) if usec else None
    ox, oy, no, use, keep = _outl(outlets)
    dxc, dyc = _f64(dx, ny), _f64(dy, ny)
    lib().orc_dinftranslimaccum(_p(ang), C.c_long(nx), C.c_long(ny), C.c_float(nodata), _p(dxc), _p(dyc), _p(tsup), C.c_float(tsup_nodata), _p(tc),
                                C.c_float(tc_nodata), _p(csa), C.c_float(cs_nodata), C.c_int(int(usec)), C.c_int(int(contcheck)), ox, oy, C.c_int(no),
                                C.c_int(use), _p(tla), _p(dep), _p(cso))
    return tla, dep, cso


def prop(a, k, dx, dy):
    return lib().orc_prop(C.c_float(a), C.c_int(k), C.c_double(dx), C.c_double(dy))


# ---------------------------------------------------------------------------------------------
# the real reference tools (oracle/_ref, built from /root/reference by oracle/Makefile)
# ---------------------------------------------------------------------------------------------
def ref_available(tool="pitremove"):
    return os.path.exists(os.path.join(REF_DIR, tool))


def run_ref(tool, args, ranks=1, timeout=3600, cwd=None):
    """Runs oracle/_ref/<tool> (under mpiexec when ranks > 1).  Returns (stdout, stderr, seconds dict)."""
    exe = os.path.join(REF_DIR, tool)
    cmd = [exe] + [str(a) for a in args]
    if ranks > 1:
        cmd = [MPIEXEC, "-n", str(ranks)] + cmd
    env = dict(os.environ)
    env["PATH"] = "/opt/conda/bin:" + env.get("PATH", "")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=cwd)
    if r.returncode != 0:
        raise RuntimeError(f"{tool} failed ({r.returncode}): {r.stdout[-2000:]} {r.stderr[-2000:]}")
    times = {}
    for key in ("Compute time", "Compute Slope time", "Resolve Flat time", "Total time", "Data read time", "Read time", "Write time"):
        m = re.search(re.escape(key) + r":\s*([0-9.eE+-]+)", r.stdout)
        if m:
            times[key] = float(m.group(1))
    return r.stdout, r.stderr, times
