"""In-memory interface to the HIP hot path (one :class:`Context` per GPU / process rank).

Every method accepts either numpy arrays (host buffers: the library stages them over PCIe) or torch
CUDA tensors on the context's device (HBM-resident: no copies, the benchmark path).  Array layout,
dtypes and nodata conventions are the reference's (include/taudem_amd.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TdxStats, check

FEL_NODATA = np.float32(-3.0e38)            # src/flood.cpp:136
P_NODATA = np.int16(-32768)                 # src/d8.cpp:231
SLOPE_NODATA = np.float32(-1.0)             # src/d8.cpp:278
AREA_NODATA = np.float32(-1.0)              # src/aread8.cpp:193
ANG_NODATA = np.float32(-3.402823466e38)    # MISSINGFLOAT, src/commonLib.h:80


def _is_torch(x):
    return x is not None and type(x).__module__.startswith("torch")


def _f64(a, n):
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (n,)))
    return a


class Context:
    """Owns a HIP stream, a scratch arena and timing events on one device."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.tdx_context_create(int(device), C.byref(h)))
        self._h = h
        self._owned = True
        self.device = int(device)

    @classmethod
    def borrow(cls, handle, device: int):
        """A Context over a tdx_context* that somebody else owns (a rank of a tdx_group: taudem_amd.distributed.StripGroup)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = handle if isinstance(handle, C.c_void_p) else C.c_void_p(handle)
        self._owned = False
        self.device = int(device)
        return self

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._lib.tdx_context_destroy(self._h)
            self._h = None

    def release_scratch(self):
        """Frees the scratch arena (it grows again on demand)."""
        check(self._lib.tdx_context_release_scratch(self._h), self._h)

    def set_option(self, name: str, value: int):
        """Context options of include/taudem_amd.h (e.g. "kernel_timing")."""
        check(self._lib.tdx_context_set_option(self._h, name.encode(), int(value)), self._h)

    def segments(self):
        """The segment trace since the last call (option "segment_trace"): [(stage, phase, kind, device_ms, wall_ms)], kind 0 = ended by a halo
        exchange, 1 = by an all-reduce, 2 = by the end of the call.  Clears the log."""
        n = int(self._lib.tdx_context_segments(self._h, None, 0))
        if n == 0:
            return []
        buf = (_lib.TdxSegment * n)()
        self._lib.tdx_context_segments(self._h, buf, n)
        return [(b.stage.decode(), b.phase.decode(), int(b.kind), float(b.device_ms), float(b.wall_ms)) for b in buf]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- helpers --------------------------------------------------------------------------
    def _ptr(self, a, dtype, shape=None, name="array"):
        """(pointer, is_device) for a numpy array or torch cuda tensor; validates dtype/contiguity."""
        if a is None:
            return None, None
        if _is_torch(a):
            import torch

            want = {np.float32: torch.float32, np.int16: torch.int16, np.int32: torch.int32}[dtype]
            if not a.is_cuda or a.device.index != self.device:
                raise ValueError(f"{name}: tensor must live on cuda:{self.device}")
            if a.dtype != want or not a.is_contiguous():
                raise ValueError(f"{name}: need contiguous {want}")
            if shape is not None and tuple(a.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(a.shape)} != {tuple(shape)}")
            return C.c_void_p(a.data_ptr()), True
        if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags.c_contiguous:
            raise ValueError(f"{name}: need C-contiguous numpy {np.dtype(dtype)}")
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(a.shape)} != {tuple(shape)}")
        return C.c_void_p(a.ctypes.data), False

    def _out(self, like, dtype, shape):
        if _is_torch(like):
            import torch

            td = {np.float32: torch.float32, np.int16: torch.int16, np.int32: torch.int32}[dtype]
            return torch.empty(shape, dtype=td, device=like.device)
        return np.empty(shape, dtype=dtype)

    def _sync_torch(self, *arrs):
        if any(_is_torch(a) for a in arrs):
            import torch

            torch.cuda.synchronize(self.device)

    @staticmethod
    def _outlets(outlets):
        if outlets is None:
            return None, None, -1, ()
        ox = np.ascontiguousarray(np.asarray(outlets[0], dtype=np.int32))
        oy = np.ascontiguousarray(np.asarray(outlets[1], dtype=np.int32))
        if ox.shape != oy.shape or ox.ndim != 1:
            raise ValueError("outlets: need two equal-length 1-D index arrays (columns, rows)")
        return C.c_void_p(ox.ctypes.data), C.c_void_p(oy.ctypes.data), int(ox.size), (ox, oy)

    def _pick(self, dev, name):
        return getattr(self._lib, name + ("_dev" if dev else ""))

    # ---- stages ---------------------------------------------------------------------------
    def pitremove(self, dem, nodata=-9999.0, mask=None, fourway=False, out=None, stats=False):
        """fel = flood(dem)  (src/flood.cpp:50)."""
        ny, nx = dem.shape
        fel = out if out is not None else self._out(dem, np.float32, (ny, nx))
        pz, dev = self._ptr(dem, np.float32, name="dem")
        pm, mdev = self._ptr(mask, np.int16, (ny, nx), "mask")
        pf, fdev = self._ptr(fel, np.float32, (ny, nx), "fel")
        if fdev != dev or (mask is not None and mdev != dev):
            raise ValueError("all rasters must be on the same side (host or device)")
        st = TdxStats()
        self._sync_torch(dem, mask)
        check(self._pick(dev, "tdx_pitremove")(self._h, pz, nx, ny, float(nodata), pm, int(bool(fourway)), pf, C.byref(st)), self._h)
        return (fel, st.as_dict()) if stats else fel

    def d8flowdir(self, fel, nodata=float(FEL_NODATA), dx=1.0, dy=1.0, want_slope=True, out=None, stats=False):
        """p, sd8 = setdird8(fel)  (src/d8.cpp:181).  dx, dy: scalars or per-row arrays (metres)."""
        ny, nx = fel.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        p = out[0] if out is not None else self._out(fel, np.int16, (ny, nx))
        sd8 = (out[1] if out is not None else self._out(fel, np.float32, (ny, nx))) if want_slope else None
        pz, dev = self._ptr(fel, np.float32, name="fel")
        pp, pdev = self._ptr(p, np.int16, (ny, nx), "p")
        ps, _ = self._ptr(sd8, np.float32, (ny, nx), "sd8")
        if pdev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        st = TdxStats()
        self._sync_torch(fel)
        check(self._pick(dev, "tdx_d8flowdir")(self._h, pz, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data),
                                                pp, ps, C.byref(st)), self._h)
        res = (p, sd8) if want_slope else (p, None)
        return (res + (st.as_dict(),)) if stats else res

    def aread8(self, p, nodata=int(P_NODATA), weights=None, weights_nodata=-9999.0, contcheck=True, outlets=None, out=None, stats=False):
        """ad8 = aread8(p)  (src/aread8.cpp:56).  outlets: (columns, rows) global indices or None."""
        ny, nx = p.shape
        ad8 = out if out is not None else self._out(p, np.float32, (ny, nx))
        pp, dev = self._ptr(p, np.int16, name="p")
        pw, wdev = self._ptr(weights, np.float32, (ny, nx), "weights")
        pa, adev = self._ptr(ad8, np.float32, (ny, nx), "ad8")
        if adev != dev or (weights is not None and wdev != dev):
            raise ValueError("all rasters must be on the same side (host or device)")
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        self._sync_torch(p, weights)
        check(self._pick(dev, "tdx_aread8")(self._h, pp, nx, ny, int(nodata), pw, float(weights_nodata), int(bool(contcheck)), ox, oy, no, pa,
                                             C.byref(st)), self._h)
        del keep
        return (ad8, st.as_dict()) if stats else ad8

    def d8flowpathextremeup(self, p, sa, nodata=int(P_NODATA), usemax=True, contcheck=True, outlets=None, out=None, stats=False):
        """ssa = d8flowpathextremeup(p, sa)  (src/D8flowpathextremeup.cpp:58): upstream max / min of sa along D8 flow paths (nodata -FLT_MAX)."""
        ny, nx = p.shape
        ssa = out if out is not None else self._out(p, np.float32, (ny, nx))
        pp, dev = self._ptr(p, np.int16, name="p")
        pa, adev = self._ptr(sa, np.float32, (ny, nx), "sa")
        ps, sdev = self._ptr(ssa, np.float32, (ny, nx), "ssa")
        if adev != dev or sdev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        self._sync_torch(p, sa)
        check(self._pick(dev, "tdx_d8flowpathextremeup")(self._h, pp, nx, ny, int(nodata), pa, int(bool(usemax)), int(bool(contcheck)), ox, oy, no, ps,
                                                          C.byref(st)), self._h)
        del keep
        return (ssa, st.as_dict()) if stats else ssa

    def gridnet(self, p, nodata=int(P_NODATA), dx=1.0, dy=1.0, mask=None, thresh=0, outlets=None, stats=False):
        """plen, tlen, gord = gridnet(p)  (src/gridnet.cpp:54).  mask: int32 raster, cells with mask >= thresh are evaluated;
        outlets: (columns, rows) - only their upstream closure is evaluated."""
        ny, nx = p.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        plen = self._out(p, np.float32, (ny, nx))
        tlen = self._out(p, np.float32, (ny, nx))
        gord = self._out(p, np.int16, (ny, nx))
        pp, dev = self._ptr(p, np.int16, name="p")
        pm, mdev = self._ptr(mask, np.int32, (ny, nx), "mask")
        ppl, _ = self._ptr(plen, np.float32, (ny, nx), "plen")
        ptl, _ = self._ptr(tlen, np.float32, (ny, nx), "tlen")
        pgo, _ = self._ptr(gord, np.int16, (ny, nx), "gord")
        if mask is not None and mdev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        st = TdxStats()
        ox, oy, no, keep = self._outlets(outlets)
        self._sync_torch(p, mask)
        check(self._pick(dev, "tdx_gridnet")(self._h, pp, nx, ny, int(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), pm, int(thresh),
                                              ox, oy, no, ppl, ptl, pgo, C.byref(st)), self._h)
        del keep
        return (plen, tlen, gord, st.as_dict()) if stats else (plen, tlen, gord)

    def threshold(self, ssa, thresh, nodata=-1.0, mask=None, stats=False):
        """src = threshold(ssa)  (src/Threshold.cpp:49): 1 where ssa >= thresh (and mask >= 0), 0 elsewhere, -32768 where ssa is nodata."""
        ny, nx = ssa.shape
        src = self._out(ssa, np.int16, (ny, nx))
        pa, dev = self._ptr(ssa, np.float32, name="ssa")
        pm, mdev = self._ptr(mask, np.float32, (ny, nx), "mask")
        ps, _ = self._ptr(src, np.int16, (ny, nx), "src")
        if mask is not None and mdev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        st = TdxStats()
        self._sync_torch(ssa, mask)
        check(self._pick(dev, "tdx_threshold")(self._h, pa, nx, ny, float(nodata), pm, float(thresh), ps, C.byref(st)), self._h)
        return (src, st.as_dict()) if stats else src

    def dinfflowdir(self, fel, nodata=float(FEL_NODATA), dx=1.0, dy=1.0, out=None, stats=False):
        """ang, slp = setdir(fel)  (src/dinf.cpp:109)."""
        ny, nx = fel.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        ang = out[0] if out is not None else self._out(fel, np.float32, (ny, nx))
        slp = out[1] if out is not None else self._out(fel, np.float32, (ny, nx))
        pz, dev = self._ptr(fel, np.float32, name="fel")
        pa, adev = self._ptr(ang, np.float32, (ny, nx), "ang")
        ps, _ = self._ptr(slp, np.float32, (ny, nx), "slp")
        if adev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        st = TdxStats()
        self._sync_torch(fel)
        check(self._pick(dev, "tdx_dinfflowdir")(self._h, pz, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data),
                                                  pa, ps, C.byref(st)), self._h)
        return (ang, slp, st.as_dict()) if stats else (ang, slp)

    def areadinf(self, ang, nodata=float(ANG_NODATA), dx=1.0, dy=1.0, weights=None, contcheck=True, outlets=None, out=None, stats=False):
        """sca = area(ang)  (src/areadinf.cpp:53)."""
        ny, nx = ang.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        sca = out if out is not None else self._out(ang, np.float32, (ny, nx))
        pa, dev = self._ptr(ang, np.float32, name="ang")
        pw, wdev = self._ptr(weights, np.float32, (ny, nx), "weights")
        ps, sdev = self._ptr(sca, np.float32, (ny, nx), "sca")
        if sdev != dev or (weights is not None and wdev != dev):
            raise ValueError("all rasters must be on the same side (host or device)")
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        self._sync_torch(ang, weights)
        check(self._pick(dev, "tdx_areadinf")(self._h, pa, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), pw,
                                               int(bool(contcheck)), ox, oy, no, ps, C.byref(st)), self._h)
        del keep
        return (sca, st.as_dict()) if stats else sca

    def dinfdecayaccum(self, ang, dm, nodata=float(ANG_NODATA), dm_nodata=-9999.0, dx=1.0, dy=1.0, weights=None, contcheck=True,
                       outlets=None, out=None, stats=False):
        """dsca = dmarea(ang, dm)  (src/dinfdecayaccum.cpp:61)."""
        ny, nx = ang.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        dsca = out if out is not None else self._out(ang, np.float32, (ny, nx))
        pa, dev = self._ptr(ang, np.float32, name="ang")
        pd, ddev = self._ptr(dm, np.float32, (ny, nx), "dm")
        pw, wdev = self._ptr(weights, np.float32, (ny, nx), "weights")
        ps, sdev = self._ptr(dsca, np.float32, (ny, nx), "dsca")
        if sdev != dev or ddev != dev or (weights is not None and wdev != dev):
            raise ValueError("all rasters must be on the same side (host or device)")
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        self._sync_torch(ang, dm, weights)
        check(self._pick(dev, "tdx_dinfdecayaccum")(self._h, pa, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data),
                                                     pd, float(dm_nodata), pw, int(bool(contcheck)), ox, oy, no, ps, C.byref(st)), self._h)
        del keep
        return (dsca, st.as_dict()) if stats else dsca

    def dinfupdependence(self, ang, dg, nodata=float(ANG_NODATA), dx=1.0, dy=1.0, stats=False):
        """dep = depgrd(ang, dg)  (src/DinfUpDependence.cpp:52): dg int32, dep float32 (nodata -1)."""
        ny, nx = ang.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        dep = self._out(ang, np.float32, (ny, nx))
        pa, dev = self._ptr(ang, np.float32, name="ang")
        pg, gdev = self._ptr(dg, np.int32, (ny, nx), "dg")
        po, _ = self._ptr(dep, np.float32, (ny, nx), "dep")
        if gdev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        st = TdxStats()
        self._sync_torch(ang, dg)
        check(self._pick(dev, "tdx_dinfupdependence")(self._h, pa, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), pg, po,
                                                       C.byref(st)), self._h)
        return (dep, st.as_dict()) if stats else dep

    def dinfrevaccum(self, ang, w, nodata=float(ANG_NODATA), w_nodata=-9999.0, dx=1.0, dy=1.0, stats=False):
        """racc, dmax = dsaccum(ang, w)  (src/DinfRevAccum.cpp:51): float32, nodata -FLT_MAX."""
        ny, nx = ang.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        racc = self._out(ang, np.float32, (ny, nx))
        dmax = self._out(ang, np.float32, (ny, nx))
        pa, dev = self._ptr(ang, np.float32, name="ang")
        pw, wdev = self._ptr(w, np.float32, (ny, nx), "w")
        pr, _ = self._ptr(racc, np.float32, (ny, nx), "racc")
        pm, _ = self._ptr(dmax, np.float32, (ny, nx), "dmax")
        if wdev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        st = TdxStats()
        self._sync_torch(ang, w)
        check(self._pick(dev, "tdx_dinfrevaccum")(self._h, pa, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), pw,
                                                   float(w_nodata), pr, pm, C.byref(st)), self._h)
        return (racc, dmax, st.as_dict()) if stats else (racc, dmax)

    def dinfconclimaccum(self, ang, dm, dg, q, csol=1.0, nodata=float(ANG_NODATA), dm_nodata=-9999.0, q_nodata=-9999.0, dx=1.0, dy=1.0, contcheck=True,
                         outlets=None, stats=False):
        """ctpt = dsllArea(ang, dm, dg, q)  (src/DinfConcLimAccum.cpp:61): dg int16, ctpt float32 (nodata -FLT_MAX)."""
        ny, nx = ang.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        ctpt = self._out(ang, np.float32, (ny, nx))
        pa, dev = self._ptr(ang, np.float32, name="ang")
        pm, mdev = self._ptr(dm, np.float32, (ny, nx), "dm")
        pg, gdev = self._ptr(dg, np.int16, (ny, nx), "dg")
        pq, qdev = self._ptr(q, np.float32, (ny, nx), "q")
        po, _ = self._ptr(ctpt, np.float32, (ny, nx), "ctpt")
        if mdev != dev or gdev != dev or qdev != dev:
            raise ValueError("all rasters must be on the same side (host or device)")
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        self._sync_torch(ang, dm, dg, q)
        check(self._pick(dev, "tdx_dinfconclimaccum")(self._h, pa, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), pm, float(dm_nodata),
                                                       pg, pq, float(q_nodata), float(csol), int(bool(contcheck)), ox, oy, no, po, C.byref(st)), self._h)
        del keep
        return (ctpt, st.as_dict()) if stats else ctpt

    def dinftranslimaccum(self, ang, tsup, tc, cs=None, nodata=float(ANG_NODATA), tsup_nodata=-9999.0, tc_nodata=-9999.0, cs_nodata=-9999.0, dx=1.0, dy=1.0,
                          contcheck=True, outlets=None, stats=False):
        """tla, tdep, ctpt = tlaccum(ang, tsup, tc[, cs])  (src/DinfTransLimAccum.cpp:61): float32, nodata -FLT_MAX; ctpt is None without cs."""
        ny, nx = ang.shape
        dxc, dyc = _f64(dx, ny), _f64(dy, ny)
        tla = self._out(ang, np.float32, (ny, nx))
        dep = self._out(ang, np.float32, (ny, nx))
        cso = self._out(ang, np.float32, (ny, nx)) if cs is not None else None
        pa, dev = self._ptr(ang, np.float32, name="ang")
        ps, sdev = self._ptr(tsup, np.float32, (ny, nx), "tsup")
        pc, cdev = self._ptr(tc, np.float32, (ny, nx), "tc")
        pi, idev = self._ptr(cs, np.float32, (ny, nx), "cs")
        pt, _ = self._ptr(tla, np.float32, (ny, nx), "tla")
        pd, _ = self._ptr(dep, np.float32, (ny, nx), "tdep")
        po, _ = self._ptr(cso, np.float32, (ny, nx), "ctpt")
        if sdev != dev or cdev != dev or (cs is not None and idev != dev):
            raise ValueError("all rasters must be on the same side (host or device)")
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        self._sync_torch(ang, tsup, tc, cs)
        check(self._pick(dev, "tdx_dinftranslimaccum")(self._h, pa, nx, ny, float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), ps,
                                                        float(tsup_nodata), pc, float(tc_nodata), pi, float(cs_nodata), int(bool(contcheck)), ox, oy, no, pt, pd, po,
                                                        C.byref(st)), self._h)
        del keep
        return (tla, dep, cso, st.as_dict()) if stats else (tla, dep, cso)

    def synth_dem(self, n_or_shape, seed=1234, x0=0, y0=0, base_wavelength=None, out=None):
        """Seeded fractal DEM generated on the device (torch tensor on cuda:<device>)."""
        import torch

        if isinstance(n_or_shape, int):
            ny = nx = int(n_or_shape)
        else:
            ny, nx = n_or_shape
        if base_wavelength is None:
            base_wavelength = synth_base_wavelength(max(nx, ny))
        t = out if out is not None else torch.empty((ny, nx), dtype=torch.float32, device=f"cuda:{self.device}")
        torch.cuda.synchronize(self.device)
        check(self._lib.tdx_synth_dem_dev(self._h, int(seed), nx, ny, int(x0), int(y0), int(base_wavelength), C.c_void_p(t.data_ptr())), self._h)
        return t


def synth_base_wavelength(n: int) -> int:
    """Largest power of two < n (at least 2): tdx_synth_base_wl() of csrc/synth_dem.h."""
    wl = 2
    while wl * 2 < n:
        wl *= 2
    return wl


# ---- raster files ---------------------------------------------------------------------------
_NP2DT = {np.dtype(np.int16): _lib.TDX_DT_I16, np.dtype(np.int32): _lib.TDX_DT_I32, np.dtype(np.float32): _lib.TDX_DT_F32}


def raster_info(path):
    info = _lib.TdxRasterInfo()
    check(_lib.load().tdx_raster_info_read(str(path).encode(), C.byref(info)))
    return {
        "nx": info.nx, "ny": info.ny, "geotransform": tuple(info.geotransform), "nodata": info.nodata,
        "has_nodata": bool(info.has_nodata), "geographic": bool(info.geographic), "dxA": info.dxA, "dyA": info.dyA,
    }


def read_raster(path, dtype=np.float32):
    """Returns (array, info dict incl. per-row 'dxc'/'dyc')."""
    info = raster_info(path)
    a = np.empty((info["ny"], info["nx"]), dtype=dtype)
    dxc = np.empty(info["ny"], dtype=np.float64)
    dyc = np.empty(info["ny"], dtype=np.float64)
    check(_lib.load().tdx_raster_read(str(path).encode(), _NP2DT[np.dtype(dtype)], C.c_void_p(a.ctypes.data), C.c_void_p(dxc.ctypes.data),
                                      C.c_void_p(dyc.ctypes.data)))
    info["dxc"], info["dyc"] = dxc, dyc
    return a, info


def write_raster(path, a, nodata, like=None, geotransform=None, geographic=False, lzw=False):
    a = np.ascontiguousarray(a)
    ny, nx = a.shape
    lib = _lib.load()
    if like is not None:
        check(lib.tdx_raster_write(str(path).encode(), _NP2DT[a.dtype], C.c_void_p(a.ctypes.data), nx, ny, float(nodata), str(like).encode(), int(lzw)))
    else:
        gt = np.asarray(geotransform if geotransform is not None else (0.0, 1.0, 0.0, float(ny), 0.0, -1.0), dtype=np.float64)
        check(lib.tdx_raster_write_geo(str(path).encode(), _NP2DT[a.dtype], C.c_void_p(a.ctypes.data), nx, ny, float(nodata),
                                       C.c_void_p(gt.ctypes.data), int(bool(geographic)), int(lzw)))
