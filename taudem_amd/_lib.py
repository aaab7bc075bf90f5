"""ctypes binding of libtaudem_amd.so (the C ABI declared in include/taudem_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C taudem_amd/csrc``.  There is
no CPU fallback: if the library is missing the import fails loudly, and compute entry points fail
with ``TDX_ERR_NOGPU`` when no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtaudem_amd.so")

TDX_OK = 0
TDX_ERR_MISMATCH = 1
TDX_ERR_OUTLETS = 5
TDX_ERR_FILE = 21
TDX_ERR_DRIVER = 22
TDX_ERR_ARG = -1
TDX_ERR_HIP = -2
TDX_ERR_NOGPU = -3
TDX_ERR_NOMEM = -999

TDX_DT_I16, TDX_DT_I32, TDX_DT_F32 = 0, 1, 2

K_STENCIL, K_RELAX, K_BFS, K_FLATDIR, K_ACCUM, K_MISC, K_TILEK = 0, 1, 2, 3, 4, 5, 6
KERNEL_CLASSES = {"stencil": K_STENCIL, "relax": K_RELAX, "bfs": K_BFS, "flatdir": K_FLATDIR, "accum": K_ACCUM, "misc": K_MISC, "tilek": K_TILEK}


class TdxStats(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double),
        ("ms_kernel", C.c_double * 8),
        ("launches", C.c_int64 * 8),
        ("rounds", C.c_int64),
        ("flats_initial", C.c_int64),
        ("flats_left", C.c_int64),
        ("flat_iterations", C.c_int64),
        ("levels_fall", C.c_int64),
        ("levels_rise", C.c_int64),
        ("cells_evaluated", C.c_int64),
        ("levels_fall_max", C.c_int64),
        ("levels_rise_max", C.c_int64),
    ]

    def as_dict(self):
        d = {
            "ms_total": self.ms_total,
            "rounds": self.rounds,
            "flats_initial": self.flats_initial,
            "flats_left": self.flats_left,
            "flat_iterations": self.flat_iterations,
            "levels_fall": self.levels_fall,
            "levels_rise": self.levels_rise,
            "cells_evaluated": self.cells_evaluated,
            "levels_fall_max": self.levels_fall_max,
            "levels_rise_max": self.levels_rise_max,
        }
        for name, k in KERNEL_CLASSES.items():
            d["ms_" + name] = self.ms_kernel[k]
            d["launches_" + name] = self.launches[k]
        return d


class TdxSegment(C.Structure):
    """struct tdx_segment of include/taudem_amd.h (segment trace of a strip run)."""
    _fields_ = [("stage", C.c_char * 24), ("phase", C.c_char * 24), ("kind", C.c_int32), ("device_ms", C.c_float), ("wall_ms", C.c_float)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32)


class TdxComm(C.Structure):
    """struct tdx_comm of include/taudem_amd.h (row strips across GPUs)."""
    _fields_ = [
        ("rank", C.c_int32),
        ("size", C.c_int32),
        ("user", C.c_void_p),
        ("exchange", EXCHANGE_FN),
        ("allreduce", ALLREDUCE_FN),
        ("send_up", C.c_void_p),
        ("send_down", C.c_void_p),
        ("recv_up", C.c_void_p),
        ("recv_down", C.c_void_p),
        ("capacity", C.c_uint64),
        ("flags", C.c_uint64),              # 0: host-synchronous contract (this Python transport)
        ("allreduce_dev", C.c_void_p),      # NULL: votes travel through `allreduce` on host values
    ]


class TdxRasterInfo(C.Structure):
    _fields_ = [
        ("nx", C.c_int64),
        ("ny", C.c_int64),
        ("geotransform", C.c_double * 6),
        ("nodata", C.c_double),
        ("has_nodata", C.c_int32),
        ("geographic", C.c_int32),
        ("dxA", C.c_double),
        ("dyA", C.c_double),
    ]


# name -> (restype, argtypes): every symbol include/taudem_amd.h declares
_P = C.c_void_p
_I64 = C.c_int64
_F = C.c_float
_SIGNATURES = {
    "tdx_context_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "tdx_context_destroy": (None, [_P]),
    "tdx_context_release_scratch": (C.c_int, [_P]),
    "tdx_last_error": (C.c_char_p, [_P]),
    "tdx_synchronize": (C.c_int, [_P]),
    "tdx_stream": (_P, [_P]),
    "tdx_version": (C.c_char_p, []),
    "tdx_context_set_option": (C.c_int, [_P, C.c_char_p, _I64]),
    "tdx_device_count": (C.c_int, []),
    "tdx_device_alloc": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "tdx_device_free": (C.c_int, [_P, _P]),
    "tdx_copy_to_device": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "tdx_copy_to_host": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "tdx_pitremove_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, C.c_int, _P, _P]),
    "tdx_pitremove": (C.c_int, [_P, _P, _I64, _I64, _F, _P, C.c_int, _P, _P]),
    "tdx_d8flowdir_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_d8flowdir": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_aread8_dev": (C.c_int, [_P, _P, _I64, _I64, C.c_int16, _P, _F, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_aread8": (C.c_int, [_P, _P, _I64, _I64, C.c_int16, _P, _F, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_d8flowpathextremeup_dev": (C.c_int, [_P, _P, _I64, _I64, C.c_int16, _P, C.c_int, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_d8flowpathextremeup": (C.c_int, [_P, _P, _I64, _I64, C.c_int16, _P, C.c_int, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_d8flowpathextremeup_strip": (C.c_int, [_P, _P, _P, _I64, _I64, C.c_int16, _P, C.c_int, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_gridnet_dev": (C.c_int, [_P, _P, _I64, _I64, C.c_int16, _P, _P, _P, C.c_int32, _P, _P, _I64, _P, _P, _P, _P]),
    "tdx_gridnet": (C.c_int, [_P, _P, _I64, _I64, C.c_int16, _P, _P, _P, C.c_int32, _P, _P, _I64, _P, _P, _P, _P]),
    "tdx_gridnet_strip": (C.c_int, [_P, _P, _P, _I64, _I64, C.c_int16, _P, _P, _P, C.c_int32, _P, _P, _I64, _P, _P, _P, _P]),
    "tdx_threshold_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _F, _P, _P]),
    "tdx_threshold": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _F, _P, _P]),
    "tdx_dinfflowdir_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_dinfflowdir": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_areadinf_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_areadinf": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_dinfdecayaccum_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_dinfdecayaccum": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_pitremove_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, C.c_int, _P, _P]),
    "tdx_d8flowdir_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_aread8_strip": (C.c_int, [_P, _P, _P, _I64, _I64, C.c_int16, _P, _F, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_dinfflowdir_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_areadinf_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_dinfdecayaccum_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_dinfupdependence_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_dinfupdependence": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_dinfupdependence_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, _P, _P]),
    "tdx_dinfrevaccum_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _P, _P]),
    "tdx_dinfrevaccum": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _P, _P]),
    "tdx_dinfrevaccum_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _P, _P]),
    "tdx_tool_dinfupdependence": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p]),
    "tdx_tool_dinfrevaccum": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    # ctx, ang, nx, ny, ang_nodata, dxc, dyc, dm, dm_nodata, dg, q, q_nodata, csol, contcheck, ox, oy, n_outlets, ctpt, stats
    "tdx_dinfconclimaccum": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _P, _F, _F, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_dinfconclimaccum_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _P, _F, _F, C.c_int, _P, _P, _I64, _P, _P]),
    "tdx_dinfconclimaccum_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _P, _F, _F, C.c_int, _P, _P, _I64, _P, _P]),
    # ctx, ang, nx, ny, ang_nodata, dxc, dyc, tsup, tsup_nodata, tc, tc_nodata, cs, cs_nodata, contcheck, ox, oy, n_outlets, tla, tdep, ctpt, stats
    "tdx_dinftranslimaccum": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _F, _P, _F, C.c_int, _P, _P, _I64, _P, _P, _P, _P]),
    "tdx_dinftranslimaccum_dev": (C.c_int, [_P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _F, _P, _F, C.c_int, _P, _P, _I64, _P, _P, _P, _P]),
    "tdx_dinftranslimaccum_strip": (C.c_int, [_P, _P, _P, _I64, _I64, _F, _P, _P, _P, _F, _P, _F, _P, _F, C.c_int, _P, _P, _I64, _P, _P, _P, _P]),
    "tdx_tool_dinfconclimaccum": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_float]),
    "tdx_tool_dinftranslimaccum": (C.c_int, [C.c_char_p] * 9 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tdx_synth_dem_dev": (C.c_int, [_P, C.c_uint64, _I64, _I64, _I64, _I64, _I64, _P]),
    "tdx_raster_info_read": (C.c_int, [C.c_char_p, C.POINTER(TdxRasterInfo)]),
    "tdx_raster_read": (C.c_int, [C.c_char_p, C.c_int, _P, _P, _P]),
    "tdx_raster_write": (C.c_int, [C.c_char_p, C.c_int, _P, _I64, _I64, C.c_double, C.c_char_p, C.c_int]),
    "tdx_raster_write_geo": (C.c_int, [C.c_char_p, C.c_int, _P, _I64, _I64, C.c_double, _P, C.c_int, C.c_int]),
    "tdx_tool_pitremove": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "tdx_tool_d8flowdir": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "tdx_tool_aread8": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "tdx_tool_dinfflowdir": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "tdx_tool_areadinf": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "tdx_tool_dinfdecayaccum": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "tdx_tool_gridnet": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tdx_tool_d8flowpathextremeup": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tdx_tool_threshold": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_float, C.c_int]),
    "tdx_outlets_read": (C.c_int, [C.c_char_p, _P, _P, _P, _I64, C.POINTER(_I64)]),
    "tdx_outlets_to_cells": (C.c_int, [C.c_char_p, _P, _P, _I64, _P, _P]),
    "tdx_tool_set_device": (C.c_int, [C.c_int]),
    "tdx_tool_set_gpus": (C.c_int, [C.c_int]),
    "tdx_rccl_unique_id": (C.c_int, [_P]),
    "tdx_rccl_comm_create": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _I64, C.POINTER(_P)]),
    "tdx_rccl_comm_handle": (_P, [_P]),
    "tdx_rccl_comm_counters": (None, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "tdx_rccl_comm_destroy": (None, [_P]),
    "tdx_rccl_selftest": (C.c_int, [_P]),
    "tdx_comm_latency": (C.c_int, [_P, _P, C.c_int32, C.c_uint64, C.POINTER(C.c_double)]),
    "tdx_rccl_latency": (C.c_int, [_P, C.c_int32, C.c_uint64, C.POINTER(C.c_double)]),
    "tdx_context_comm_counters": (None, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tdx_context_segments": (_I64, [_P, C.POINTER(TdxSegment), _I64]),
    "tdx_group_create": (C.c_int, [C.c_int32, _P, _I64, C.POINTER(_P)]),
    "tdx_group_context": (_P, [_P, C.c_int32]),
    "tdx_group_comm": (_P, [_P, C.c_int32]),
    "tdx_group_transport": (C.c_char_p, [_P]),
    "tdx_group_abort": (None, [_P]),
    "tdx_group_destroy": (None, [_P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the shared library (once).  Raises ImportError with a build hint if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C taudem_amd/csrc` (hipcc, --offload-arch=gfx950). taudem_amd has no CPU fallback."
        )
    # One HIP/HSA runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.
    # If torch is importable, load it FIRST so that this library binds to the already-loaded runtime
    # (same SONAME) instead of pulling /opt/rocm's copy in beside it - with two HSA runtimes in one
    # process whichever initialises second sees "no HIP GPUs".  torch is plumbing here (device memory,
    # streams, torch.distributed), not a dependency of the C ABI itself.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less hosts use the system ROCm runtime
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class TdxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"taudem_amd error {code}: {msg}")
        self.code = code


def last_error(ctx=None):
    s = load().tdx_last_error(ctx)
    return s.decode("utf-8", "replace") if s else ""


def check(rc, ctx=None):
    if rc != TDX_OK:
        raise TdxError(rc, last_error(ctx) or last_error(None))
