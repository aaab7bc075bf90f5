import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer-running case")


def golden_cases():
    return sorted(n for n in (os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "case_*.npz"))) if not n.endswith("_gridnet") and not n.endswith("_flowalg"))


def load_golden_gridnet(name):
    """GridNet / Threshold rasters of the reference for the D8 rasters of case `name` (tests/golden/make_golden_gridnet.py)."""
    g = np.load(os.path.join(GOLDEN_DIR, f"case_{name}_gridnet.npz"), allow_pickle=False)
    return {k: g[k] for k in g.files}


def load_golden_flowalg(name):
    """DinfUpDependence / DinfRevAccum rasters of the reference for the angles of case `name` (tests/golden/make_golden_flowalg.py)."""
    g = np.load(os.path.join(GOLDEN_DIR, f"case_{name}_flowalg.npz"), allow_pickle=False)
    return {k: g[k] for k in g.files}


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, f"case_{name}.npz"), allow_pickle=False)
    return {k: g[k] for k in g.files}


def outlets_to_indices(g):
    """tiffIO::geoToGlobalXY (src/tiffIO.cpp:580-588) for the golden case's geotransform."""
    ny, nx = g["dem"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    if bool(g["geographic"]):
        x0, y0 = -111.9, 41.9
    else:
        x0, y0 = 1000.0, 5000.0 + dy * ny
    xs, ys = g["outlet_xy"]
    ox = ((xs - x0) / dx).astype(np.int64).astype(np.int32)   # (int) truncation
    oy = ((y0 - ys) / dy).astype(np.int64).astype(np.int32)
    return ox, oy


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def describe_diff(a, b, name=""):
    a = np.asarray(a); b = np.asarray(b)
    neq = (a.view(np.uint32) != b.view(np.uint32)) if a.dtype == np.float32 else (a != b)
    idx = np.argwhere(neq)
    head = ", ".join(f"({y},{x}): {a[y, x]!r} vs {b[y, x]!r}" for y, x in idx[:6])
    return f"{name}: {len(idx)} of {a.size} cells differ; first: {head}"


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def ctx():
    import taudem_amd

    c = taudem_amd.Context(0)   # raises TdxError(TDX_ERR_NOGPU) without a device: no CPU fallback
    yield c
    c.close()
