"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the header
declares, there is no CPU fallback, GeoTIFF I/O round-trips, and the command-line mains keep the
reference's flag surface."""
import os
import re
import subprocess

import numpy as np
import pytest

import taudem_amd as T
from taudem_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taudem_amd", "bin")


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "taudem_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tdx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    lib = T.load()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"libtaudem_amd.so does not export {s}"
        assert s in _lib.EXPORTED_SYMBOLS, f"{s} has no ctypes signature in taudem_amd/_lib.py"
    assert set(_lib.EXPORTED_SYMBOLS) <= set(syms), "ctypes binds symbols the header does not declare"


def test_version_string():
    assert b"gfx950" in T.load().tdx_version()


def test_no_cpu_fallback():
    lib = T.load()
    if lib.tdx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(T.TdxError) as e:
        T.Context(0)
    assert e.value.code == _lib.TDX_ERR_NOGPU
    assert "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("dtype,nodata", [(np.float32, -9999.0), (np.int16, -32768), (np.int32, -1)])
@pytest.mark.parametrize("lzw", [False, True])
def test_geotiff_round_trip(tmp_path, dtype, nodata, lzw):
    rng = np.random.default_rng(1)
    a = (rng.random((37, 91)) * 1000).astype(dtype)
    a[3, 4] = nodata
    path = str(tmp_path / "a.tif")
    gt = (500000.0, 30.0, 0.0, 4600000.0, 0.0, -30.0)
    T.write_raster(path, a, nodata, geotransform=gt, lzw=lzw)
    b, info = T.read_raster(path, dtype)
    assert np.array_equal(a, b)
    assert info["nx"] == 91 and info["ny"] == 37 and info["has_nodata"] and info["nodata"] == nodata
    assert tuple(info["geotransform"]) == gt and not info["geographic"]
    assert np.all(info["dxc"] == 30.0) and np.all(info["dyc"] == 30.0)
    # georeferencing is copied from the input like tiffIO's copy constructor (src/tiffIO.cpp:344-349)
    path2 = str(tmp_path / "b.tif")
    T.write_raster(path2, a, nodata, like=path, lzw=lzw)
    assert tuple(T.raster_info(path2)["geotransform"]) == gt


def test_geotiff_type_conversion_on_read(tmp_path):
    a = np.arange(12, dtype=np.int16).reshape(3, 4)
    path = str(tmp_path / "i.tif")
    T.write_raster(path, a, -32768)
    f, _ = T.read_raster(path, np.float32)      # GDALRasterIO converts to the requested type (src/tiffIO.cpp:255)
    assert f.dtype == np.float32 and np.array_equal(f, a.astype(np.float32))


def test_missing_file_is_error_21(tmp_path):
    from taudem_amd import tools

    assert tools.flood(str(tmp_path / "nope.tif"), str(tmp_path / "out.tif")) == _lib.TDX_ERR_FILE   # MPI_Abort(MCW, 21), src/tiffIO.cpp:69


@pytest.mark.parametrize("tool", ["pitremove", "d8flowdir", "aread8", "dinfflowdir", "areadinf", "dinfdecayaccum", "gridnet", "threshold",
                                  "d8flowpathextremeup"])
def test_cli_usage(tool):
    exe = os.path.join(BIN, tool)
    assert os.path.exists(exe), "build with __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0                       # the reference prints the usage and exit(0)s (e.g. src/aread8mn.cpp:150-176)
    assert "use" in r.stdout.lower() or "usage" in r.stdout.lower()
    r = subprocess.run([exe, "-bogus", "x"], capture_output=True, text=True)
    assert r.returncode == 0 and ("use" in r.stdout.lower() or "usage" in r.stdout.lower())


def test_gridnet_mask_needs_thresh():
    """-thresh has to follow the mask file immediately (src/gridnetmn.cpp:150-166): anything else is a usage error."""
    r = subprocess.run([os.path.join(BIN, "gridnet"), "-p", "p.tif", "-plen", "a.tif", "-tlen", "b.tif", "-gord", "c.tif", "-mask", "m.tif"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "usage" in r.stdout.lower()


def test_new_tools_report_missing_files(tmp_path):
    from taudem_amd import tools

    nope = str(tmp_path / "nope.tif")
    assert tools.gridnet(nope, nope, nope, nope) == _lib.TDX_ERR_FILE
    assert tools.threshold(nope, nope) == _lib.TDX_ERR_FILE
    assert tools.d8flowpathextremeup(nope, nope, nope) == _lib.TDX_ERR_FILE
    assert tools.gridnet(nope, nope, nope, nope, useOutlets=1) == _lib.TDX_ERR_FILE


def test_critical_path_projection_from_segment_traces():
    """taudem_amd.distributed.project_critical_path: sum over the segments between collectives of the slowest rank + collectives x assumed latencies
    (the role of the outer loops with share() / MPI_Allreduce, src/aread8.cpp:282-303, src/linearpart.h:313-384); a rank that went through another
    sequence of collectives is an error (the protocol is rank-symmetric)."""
    import pytest

    from taudem_amd.distributed import project_critical_path

    # (stage, phase, kind, device_ms, wall_ms): kind 0 = ended by an exchange, 1 = by an all-reduce, 2 = by the end of the call
    r0 = [("pitremove", "", 0, 1.0, 1.5), ("pitremove", "fine level", 1, 4.0, 4.0), ("pitremove", "fine level", 2, 0.5, 0.5), ("aread8", "forest", 0, 2.0, 2.0), ("aread8", "", 2, 0.1, 0.1)]
    r1 = [("pitremove", "", 0, 2.0, 2.0), ("pitremove", "fine level", 1, 1.0, 1.0), ("pitremove", "fine level", 2, 0.7, 0.7), ("aread8", "forest", 0, 3.0, 3.5), ("aread8", "", 2, 0.1, 0.2)]
    p = project_critical_path([r0, r1], exchange_us=10.0, vote_us=30.0)
    pit, ad8 = p["per_stage"]["pitremove"], p["per_stage"]["aread8"]
    assert pit["work_ms"] == pytest.approx(2.0 + 4.0 + 0.7) and pit["latency_ms"] == pytest.approx(0.010 + 0.030)
    assert (pit["segments"], pit["exchanges"], pit["allreduces"]) == (3, 1, 1)
    assert pit["sum_over_ranks_ms"] == pytest.approx(1.5 + 4.0 + 0.5 + 2.0 + 1.0 + 0.7)
    assert pit["phases"]["fine level"] == pytest.approx(4.0 + 0.030 + 0.7)
    assert ad8["ms"] == pytest.approx(3.5 + 0.010 + 0.2)
    assert p["total_ms"] == pytest.approx(pit["ms"] + ad8["ms"])
    assert project_critical_path([r0, r1], use="device")["per_stage"]["aread8"]["work_ms"] == pytest.approx(3.0 + 0.1)
    with pytest.raises(ValueError):
        project_critical_path([r0, r1[:-1]])
    bad = list(r1); bad[1] = ("pitremove", "fine level", 0, 1.0, 1.0)      # an exchange where rank 0 voted
    with pytest.raises(ValueError):
        project_critical_path([r0, bad])
