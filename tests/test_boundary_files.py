"""The file boundary against INDEPENDENT encoders (no GPU needed):
  * outlet readers (taudem_amd/csrc/outlets.cpp, replaces src/ReadOutlets.cpp:49-198): ESRI shapefile, GeoJSON and text files
    written here byte by byte, read back through the C ABI, and mapped to cells like tiffIO::geoToGlobalXY (src/tiffIO.cpp:580-588);
  * GeoTIFF codec (taudem_amd/csrc/geotiff.cpp, replaces GDAL behind src/tiffIO.cpp): files written by libtiff 4.2 (tiled + Deflate +
    floating-point predictor, LZW strips + horizontal predictor, PackBits, BigTIFF) are decoded by our reader, and files written by
    our writer (LZW like src/tiffIO.cpp:316-318, plain, BigTIFF) are decoded by libtiff - pixel for pixel.
tests/tiffx/tiffx.c is the libtiff side; it is compiled on first use against /opt/conda (skipped where libtiff is absent)."""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import taudem_amd as T

HERE = os.path.dirname(os.path.abspath(__file__))
TIFFX_SRC = os.path.join(HERE, "tiffx", "tiffx.c")
CONDA = "/opt/conda"


# ---- outlet files ------------------------------------------------------------------------------------------------------
def write_shp(path, pts, shptype=1):
    """ESRI shapefile main file (.shp) with Point (1) / PointZ (11) / PointM (21) records."""
    recs = b""
    for i, (x, y) in enumerate(pts):
        body = struct.pack("<idd", shptype, x, y)
        if shptype == 11:
            body += struct.pack("<dd", 5.0, 0.0)
        if shptype == 21:
            body += struct.pack("<d", 7.0)
        recs += struct.pack(">ii", i + 1, len(body) // 2) + body
    xs, ys = [p[0] for p in pts], [p[1] for p in pts]
    hdr = struct.pack(">iiiiiii", 9994, 0, 0, 0, 0, 0, (100 + len(recs)) // 2) + struct.pack("<ii", 1000, shptype)
    hdr += struct.pack("<dddddddd", min(xs), min(ys), max(xs), max(ys), 0, 0, 0, 0)
    assert len(hdr) == 100
    open(path, "wb").write(hdr + recs)


def write_geojson(path, pts):
    feats = [{"type": "Feature", "properties": {"id": i + 1}, "geometry": {"type": "Point", "coordinates": [x, y]}} for i, (x, y) in enumerate(pts)]
    feats.insert(1, {"type": "Feature", "properties": {}, "geometry": {"type": "LineString", "coordinates": [[0.5, 0.5], [1.5, 1.5]]}})   # not a point: skipped
    json.dump({"type": "FeatureCollection", "features": feats}, open(path, "w"), indent=1)


def read_outlets(path):
    lib = T.load()
    n = C.c_int64()
    assert lib.tdx_outlets_read(path.encode(), None, None, None, 0, C.byref(n)) == 0
    x = np.empty(n.value); y = np.empty(n.value); i = np.empty(n.value, np.int32)
    assert lib.tdx_outlets_read(path.encode(), x.ctypes.data, y.ctypes.data, i.ctypes.data, n.value, C.byref(n)) == 0
    return x, y, i


PTS = [(1005.25, 5200.75), (2990.0, 6100.5), (-12.5, 40.0), (1234567.875, -7654321.125)]


@pytest.mark.parametrize("kind", ["shp", "shpz", "shpm", "geojson", "txt"])
def test_outlet_readers(tmp_path, kind):
    path = str(tmp_path / ("o." + {"shp": "shp", "shpz": "shp", "shpm": "SHP", "geojson": "geojson", "txt": "txt"}[kind]))
    if kind.startswith("shp"):
        write_shp(path, PTS, {"shp": 1, "shpz": 11, "shpm": 21}[kind])
    elif kind == "geojson":
        write_geojson(path, PTS)
    else:
        with open(path, "w") as f:
            f.write("# x y id\n")
            for i, (x, y) in enumerate(PTS):
                f.write(f"{x!r}, {y!r}, {10 + i}\n" if i % 2 else f"{x!r}\t{y!r}\n")
    x, y, ids = read_outlets(path)
    assert list(zip(x, y)) == PTS          # doubles survive exactly
    if kind == "txt":
        assert list(ids) == [1, 11, 3, 13]
    else:
        assert list(ids) == [1, 2, 3, 4]


def test_outlets_missing_file_is_error_5(tmp_path):
    lib = T.load()
    n = C.c_int64()
    assert lib.tdx_outlets_read(str(tmp_path / "nope.shp").encode(), None, None, None, 0, C.byref(n)) == 5   # src/aread8.cpp:125


def test_geo_to_cells_truncates_like_the_reference(tmp_path):
    lib = T.load()
    a = np.zeros((50, 80), np.float32)
    path = str(tmp_path / "r.tif")
    T.write_raster(path, a, -1.0, geotransform=(1000.0, 10.0, 0.0, 6000.0, 0.0, -25.0))
    x = np.array([1000.0, 1009.99, 1010.0, 1795.0, 995.0, 2000.0]); y = np.array([6000.0, 5975.01, 5975.0, 4751.0, 6010.0, 0.0])
    col = np.empty(6, np.int32); row = np.empty(6, np.int32)
    assert lib.tdx_outlets_to_cells(path.encode(), x.ctypes.data, y.ctypes.data, 6, col.ctypes.data, row.ctypes.data) == 0
    # (int)((x - xleft) / dx), (int)((ytop - y) / dy): truncation towards zero, no bounds check (src/tiffIO.cpp:580-588)
    assert list(col) == [0, 0, 1, 79, 0, 100] and list(row) == [0, 0, 1, 49, 0, 240]


# ---- GeoTIFF against libtiff -----------------------------------------------------------------------------------------------
@pytest.fixture(scope="session")
def tiffx(tmp_path_factory):
    if not os.path.exists(os.path.join(CONDA, "include", "tiffio.h")):
        pytest.skip("libtiff headers not present")
    exe = str(tmp_path_factory.mktemp("tiffx") / "tiffx")
    r = subprocess.run(["gcc", "-O2", "-o", exe, TIFFX_SRC, f"-I{CONDA}/include", f"-L{CONDA}/lib", "-ltiff", f"-Wl,-rpath,{CONDA}/lib:/usr/lib/x86_64-linux-gnu"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the libtiff helper: " + r.stderr[-400:])
    return exe


def expected(nx, ny, dtype):
    x, y = np.meshgrid(np.arange(nx, dtype=np.int64), np.arange(ny, dtype=np.int64))
    v = ((x * 31 + y * 17) % 1000) * 0.25 - 50.0 + ((x ^ y) & 7)
    return v.astype(np.float32) if dtype == np.float32 else np.trunc(v).astype(dtype)


@pytest.mark.parametrize("kind,dtype", [("f32_tiled_deflate_pred3", np.float32), ("i16_strip_lzw_pred2", np.int16), ("i32_tiled_none", np.int32),
                                        ("f32_strip_packbits", np.float32), ("f32_big_tiled_deflate", np.float32), ("f32_strip_lzw", np.float32)])
def test_our_reader_decodes_libtiff_files(tmp_path, tiffx, kind, dtype):
    nx, ny = 211, 157   # not multiples of the 64 x 48 tiles or of the 7-row strips
    path = str(tmp_path / (kind + ".tif"))
    assert subprocess.run([tiffx, "write", path, kind, str(nx), str(ny)]).returncode == 0
    a, info = T.read_raster(path, dtype)
    assert info["nx"] == nx and info["ny"] == ny
    assert np.array_equal(a, expected(nx, ny, dtype))


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.int32])
@pytest.mark.parametrize("lzw", [True, False])
def test_libtiff_decodes_our_files(tmp_path, tiffx, dtype, lzw):
    nx, ny = 301, 173
    a = expected(nx, ny, dtype)
    path, raw = str(tmp_path / "ours.tif"), str(tmp_path / "ours.raw")
    T.write_raster(path, a, -9999.0 if dtype == np.float32 else -32768, geotransform=(10.0, 2.0, 0.0, 500.0, 0.0, -2.0), lzw=lzw)
    assert subprocess.run([tiffx, "dump", path, raw]).returncode == 0
    b = np.fromfile(raw, dtype=dtype).reshape(ny, nx)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("lzw", [True, False])
def test_row_bands_written_by_several_threads_give_the_same_file(tmp_path, monkeypatch, lzw):
    """TiffWriter::write_all: strips encoded / row bands written side by side (what `--gpus N` does for the tools' outputs, like the ranks
    of src/tiffIO.cpp:382-427) - the bytes of the file do not depend on the number of threads, and the file reads back."""
    nx, ny = 3001, 1003   # 12 KB rows -> 87-row strips, a ragged last strip
    a = expected(nx, ny, np.float32)
    files = []
    for threads in (None, "5", "64"):
        if threads is None:
            monkeypatch.delenv("TAUDEM_AMD_IO_THREADS", raising=False)
        else:
            monkeypatch.setenv("TAUDEM_AMD_IO_THREADS", threads)
        path = str(tmp_path / f"t{threads}.tif")
        T.write_raster(path, a, -9999.0, geotransform=(10.0, 2.0, 0.0, 500.0, 0.0, -2.0), lzw=lzw)
        files.append(open(path, "rb").read())
        b, info = T.read_raster(path, np.float32)
        assert np.array_equal(a, b)
    assert files[0] == files[1] == files[2]
