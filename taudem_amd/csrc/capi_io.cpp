// C ABI for raster files (host side): thin wrappers over geotiff.cpp.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/taudem_amd.h"
#include "geotiff.hpp"
#include "outlets.hpp"

extern thread_local std::string g_tdx_thread_error;

namespace {
tdx::DType to_dt(int t) { return t == TDX_DT_I16 ? tdx::DType::I16 : (t == TDX_DT_I32 ? tdx::DType::I32 : tdx::DType::F32); }
void fill_info(const tdx::RasterInfo& ri, tdx_raster_info* info) {
    info->nx = ri.nx; info->ny = ri.ny;
    memcpy(info->geotransform, ri.gt, sizeof ri.gt);
    info->nodata = ri.nodata; info->has_nodata = ri.has_nodata ? 1 : 0;
    info->geographic = ri.geographic ? 1 : 0;
    info->dxA = ri.dxA(); info->dyA = ri.dyA();
}
}  // namespace

extern "C" {

int tdx_raster_info_read(const char* path, tdx_raster_info* info) {
    if (!path || !info) return TDX_ERR_ARG;
    tdx::TiffReader rd;
    if (!rd.open(path)) { g_tdx_thread_error = rd.error(); return TDX_ERR_FILE; }
    fill_info(rd.info(), info);
    return TDX_OK;
}

int tdx_raster_read(const char* path, int dtype, void* data, double* dxc, double* dyc) {
    if (!path || !data) return TDX_ERR_ARG;
    tdx::TiffReader rd;
    if (!rd.open(path)) { g_tdx_thread_error = rd.error(); return TDX_ERR_FILE; }
    const tdx::RasterInfo& ri = rd.info();
    if (!rd.read_window(0, 0, ri.nx, ri.ny, to_dt(dtype), data)) { g_tdx_thread_error = rd.error(); return TDX_ERR_FILE; }
    if (dxc) memcpy(dxc, ri.dxc.data(), size_t(ri.ny) * sizeof(double));
    if (dyc) memcpy(dyc, ri.dyc.data(), size_t(ri.ny) * sizeof(double));
    return TDX_OK;
}

static int write_common(const char* path, int dtype, const void* data, int64_t nx, int64_t ny, double nodata,
                        const tdx::RasterInfo* georef, int lzw) {
    std::string name = path;
    if (tdx::resolve_output_name(name) != 0) { g_tdx_thread_error = "GDAL driver is not available"; return TDX_ERR_DRIVER; }
    tdx::TiffWriter wr;
    if (!wr.create(name, nx, ny, to_dt(dtype), nodata, georef, lzw != 0)) { g_tdx_thread_error = wr.error(); return TDX_ERR_FILE; }
    // TAUDEM_AMD_IO_THREADS host threads encode / write the rows (default 1; the bytes of the file do not depend on it)
    const char* e = getenv("TAUDEM_AMD_IO_THREADS");
    if (!wr.write_all(data, e ? std::max(1, std::min(atoi(e), 64)) : 1)) { g_tdx_thread_error = wr.error(); return TDX_ERR_FILE; }
    if (!wr.close()) { g_tdx_thread_error = wr.error(); return TDX_ERR_FILE; }
    return TDX_OK;
}

int tdx_raster_write(const char* path, int dtype, const void* data, int64_t nx, int64_t ny, double nodata,
                     const char* georef_from, int lzw) {
    if (!path || !data) return TDX_ERR_ARG;
    tdx::RasterInfo ri;
    const tdx::RasterInfo* g = nullptr;
    if (georef_from && georef_from[0]) {
        tdx::TiffReader rd;
        if (!rd.open(georef_from)) { g_tdx_thread_error = rd.error(); return TDX_ERR_FILE; }
        ri = rd.info();
        g = &ri;
    }
    return write_common(path, dtype, data, nx, ny, nodata, g, lzw);
}

int tdx_raster_write_geo(const char* path, int dtype, const void* data, int64_t nx, int64_t ny, double nodata,
                         const double* gt, int geographic, int lzw) {
    if (!path || !data) return TDX_ERR_ARG;
    tdx::RasterInfo ri;
    ri.nx = nx; ri.ny = ny;
    if (gt) {
        memcpy(ri.gt, gt, 6 * sizeof(double));
        ri.geo.pixel_scale = {gt[1], -gt[5], 0.0};
        ri.geo.tiepoints = {0, 0, 0, gt[0], gt[3], 0};
    }
    if (geographic) ri.geo.geokeys = {1, 1, 0, 2, 1024, 0, 1, 2, 1025, 0, 1, 1};
    ri.geographic = geographic != 0;
    return write_common(path, dtype, data, nx, ny, nodata, &ri, lzw);
}

int tdx_outlets_read(const char* path, double* x, double* y, int32_t* id, int64_t capacity, int64_t* count) {
    if (!path || !count) return TDX_ERR_ARG;
    std::vector<double> vx, vy;
    std::vector<int> vid;
    std::string err;
    if (!tdx::read_outlets(path, vx, vy, vid, err)) { g_tdx_thread_error = err; return TDX_ERR_OUTLETS; }
    *count = int64_t(vx.size());
    for (int64_t i = 0; i < int64_t(vx.size()) && i < capacity; i++) {
        if (x) x[i] = vx[size_t(i)];
        if (y) y[i] = vy[size_t(i)];
        if (id) id[i] = vid[size_t(i)];
    }
    return TDX_OK;
}

int tdx_outlets_to_cells(const char* rasterpath, const double* x, const double* y, int64_t n, int32_t* col, int32_t* row) {
    if (!rasterpath || !x || !y || !col || !row || n < 0) return TDX_ERR_ARG;
    tdx::TiffReader rd;
    if (!rd.open(rasterpath)) { g_tdx_thread_error = rd.error(); return TDX_ERR_FILE; }
    const tdx::RasterInfo ri = rd.info();
    for (int64_t i = 0; i < n; i++) {
        int gx, gy;
        tdx::geo_to_global_xy(x[i], y[i], ri.xleftedge, ri.ytopedge, ri.dlon, ri.dlat, gx, gy);
        col[i] = gx; row[i] = gy;
    }
    return TDX_OK;
}

}  // extern "C"
