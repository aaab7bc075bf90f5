// File-level tool functions: same argument lists, banners, timing blocks and error codes as the
// reference's tool functions, with the compute part on the GPU.  `tool --gpus N` / TAUDEM_AMD_GPUS=N / tdx_tool_set_gpus(N)
// partitions the raster into N row strips, one GPU (and one host thread) each, exchanging boundary rows over RCCL
// (tool_strips.hpp, comm.cpp) - the place of `mpiexec -n N` in the reference.
//   tdx_tool_pitremove       <- flood()     src/flood.cpp:50-526
//   tdx_tool_d8flowdir       <- setdird8()  src/d8.cpp:181-355
//   tdx_tool_aread8          <- aread8()    src/aread8.cpp:56-322
//   tdx_tool_dinfflowdir     <- setdir()    src/dinf.cpp:109-284
//   tdx_tool_areadinf        <- area()      src/areadinf.cpp:53-300
//   tdx_tool_dinfdecayaccum  <- dmarea()    src/dinfdecayaccum.cpp:61-324
//   tdx_tool_dinfconclimaccum  <- dsllArea()  src/DinfConcLimAccum.cpp:61-326
//   tdx_tool_dinftranslimaccum <- tlaccum()   src/DinfTransLimAccum.cpp:61-372
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "context.hpp"
#include "geotiff.hpp"
#include "outlets.hpp"
#include "tool_strips.hpp"

#define TDVERSION "5.4.0"   /* src/commonLib.h:63 */

namespace {

int g_tool_device = -1;
int g_tool_gpus = -1;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int tool_device() {
    if (g_tool_device >= 0) return g_tool_device;
    const char* e = getenv("TAUDEM_AMD_DEVICE");
    return e ? atoi(e) : 0;
}
int tool_gpus() {
    int n = g_tool_gpus;
    if (n < 1) { const char* e = getenv("TAUDEM_AMD_GPUS"); n = e ? atoi(e) : 1; }
    return n < 1 ? 1 : n;
}
void report(tdx_context* c) { fprintf(stderr, "taudem_amd: %s\n", tdx_last_error(c)); }
bool want_lzw() {
    const char* e = getenv("TAUDEM_AMD_COMPRESS");
    if (e && (strcmp(e, "NONE") == 0 || strcmp(e, "none") == 0)) return false;
    return true;   // the reference writes COMPRESS=LZW (src/tiffIO.cpp:316-318)
}

struct Raster {
    tdx::RasterInfo info;
    std::vector<float> f;
    std::vector<int16_t> s;
    std::vector<int32_t> l;
};

// tiffIO constructor + CreateNewPartition prints + read (src/tiffIO.cpp:54-183, src/createpart.h:49-87)
// Reads rows [y0, y1) of an open raster into the full-size host array `base`.
static bool read_rows(tdx::TiffReader& rd, tdx::DType type, void* base, int64_t nx, int64_t y0, int64_t y1) {
    return rd.read_window(0, y0, nx, y1 - y0, type, static_cast<char*>(base) + size_t(y0) * size_t(nx) * tdx::dtype_size(type));
}

int load_raster(const char* path, tdx::DType type, Raster& r) {
    tdx::TiffReader rd;
    if (!rd.open(path)) {
        printf("Error opening file %s.\n", path);
        fflush(stdout);
        g_tdx_thread_error = rd.error();
        return TDX_ERR_FILE;
    }
    r.info = rd.info();
    if (!r.info.geographic) printf("Input file %s has projected coordinate system.\n", path);
    else printf("Input file %s has geographic coordinate system.\n", path);
    const size_t n = size_t(r.info.nx) * size_t(r.info.ny);
    printf("Nodata value input to create partition from file: %lf\n", r.info.nodata);
    void* base;
    if (type == tdx::DType::F32) {
        printf("Nodata value recast to float used in partition raster: %f\n", (float)r.info.nodata);
        r.f.resize(n);
        base = r.f.data();
    } else if (type == tdx::DType::I32) {
        printf("Nodata value recast to int32_t used in partition raster: %d\n", (int32_t)r.info.nodata);
        r.l.resize(n);
        base = r.l.data();
    } else {
        printf("Nodata value recast to int16_t used in partition raster: %d\n", (int16_t)r.info.nodata);
        r.s.resize(n);
        base = r.s.data();
    }
    fflush(stdout);
    // With --gpus N every rank reads its OWN rows - the strip partition of the compute (src/linearpart.h:133-134), each through its own
    // reader on the file, like the ranks of the reference (src/tiffIO.cpp:186-290) - instead of one thread decoding a 17 GB raster.
    const int nrd = int(std::min<int64_t>(std::min(tool_gpus(), 64), r.info.ny));
    bool ok = true;
    std::string err;
    if (nrd <= 1) {
        ok = read_rows(rd, type, base, r.info.nx, 0, r.info.ny);
        if (!ok) err = rd.error();
    } else {
        rd.close();
        std::vector<std::thread> th;
        std::vector<std::string> errs(static_cast<size_t>(nrd));
        const int64_t rows = r.info.ny / nrd;
        for (int k = 0; k < nrd; k++)
            th.emplace_back([&, k] {
                tdx::TiffReader mine;
                const int64_t y0 = int64_t(k) * rows, y1 = (k == nrd - 1) ? r.info.ny : int64_t(k + 1) * rows;
                if (!mine.open(path) || !read_rows(mine, type, base, r.info.nx, y0, y1)) errs[size_t(k)] = mine.error().empty() ? "read failed" : mine.error();
            });
        for (auto& t : th) t.join();
        for (const std::string& e : errs) if (!e.empty()) { ok = false; err = e; break; }
    }
    if (!ok) { printf("Error opening file %s.\n", path); g_tdx_thread_error = err; return TDX_ERR_FILE; }
    return TDX_OK;
}

// tiffIO::compareTiff (src/tiffIO.cpp:449-541)
bool compare_rasters(const tdx::RasterInfo& a, const char* an, const tdx::RasterInfo& b, const char* bn) {
    const double tol = 0.0001;
    if (a.nx != b.nx) { printf("Columns do not match: %d %d\n", int(a.nx), int(b.nx)); return false; }
    if (a.ny != b.ny) { printf("Rows do not match: %d %d\n", int(a.ny), int(b.ny)); return false; }
    if (std::fabs(a.dxA() - b.dxA()) > tol) { printf("dx does not match: %lf %lf\n", a.dxA(), b.dxA()); return false; }
    if (std::fabs(a.dyA() - b.dyA()) > tol) { printf("dy does not match: %lf %lf\n", a.dyA(), b.dyA()); return false; }
    if (std::fabs(a.xleftedge - b.xleftedge) > 0.0) {
        printf("Warning! Left edge does not match exactly:\n %lf in file %s\n %lf in file %s\n", a.xleftedge, an, b.xleftedge, bn);
    }
    if (std::fabs(a.ytopedge - b.ytopedge) > 0.0) {
        printf("Warning! Top edge does not match exactly:\n %lf in file %s\n %lf in file %s\n", a.ytopedge, an, b.ytopedge, bn);
    }
    return true;
}

int save_raster(const char* path, tdx::DType type, const void* data, const tdx::RasterInfo& like, double nodata) {
    std::string name = path;
    if (tdx::resolve_output_name(name) != 0) { printf("GDAL driver is not available\n"); fflush(stdout); return TDX_ERR_DRIVER; }
    const size_t cb = tdx::dtype_size(type);
    const double fileGB = double(cb) * double(like.nx) * double(like.ny) / 1000000000.0;
    if (fileGB > 4.0) printf("Setting BIGTIFF, File: %s, Anticipated size (GB):%.2f\n", name.c_str(), fileGB);
    tdx::TiffWriter wr;
    if (!wr.create(name, like.nx, like.ny, type, nodata, &like, want_lzw())) { printf("Error opening file %s.\n", name.c_str()); g_tdx_thread_error = wr.error(); return TDX_ERR_FILE; }
    // with --gpus N the rows are encoded / written by N host threads (each rank's rows, src/tiffIO.cpp:382-427); same bytes for any N
    if (!wr.write_all(data, std::min(tool_gpus(), 64))) { g_tdx_thread_error = wr.error(); return TDX_ERR_FILE; }
    if (!wr.close()) { g_tdx_thread_error = wr.error(); return TDX_ERR_FILE; }
    return TDX_OK;
}

struct CtxGuard {
    tdx_context* c = nullptr;
    int rc;
    CtxGuard() { rc = tdx_context_create(tool_device(), &c); if (rc != TDX_OK) fprintf(stderr, "taudem_amd: %s\n", tdx_last_error(nullptr)); }
    ~CtxGuard() { tdx_context_destroy(c); }
};

// outlets: readoutlets + geoToGlobalXY (src/aread8.cpp:114-136,179-189)
int load_outlets(const char* datasrc, const tdx::RasterInfo& ri, std::vector<int32_t>& ox, std::vector<int32_t>& oy) {
    std::vector<double> x, y; std::vector<int> id; std::string err;
    if (!tdx::read_outlets(datasrc, x, y, id, err)) {
        printf("Error Opening OGR Data Source .\n");
        printf("Error opening shapefile. Exiting \n");
        fflush(stdout);
        g_tdx_thread_error = err;
        return TDX_ERR_OUTLETS;
    }
    printf("Warning: Spatial References of Outlet feature and Raster data are missing.\n");
    ox.resize(x.size()); oy.resize(x.size());
    for (size_t i = 0; i < x.size(); i++) {
        int gx, gy;
        tdx::geo_to_global_xy(x[i], y[i], ri.xleftedge, ri.ytopedge, ri.dlon, ri.dlat, gx, gy);
        ox[i] = gx; oy[i] = gy;
    }
    return TDX_OK;
}

void print_gpu_stats(const char* tool, const tdx_stats& st, int64_t cells) {
    if (!getenv("TAUDEM_AMD_STATS")) return;
    fprintf(stderr, "{\"tool\": \"%s\", \"cells\": %lld, \"device_ms\": %.3f, \"mcells_per_s\": %.3f, \"rounds\": %lld, \"flats\": %lld, "
                    "\"levels_fall\": %lld, \"levels_rise\": %lld}\n",
            tool, (long long)cells, st.ms_total, st.ms_total > 0 ? double(cells) / st.ms_total / 1000.0 : 0.0, (long long)st.rounds,
            (long long)st.flats_initial, (long long)st.levels_fall, (long long)st.levels_rise);
}

}  // namespace

extern "C" {

int tdx_tool_gridnet(const char* pfile, const char* plenfile, const char* tlenfile, const char* gordfile, const char* maskfile, const char* datasrc,
                     const char* /*lyrname*/, int /*uselyrname*/, int /*lyrno*/, int useMask, int useOutlets, int thresh) {
    printf("GridNet version %s\n", TDVERSION);
    fflush(stdout);
    const double begint = now_s();
    Raster p, mask;
    int rc = load_raster(pfile, tdx::DType::I16, p);
    if (rc != TDX_OK) return rc;
    std::vector<int32_t> ox, oy;
    if (useOutlets == 1) { rc = load_outlets(datasrc, p.info, ox, oy); if (rc != TDX_OK) return rc; }   // src/gridnet.cpp:78-120
    if (useMask == 1) {
        rc = load_raster(maskfile, tdx::DType::I32, mask);
        if (rc != TDX_OK) return rc;
        if (!compare_rasters(p.info, pfile, mask.info, maskfile)) { printf("File sizes do not match\n%s\n", maskfile); fflush(stdout); return TDX_ERR_OUTLETS; }   // src/gridnet.cpp:147-152
    }
    const double readt = now_s();
    const size_t n = p.s.size();
    std::vector<float> plen(n), tlen(n);
    std::vector<int16_t> gord(n);
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), p.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), p.info.nx, p.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            int16_t* d_p = j.strip<int16_t>(p.s.data());
            int32_t* d_m = useMask ? j.strip<int32_t>(mask.l.data()) : nullptr;
            float* d_pl = j.strip<float>(nullptr);
            float* d_tl = j.strip<float>(nullptr);
            int16_t* d_go = j.strip<int16_t>(nullptr);
            if (!d_p || !d_pl || !d_tl || !d_go || (useMask && !d_m)) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(p.info.dxc), dys = j.rows_of(p.info.dyc);
            const std::vector<int32_t> lrow = j.local_rows(oy);
            const int e = tdx_gridnet_strip(j.ctx, j.comm, d_p, j.nx, j.nyl, (int16_t)p.info.nodata, dxs.data(), dys.data(), d_m, thresh, useOutlets ? ox.data() : nullptr,
                                            useOutlets ? lrow.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, d_pl, d_tl, d_go, s);
            if (e != TDX_OK) return e;
            return (j.fetch(plen.data(), d_pl) && j.fetch(tlen.data(), d_tl) && j.fetch(gord.data(), d_go)) ? TDX_OK : TDX_ERR_HIP;
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_gridnet(g.c, p.s.data(), p.info.nx, p.info.ny, (int16_t)p.info.nodata, p.info.dxc.data(), p.info.dyc.data(), useMask ? mask.l.data() : nullptr, thresh,
                         useOutlets ? ox.data() : nullptr, useOutlets ? oy.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, plen.data(), tlen.data(),
                         gord.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(gordfile, tdx::DType::I16, gord.data(), p.info, -1.0);   // src/gridnet.cpp:474-481
    if (rc != TDX_OK) return rc;
    rc = save_raster(plenfile, tdx::DType::F32, plen.data(), p.info, -1.0);
    if (rc != TDX_OK) return rc;
    rc = save_raster(tlenfile, tdx::DType::F32, tlen.data(), p.info, -1.0);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt,
           writet - computet, writet - begint);
    print_gpu_stats("gridnet", st, p.info.nx * p.info.ny);
    return 0;
}

int tdx_tool_d8flowpathextremeup(const char* pfile, const char* safile, const char* ssafile, int usemax, const char* datasrc, const char* /*lyrname*/,
                                 int /*uselyrname*/, int /*lyrno*/, int useOutlets, int contcheck) {
    printf("D8FlowPathExtremeUp version %s\n", TDVERSION);
    fflush(stdout);
    const double begint = now_s();
    Raster p, sa;
    int rc = load_raster(pfile, tdx::DType::I16, p);
    if (rc != TDX_OK) return rc;
    std::vector<int32_t> ox, oy;
    if (useOutlets == 1) { rc = load_outlets(datasrc, p.info, ox, oy); if (rc != TDX_OK) return rc; }
    rc = load_raster(safile, tdx::DType::F32, sa);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(p.info, pfile, sa.info, safile)) { printf("File sizes do not match\n%s\n", safile); fflush(stdout); return TDX_ERR_OUTLETS; }   // src/D8flowpathextremeup.cpp:120-125
    const double readt = now_s();
    std::vector<float> ssa(p.s.size());
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), p.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), p.info.nx, p.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            int16_t* d_p = j.strip<int16_t>(p.s.data());
            float* d_sa = j.strip<float>(sa.f.data());
            float* d_out = j.strip<float>(nullptr);
            if (!d_p || !d_sa || !d_out) return TDX_ERR_NOMEM;
            const std::vector<int32_t> lrow = j.local_rows(oy);
            const int e = tdx_d8flowpathextremeup_strip(j.ctx, j.comm, d_p, j.nx, j.nyl, (int16_t)p.info.nodata, d_sa, usemax, contcheck, useOutlets ? ox.data() : nullptr,
                                                        useOutlets ? lrow.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, d_out, s);
            return e != TDX_OK ? e : (j.fetch(ssa.data(), d_out) ? TDX_OK : TDX_ERR_HIP);
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_d8flowpathextremeup(g.c, p.s.data(), p.info.nx, p.info.ny, (int16_t)p.info.nodata, sa.f.data(), usemax, contcheck,
                                     useOutlets ? ox.data() : nullptr, useOutlets ? oy.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, ssa.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(ssafile, tdx::DType::F32, ssa.data(), p.info, (double)TDX_ANG_NODATA);   // MISSINGFLOAT = -FLT_MAX (src/commonLib.h:80)
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt,
           writet - computet, writet - begint);
    print_gpu_stats("d8flowpathextremeup", st, p.info.nx * p.info.ny);
    return 0;
}

int tdx_tool_threshold(const char* ssafile, const char* srcfile, const char* maskfile, float thresh, int usemask) {
    printf("Threshold version %s\n", TDVERSION);
    fflush(stdout);
    const double begint = now_s();
    Raster ssa, mask;
    int rc = load_raster(ssafile, tdx::DType::F32, ssa);
    if (rc != TDX_OK) return rc;
    if (usemask == 1) {
        rc = load_raster(maskfile, tdx::DType::F32, mask);
        if (rc != TDX_OK) return rc;
        if (!compare_rasters(ssa.info, ssafile, mask.info, maskfile)) return TDX_ERR_MISMATCH;   // src/Threshold.cpp:90
    }
    const double readt = now_s();
    CtxGuard g;
    if (g.rc != TDX_OK) return g.rc;
    std::vector<int16_t> src(ssa.f.size());
    tdx_stats st;
    rc = tdx_threshold(g.c, ssa.f.data(), ssa.info.nx, ssa.info.ny, (float)ssa.info.nodata, usemask ? mask.f.data() : nullptr, thresh, src.data(), &st);
    if (rc != TDX_OK) { fprintf(stderr, "taudem_amd: %s\n", tdx_last_error(g.c)); return rc; }
    const double computet = now_s();
    rc = save_raster(srcfile, tdx::DType::I16, src.data(), ssa.info, -32768.0);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", 1, readt - begint, computet - readt,
           writet - computet, writet - begint);
    return 0;
}

int tdx_tool_dinfupdependence(const char* angfile, const char* dgfile, const char* depfile) {
    printf("DinfUpDependence version %s\n", TDVERSION);
    fflush(stdout);
    const double begint = now_s();
    Raster ang, dg;
    int rc = load_raster(angfile, tdx::DType::F32, ang);
    if (rc != TDX_OK) return rc;
    rc = load_raster(dgfile, tdx::DType::I32, dg);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, dg.info, dgfile)) return TDX_ERR_MISMATCH;   // src/DinfUpDependence.cpp:103
    const double readt = now_s();
    std::vector<float> dep(ang.f.size());
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), ang.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), ang.info.nx, ang.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_ang = j.strip<float>(ang.f.data());
            int32_t* d_dg = j.strip<int32_t>(dg.l.data());
            float* d_dep = j.strip<float>(nullptr);
            if (!d_ang || !d_dg || !d_dep) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(ang.info.dxc), dys = j.rows_of(ang.info.dyc);
            const int e = tdx_dinfupdependence_strip(j.ctx, j.comm, d_ang, j.nx, j.nyl, (float)ang.info.nodata, dxs.data(), dys.data(), d_dg, d_dep, s);
            return e != TDX_OK ? e : (j.fetch(dep.data(), d_dep) ? TDX_OK : TDX_ERR_HIP);
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_dinfupdependence(g.c, ang.f.data(), ang.info.nx, ang.info.ny, (float)ang.info.nodata, ang.info.dxc.data(), ang.info.dyc.data(), dg.l.data(), dep.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(depfile, tdx::DType::F32, dep.data(), ang.info, -1.0);   // depNodata = -1 (src/DinfUpDependence.cpp:113)
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt, writet - computet,
           writet - begint);
    print_gpu_stats("dinfupdependence", st, ang.info.nx * ang.info.ny);
    return 0;
}

int tdx_tool_dinfconclimaccum(const char* angfile, const char* ctptfile, const char* dmfile, const char* datasrc, const char* /*lyrname*/, int /*uselyrname*/,
                              int /*lyrno*/, const char* qfile, const char* dgfile, int useOutlets, int contcheck, float cSol) {
    printf("DinfConcLimAccum version %s\n", TDVERSION);
    const double begint = now_s();
    Raster ang, dm, dg, q;
    int rc = load_raster(angfile, tdx::DType::F32, ang);
    if (rc != TDX_OK) return rc;
    std::vector<int32_t> ox, oy;
    if (useOutlets == 1) { rc = load_outlets(datasrc, ang.info, ox, oy); if (rc != TDX_OK) return rc; }
    rc = load_raster(dmfile, tdx::DType::F32, dm);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, dm.info, dmfile)) { printf("File sizes do not match\n%s\n", dmfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    rc = load_raster(dgfile, tdx::DType::I16, dg);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, dg.info, dgfile)) { printf("File sizes do not match\n%s\n", dgfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    rc = load_raster(qfile, tdx::DType::F32, q);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, q.info, qfile)) { printf("File sizes do not match\n%s\n", qfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    const double readt = now_s();
    std::vector<float> out(ang.f.size());
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), ang.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), ang.info.nx, ang.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_ang = j.strip<float>(ang.f.data());
            float* d_dm = j.strip<float>(dm.f.data());
            float* d_q = j.strip<float>(q.f.data());
            int16_t* d_dg = j.strip<int16_t>(dg.s.data());
            float* d_out = j.strip<float>(nullptr);
            if (!d_ang || !d_dm || !d_q || !d_dg || !d_out) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(ang.info.dxc), dys = j.rows_of(ang.info.dyc);
            const std::vector<int32_t> lrow = j.local_rows(oy);
            const int e = tdx_dinfconclimaccum_strip(j.ctx, j.comm, d_ang, j.nx, j.nyl, (float)ang.info.nodata, dxs.data(), dys.data(), d_dm, (float)dm.info.nodata, d_dg,
                                                     d_q, (float)q.info.nodata, cSol, contcheck, useOutlets ? ox.data() : nullptr, useOutlets ? lrow.data() : nullptr,
                                                     useOutlets ? int64_t(ox.size()) : -1, d_out, s);
            return e != TDX_OK ? e : (j.fetch(out.data(), d_out) ? TDX_OK : TDX_ERR_HIP);
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_dinfconclimaccum(g.c, ang.f.data(), ang.info.nx, ang.info.ny, (float)ang.info.nodata, ang.info.dxc.data(), ang.info.dyc.data(), dm.f.data(),
                                  (float)dm.info.nodata, dg.s.data(), q.f.data(), (float)q.info.nodata, cSol, contcheck, useOutlets ? ox.data() : nullptr,
                                  useOutlets ? oy.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, out.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(ctptfile, tdx::DType::F32, out.data(), ang.info, (double)TDX_ANG_NODATA);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt, writet - computet,
           writet - begint);
    print_gpu_stats("dinfconclimaccum", st, ang.info.nx * ang.info.ny);
    return 0;
}

int tdx_tool_dinftranslimaccum(const char* angfile, const char* tsupfile, const char* tcfile, const char* tlafile, const char* depfile, const char* cinfile,
                               const char* coutfile, const char* datasrc, const char* /*lyrname*/, int /*uselyrname*/, int /*lyrno*/, int useOutlets, int usec,
                               int contcheck) {
    printf("DinfTransLimAccum version %s\n", TDVERSION);
    const double begint = now_s();
    Raster ang, tsup, tc, cin;
    int rc = load_raster(angfile, tdx::DType::F32, ang);
    if (rc != TDX_OK) return rc;
    std::vector<int32_t> ox, oy;
    if (useOutlets == 1) { rc = load_outlets(datasrc, ang.info, ox, oy); if (rc != TDX_OK) return rc; }
    rc = load_raster(tsupfile, tdx::DType::F32, tsup);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, tsup.info, tsupfile)) { printf("File sizes do not match\n%s\n", tsupfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    rc = load_raster(tcfile, tdx::DType::F32, tc);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, tc.info, tcfile)) { printf("File sizes do not match\n%s\n", tcfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    if (usec == 1) {
        rc = load_raster(cinfile, tdx::DType::F32, cin);
        if (rc != TDX_OK) return rc;
        if (!compare_rasters(ang.info, angfile, cin.info, cinfile)) { printf("File sizes do not match\n%s\n", cinfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    }
    const double readt = now_s();
    std::vector<float> tla(ang.f.size()), dep(ang.f.size()), cso(usec ? ang.f.size() : 0);
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), ang.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), ang.info.nx, ang.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_ang = j.strip<float>(ang.f.data());
            float* d_ts = j.strip<float>(tsup.f.data());
            float* d_tc = j.strip<float>(tc.f.data());
            float* d_ci = usec ? j.strip<float>(cin.f.data()) : nullptr;
            float* d_tla = j.strip<float>(nullptr);
            float* d_dep = j.strip<float>(nullptr);
            float* d_co = usec ? j.strip<float>(nullptr) : nullptr;
            if (!d_ang || !d_ts || !d_tc || !d_tla || !d_dep || (usec && (!d_ci || !d_co))) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(ang.info.dxc), dys = j.rows_of(ang.info.dyc);
            const std::vector<int32_t> lrow = j.local_rows(oy);
            const int e = tdx_dinftranslimaccum_strip(j.ctx, j.comm, d_ang, j.nx, j.nyl, (float)ang.info.nodata, dxs.data(), dys.data(), d_ts, (float)tsup.info.nodata, d_tc,
                                                      (float)tc.info.nodata, d_ci, usec ? (float)cin.info.nodata : 0.f, contcheck, useOutlets ? ox.data() : nullptr,
                                                      useOutlets ? lrow.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, d_tla, d_dep, d_co, s);
            if (e != TDX_OK) return e;
            if (!j.fetch(tla.data(), d_tla) || !j.fetch(dep.data(), d_dep) || (usec && !j.fetch(cso.data(), d_co))) return TDX_ERR_HIP;
            return TDX_OK;
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_dinftranslimaccum(g.c, ang.f.data(), ang.info.nx, ang.info.ny, (float)ang.info.nodata, ang.info.dxc.data(), ang.info.dyc.data(), tsup.f.data(),
                                   (float)tsup.info.nodata, tc.f.data(), (float)tc.info.nodata, usec ? cin.f.data() : nullptr, usec ? (float)cin.info.nodata : 0.f,
                                   contcheck, useOutlets ? ox.data() : nullptr, useOutlets ? oy.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, tla.data(),
                                   dep.data(), usec ? cso.data() : nullptr, &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(tlafile, tdx::DType::F32, tla.data(), ang.info, (double)TDX_ANG_NODATA);
    if (rc != TDX_OK) return rc;
    rc = save_raster(depfile, tdx::DType::F32, dep.data(), ang.info, (double)TDX_ANG_NODATA);
    if (rc != TDX_OK) return rc;
    if (usec == 1) {
        rc = save_raster(coutfile, tdx::DType::F32, cso.data(), ang.info, (double)TDX_ANG_NODATA);
        if (rc != TDX_OK) return rc;
    }
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt, writet - computet,
           writet - begint);
    print_gpu_stats("dinftranslimaccum", st, ang.info.nx * ang.info.ny);
    return 0;
}

int tdx_tool_dinfrevaccum(const char* angfile, const char* wgfile, const char* raccfile, const char* dmaxfile) {
    printf("DinfRevAccum version %s\n", TDVERSION);
    fflush(stdout);
    const double begint = now_s();
    Raster ang, w;
    int rc = load_raster(angfile, tdx::DType::F32, ang);
    if (rc != TDX_OK) return rc;
    rc = load_raster(wgfile, tdx::DType::F32, w);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, w.info, wgfile)) { printf("File sizes do not match\n%s\n", wgfile); fflush(stdout); return TDX_ERR_OUTLETS; }   // src/DinfRevAccum.cpp:94-99
    const double readt = now_s();
    const size_t n = ang.f.size();
    std::vector<float> racc(n), dmax(n);
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), ang.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), ang.info.nx, ang.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_ang = j.strip<float>(ang.f.data());
            float* d_w = j.strip<float>(w.f.data());
            float* d_r = j.strip<float>(nullptr);
            float* d_m = j.strip<float>(nullptr);
            if (!d_ang || !d_w || !d_r || !d_m) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(ang.info.dxc), dys = j.rows_of(ang.info.dyc);
            const int e = tdx_dinfrevaccum_strip(j.ctx, j.comm, d_ang, j.nx, j.nyl, (float)ang.info.nodata, dxs.data(), dys.data(), d_w, (float)w.info.nodata, d_r, d_m, s);
            if (e != TDX_OK) return e;
            return (j.fetch(racc.data(), d_r) && j.fetch(dmax.data(), d_m)) ? TDX_OK : TDX_ERR_HIP;
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_dinfrevaccum(g.c, ang.f.data(), ang.info.nx, ang.info.ny, (float)ang.info.nodata, ang.info.dxc.data(), ang.info.dyc.data(), w.f.data(),
                              (float)w.info.nodata, racc.data(), dmax.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(raccfile, tdx::DType::F32, racc.data(), ang.info, (double)TDX_ANG_NODATA);   // MISSINGFLOAT (src/DinfRevAccum.cpp:253-258)
    if (rc != TDX_OK) return rc;
    rc = save_raster(dmaxfile, tdx::DType::F32, dmax.data(), ang.info, (double)TDX_ANG_NODATA);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt, writet - computet,
           writet - begint);
    print_gpu_stats("dinfrevaccum", st, ang.info.nx * ang.info.ny);
    return 0;
}

int tdx_tool_set_device(int device) { g_tool_device = device; return TDX_OK; }
int tdx_tool_set_gpus(int ngpus) { g_tool_gpus = ngpus; return TDX_OK; }

int tdx_tool_pitremove(const char* demfile, const char* felfile, const char* /*sfdrfile*/, int /*usesfdr*/,
                       int verbose, int is_4Point, int use_mask, const char* maskfile) {
    printf("PitRemove version %s\n", TDVERSION);
    fflush(stdout);
    const double begint = now_s();
    Raster dem, mask;
    int rc = load_raster(demfile, tdx::DType::F32, dem);
    if (rc != TDX_OK) return rc;
    if (use_mask) {
        rc = load_raster(maskfile, tdx::DType::I16, mask);
        if (rc != TDX_OK) return rc;
        if (!compare_rasters(dem.info, demfile, mask.info, maskfile)) {
            printf("Error: depression mask and input DEM are not similar. Files must have the same number of rows/columns.\n");
            fflush(stdout);
            return TDX_ERR_MISMATCH;
        }
    }
    const double readt = now_s();
    if (verbose) { printf("Header read\nData read\n"); if (use_mask) printf("Process: 0, Using depression mask data...\n"); fflush(stdout); }
    std::vector<float> fel(dem.f.size());
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), dem.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), dem.info.nx, dem.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_dem = j.strip<float>(dem.f.data());
            int16_t* d_mask = use_mask ? j.strip<int16_t>(mask.s.data()) : nullptr;
            float* d_fel = j.strip<float>(nullptr);
            if (!d_dem || !d_fel || (use_mask && !d_mask)) return TDX_ERR_NOMEM;
            const int e = tdx_pitremove_strip(j.ctx, j.comm, d_dem, j.nx, j.nyl, (float)dem.info.nodata, d_mask, is_4Point, d_fel, s);
            return e != TDX_OK ? e : (j.fetch(fel.data(), d_fel) ? TDX_OK : TDX_ERR_HIP);
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_pitremove(g.c, dem.f.data(), dem.info.nx, dem.info.ny, (float)dem.info.nodata, use_mask ? mask.s.data() : nullptr,
                           is_4Point, fel.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    if (verbose) printf("Process: 0, Pass: %lld, Remaining: 0\n", (long long)st.rounds);
    const double computet = now_s();
    const float felNodata = -3.0e38f;
    rc = save_raster(felfile, tdx::DType::F32, fel.data(), dem.info, (double)felNodata);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processes: %d\nHeader read time: %f\nData read time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n",
           nproc, 0.0, readt - begint, computet - readt, writet - computet, writet - begint);
    print_gpu_stats("pitremove", st, dem.info.nx * dem.info.ny);
    return 0;
}

int tdx_tool_d8flowdir(const char* demfile, const char* pointfile, const char* slopefile, const char* /*flowfile*/, int useflowfile) {
    printf("D8FlowDir version %s\n", TDVERSION);
    fflush(stdout);
    if (useflowfile == 1) {
        // the reference's -sfdr branch reads an int32 partition through the int16 accessor and aborts
        // (src/d8.cpp:119,239-241 -> src/partition.h:100-107): treated as unsupported, same exit code
        printf("Attempt to access short grid with incorrect data type\n");
        return 41;
    }
    const double begint = now_s();
    Raster dem;
    int rc = load_raster(demfile, tdx::DType::F32, dem);
    if (rc != TDX_OK) return rc;
    const double readt = now_s();
    const size_t n = dem.f.size();
    std::vector<int16_t> p(n);
    std::vector<float> sd8(n);
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), dem.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), dem.info.nx, dem.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_fel = j.strip<float>(dem.f.data());
            int16_t* d_p = j.strip<int16_t>(nullptr);
            float* d_sd8 = j.strip<float>(nullptr);
            if (!d_fel || !d_p || !d_sd8) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(dem.info.dxc), dys = j.rows_of(dem.info.dyc);
            const int e = tdx_d8flowdir_strip(j.ctx, j.comm, d_fel, j.nx, j.nyl, (float)dem.info.nodata, dxs.data(), dys.data(), d_p, d_sd8, s);
            if (e != TDX_OK) return e;
            return (j.fetch(p.data(), d_p) && j.fetch(sd8.data(), d_sd8)) ? TDX_OK : TDX_ERR_HIP;
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_d8flowdir(g.c, dem.f.data(), dem.info.nx, dem.info.ny, (float)dem.info.nodata, dem.info.dxc.data(), dem.info.dyc.data(),
                           p.data(), sd8.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    fprintf(stderr, "All slopes evaluated. %ld flats to resolve.\n", (long)st.flats_initial);
    if (st.flat_iterations > 0 && st.flats_left > 0) fprintf(stderr, "Iteration complete. Number of flats remaining: %ld\n", (long)st.flats_left);
    rc = save_raster(slopefile, tdx::DType::F32, sd8.data(), dem.info, -1.0);
    if (rc != TDX_OK) return rc;
    const double writeSlopet = now_s();
    rc = save_raster(pointfile, tdx::DType::I16, p.data(), dem.info, -32768.0);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    const double slope_s = st.ms_kernel[TDX_K_STENCIL] / 1000.0;
    printf("Processors: %d\nHeader read time: %f\nData read time: %f\nCompute Slope time: %f\nWrite Slope time: %f\nResolve Flat time: %f\nWrite Flat time: %f\nTotal time: %f\n",
           nproc, 0.0, readt - begint, slope_s, writeSlopet - computet, (computet - readt) - slope_s, writet - writeSlopet, writet - begint);
    print_gpu_stats("d8flowdir", st, dem.info.nx * dem.info.ny);
    return 0;
}

int tdx_tool_aread8(const char* pfile, const char* afile, const char* datasrc, const char* /*lyrname*/, int /*uselyrname*/, int /*lyrno*/,
                    const char* wfile, int useOutlets, int usew, int contcheck) {
    {   // existence probe of the reference (src/aread8.cpp:62-86)
        FILE* fp = fopen(pfile, "r");
        if (!fp) { fprintf(stderr, "Error: Input file %s does not exist.\n", pfile); return TDX_ERR_FILE; }
        fclose(fp);
    }
    printf("AreaD8 version %s\n", TDVERSION);
    const double begint = now_s();
    Raster p, w;
    int rc = load_raster(pfile, tdx::DType::I16, p);
    if (rc != TDX_OK) return rc;
    std::vector<int32_t> ox, oy;
    if (useOutlets == 1) { rc = load_outlets(datasrc, p.info, ox, oy); if (rc != TDX_OK) return rc; }
    if (usew == 1) {
        rc = load_raster(wfile, tdx::DType::F32, w);
        if (rc != TDX_OK) return rc;
        if (!compare_rasters(p.info, pfile, w.info, wfile)) { printf("File sizes do not match\n%s\n", wfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    }
    const double readt = now_s();
    std::vector<float> a(p.s.size());
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), p.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), p.info.nx, p.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            int16_t* d_p = j.strip<int16_t>(p.s.data());
            float* d_w = usew ? j.strip<float>(w.f.data()) : nullptr;
            float* d_a = j.strip<float>(nullptr);
            if (!d_p || !d_a || (usew && !d_w)) return TDX_ERR_NOMEM;
            const std::vector<int32_t> lrow = j.local_rows(oy);
            const int e = tdx_aread8_strip(j.ctx, j.comm, d_p, j.nx, j.nyl, (int16_t)p.info.nodata, d_w, usew ? (float)w.info.nodata : 0.f, contcheck,
                                           useOutlets ? ox.data() : nullptr, useOutlets ? lrow.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, d_a, s);
            return e != TDX_OK ? e : (j.fetch(a.data(), d_a) ? TDX_OK : TDX_ERR_HIP);
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_aread8(g.c, p.s.data(), p.info.nx, p.info.ny, (int16_t)p.info.nodata, usew ? w.f.data() : nullptr, usew ? (float)w.info.nodata : 0.f,
                        contcheck, useOutlets ? ox.data() : nullptr, useOutlets ? oy.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, a.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(afile, tdx::DType::F32, a.data(), p.info, -1.0);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Number of Processes: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt,
           writet - computet, writet - begint);
    print_gpu_stats("aread8", st, p.info.nx * p.info.ny);
    return 0;
}

int tdx_tool_dinfflowdir(const char* demfile, const char* angfile, const char* slopefile, const char* /*flowfile*/, int /*useflowfile*/) {
    printf("DinfFlowDir version %s\n", TDVERSION);
    fflush(stdout);
    const double begint = now_s();
    Raster dem;
    int rc = load_raster(demfile, tdx::DType::F32, dem);
    if (rc != TDX_OK) return rc;
    const double readt = now_s();
    const size_t n = dem.f.size();
    std::vector<float> ang(n), slp(n);
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), dem.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), dem.info.nx, dem.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_fel = j.strip<float>(dem.f.data());
            float* d_ang = j.strip<float>(nullptr);
            float* d_slp = j.strip<float>(nullptr);
            if (!d_fel || !d_ang || !d_slp) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(dem.info.dxc), dys = j.rows_of(dem.info.dyc);
            const int e = tdx_dinfflowdir_strip(j.ctx, j.comm, d_fel, j.nx, j.nyl, (float)dem.info.nodata, dxs.data(), dys.data(), d_ang, d_slp, s);
            if (e != TDX_OK) return e;
            return (j.fetch(ang.data(), d_ang) && j.fetch(slp.data(), d_slp)) ? TDX_OK : TDX_ERR_HIP;
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_dinfflowdir(g.c, dem.f.data(), dem.info.nx, dem.info.ny, (float)dem.info.nodata, dem.info.dxc.data(), dem.info.dyc.data(),
                             ang.data(), slp.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    fprintf(stderr, "All slopes evaluated. %ld flats to resolve.\n", (long)st.flats_initial);
    rc = save_raster(slopefile, tdx::DType::F32, slp.data(), dem.info, -1.0);
    if (rc != TDX_OK) return rc;
    const double writeSlopet = now_s();
    rc = save_raster(angfile, tdx::DType::F32, ang.data(), dem.info, (double)TDX_ANG_NODATA);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    const double slope_s = st.ms_kernel[TDX_K_STENCIL] / 1000.0;
    printf("Processors: %d\nHeader read time: %f\nData read time: %f\nCompute Slope time: %f\nWrite Slope time: %f\nResolve Flat time: %f\nWrite Flat time: %f\nTotal time: %f\n",
           nproc, 0.0, readt - begint, slope_s, writeSlopet - computet, (computet - readt) - slope_s, writet - writeSlopet, writet - begint);
    print_gpu_stats("dinfflowdir", st, dem.info.nx * dem.info.ny);
    return 0;
}

int tdx_tool_areadinf(const char* angfile, const char* scafile, const char* datasrc, const char* /*lyrname*/, int /*uselyrname*/, int /*lyrno*/,
                      const char* wfile, int useOutlets, int usew, int contcheck) {
    printf("AreaDinf version %s\n", TDVERSION);
    const double begint = now_s();
    Raster ang, w;
    int rc = load_raster(angfile, tdx::DType::F32, ang);
    if (rc != TDX_OK) return rc;
    std::vector<int32_t> ox, oy;
    if (useOutlets == 1) { rc = load_outlets(datasrc, ang.info, ox, oy); if (rc != TDX_OK) return rc; }
    if (usew == 1) {
        rc = load_raster(wfile, tdx::DType::F32, w);
        if (rc != TDX_OK) return rc;
        if (!compare_rasters(ang.info, angfile, w.info, wfile)) return TDX_ERR_MISMATCH;   // src/areadinf.cpp:134
    }
    const double readt = now_s();
    std::vector<float> sca(ang.f.size());
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), ang.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), ang.info.nx, ang.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_ang = j.strip<float>(ang.f.data());
            float* d_w = usew ? j.strip<float>(w.f.data()) : nullptr;
            float* d_out = j.strip<float>(nullptr);
            if (!d_ang || !d_out || (usew && !d_w)) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(ang.info.dxc), dys = j.rows_of(ang.info.dyc);
            const std::vector<int32_t> lrow = j.local_rows(oy);
            const int e = tdx_areadinf_strip(j.ctx, j.comm, d_ang, j.nx, j.nyl, (float)ang.info.nodata, dxs.data(), dys.data(), d_w, contcheck,
                                             useOutlets ? ox.data() : nullptr, useOutlets ? lrow.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, d_out, s);
            return e != TDX_OK ? e : (j.fetch(sca.data(), d_out) ? TDX_OK : TDX_ERR_HIP);
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_areadinf(g.c, ang.f.data(), ang.info.nx, ang.info.ny, (float)ang.info.nodata, ang.info.dxc.data(), ang.info.dyc.data(),
                          usew ? w.f.data() : nullptr, contcheck, useOutlets ? ox.data() : nullptr, useOutlets ? oy.data() : nullptr,
                          useOutlets ? int64_t(ox.size()) : -1, sca.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(scafile, tdx::DType::F32, sca.data(), ang.info, -1.0);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt,
           writet - computet, writet - begint);
    print_gpu_stats("areadinf", st, ang.info.nx * ang.info.ny);
    return 0;
}

int tdx_tool_dinfdecayaccum(const char* angfile, const char* adecfile, const char* dmfile, const char* datasrc, const char* /*lyrname*/,
                            int /*uselyrname*/, int /*lyrno*/, const char* wfile, int useOutlets, int usew, int contcheck) {
    printf("DinfDecayAccum version %s\n", TDVERSION);
    const double begint = now_s();
    Raster ang, dm, w;
    int rc = load_raster(angfile, tdx::DType::F32, ang);
    if (rc != TDX_OK) return rc;
    std::vector<int32_t> ox, oy;
    if (useOutlets == 1) { rc = load_outlets(datasrc, ang.info, ox, oy); if (rc != TDX_OK) return rc; }
    rc = load_raster(dmfile, tdx::DType::F32, dm);
    if (rc != TDX_OK) return rc;
    if (!compare_rasters(ang.info, angfile, dm.info, dmfile)) { printf("File sizes do not match\n%s\n", dmfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    if (usew == 1) {
        rc = load_raster(wfile, tdx::DType::F32, w);
        if (rc != TDX_OK) return rc;
        if (!compare_rasters(ang.info, angfile, w.info, wfile)) { printf("File sizes do not match\n%s\n", wfile); fflush(stdout); return TDX_ERR_OUTLETS; }
    }
    const double readt = now_s();
    std::vector<float> out(ang.f.size());
    tdx_stats st;
    const int nproc = int(std::min<int64_t>(tool_gpus(), ang.info.ny));   // (at least one row per rank: the count that is printed is the count that ran)
    if (nproc > 1) {
        rc = toolstrips::run(nproc, tool_device(), ang.info.nx, ang.info.ny, &st, [&](toolstrips::RankJob& j, tdx_stats* s) {
            float* d_ang = j.strip<float>(ang.f.data());
            float* d_dm = j.strip<float>(dm.f.data());
            float* d_w = usew ? j.strip<float>(w.f.data()) : nullptr;
            float* d_out = j.strip<float>(nullptr);
            if (!d_ang || !d_dm || !d_out || (usew && !d_w)) return TDX_ERR_NOMEM;
            const std::vector<double> dxs = j.rows_of(ang.info.dxc), dys = j.rows_of(ang.info.dyc);
            const std::vector<int32_t> lrow = j.local_rows(oy);
            const int e = tdx_dinfdecayaccum_strip(j.ctx, j.comm, d_ang, j.nx, j.nyl, (float)ang.info.nodata, dxs.data(), dys.data(), d_dm, (float)dm.info.nodata, d_w,
                                                   contcheck, useOutlets ? ox.data() : nullptr, useOutlets ? lrow.data() : nullptr,
                                                   useOutlets ? int64_t(ox.size()) : -1, d_out, s);
            return e != TDX_OK ? e : (j.fetch(out.data(), d_out) ? TDX_OK : TDX_ERR_HIP);
        });
        if (rc != TDX_OK) return rc;
    } else {
        CtxGuard g;
        if (g.rc != TDX_OK) return g.rc;
        rc = tdx_dinfdecayaccum(g.c, ang.f.data(), ang.info.nx, ang.info.ny, (float)ang.info.nodata, ang.info.dxc.data(), ang.info.dyc.data(),
                                dm.f.data(), (float)dm.info.nodata, usew ? w.f.data() : nullptr, contcheck, useOutlets ? ox.data() : nullptr,
                                useOutlets ? oy.data() : nullptr, useOutlets ? int64_t(ox.size()) : -1, out.data(), &st);
        if (rc != TDX_OK) { report(g.c); return rc; }
    }
    const double computet = now_s();
    rc = save_raster(adecfile, tdx::DType::F32, out.data(), ang.info, (double)TDX_ANG_NODATA);
    if (rc != TDX_OK) return rc;
    const double writet = now_s();
    printf("Processors: %d\nRead time: %f\nCompute time: %f\nWrite time: %f\nTotal time: %f\n", nproc, readt - begint, computet - readt,
           writet - computet, writet - begint);
    print_gpu_stats("dinfdecayaccum", st, ang.info.nx * ang.info.ny);
    return 0;
}

}  // extern "C"
