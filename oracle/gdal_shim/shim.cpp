// Test infrastructure (oracle/): implementation of the GDAL/OGR entry points declared in gdal.h
// over this repository's GeoTIFF reader/writer and outlet reader, so that the unmodified
// reference sources under /root/reference/src build into oracle/_ref/ (see oracle/Makefile).
// Behavioural requirements taken from the reference's call sites:
//   * GDALRasterIO honours window offsets and converts file dtype <-> buffer dtype
//     (src/tiffIO.cpp:255,375,413)
//   * GDALOpen(GA_Update) lets ranks 1..P-1 write their strip into the file rank 0 created
//     (src/tiffIO.cpp:402-418)
//   * GDALGetRasterNoDataValue sets *pbSuccess (src/tiffIO.cpp:162)
// Output files are written as uncompressed strip GeoTIFF (pixel parity, not file-byte parity).
#include "gdal.h"

#include <cstdio>
#include <vector>

#include "../../taudem_amd/csrc/geotiff.hpp"
#include "../../taudem_amd/csrc/outlets.hpp"

namespace {

struct ShimDS {
    std::string path;
    bool writable = false, created = false, materialised = false;
    tdx::TiffReader rd;
    tdx::TiffWriter wr;
    tdx::RasterInfo info;     // for created datasets: assembled from the setters
    tdx::DType type = tdx::DType::F32;
    std::string wkt;
    bool geotransform_set = false;
};

std::string wkt_for(const tdx::RasterInfo& ri) {
    if (ri.geographic) return "GEOGCS[\"shim_geographic\"]";
    if (!ri.geo.geokeys.empty()) return "PROJCS[\"shim_projected\"]";
    return "";
}

tdx::DType to_dtype(GDALDataType t) {
    switch (t) {
        case GDT_Int16: return tdx::DType::I16;
        case GDT_Int32: return tdx::DType::I32;
        default: return tdx::DType::F32;
    }
}

bool materialise(ShimDS* d) {
    if (d->materialised) return true;
    // georeferencing tags from the geotransform (+ model type key so geographic survives a round trip)
    tdx::RasterInfo& ri = d->info;
    ri.geo = tdx::GeoTags();
    if (d->geotransform_set) {
        ri.geo.pixel_scale = {ri.gt[1], -ri.gt[5], 0.0};
        ri.geo.tiepoints = {0, 0, 0, ri.gt[0], ri.gt[3], 0};
    }
    const bool geog = d->wkt.rfind("GEOGCS", 0) == 0;
    const bool proj = d->wkt.rfind("PROJCS", 0) == 0;
    if (geog || proj) {
        ri.geo.geokeys = {1, 1, 0, 2, 1024, 0, 1, uint16_t(geog ? 2 : 1), 1025, 0, 1, 1};
    }
    if (!d->wr.create(d->path, ri.nx, ri.ny, d->type, ri.nodata, &ri, false)) {
        fprintf(stderr, "gdal shim: %s\n", d->wr.error().c_str());
        return false;
    }
    d->materialised = true;
    return true;
}

struct ShimSRS { std::string wkt; };

struct ShimLayer {
    std::vector<double> x, y;
    std::vector<int> id;
    size_t cursor = 0;
    std::string name;
};
struct ShimFeature { ShimLayer* layer; size_t idx; };
int g_field_dummy = 0;

}  // namespace

extern "C" {

void GDALAllRegister(void) {}

GDALDatasetH GDALOpen(const char* filename, GDALAccess access) {
    ShimDS* d = new ShimDS;
    d->path = filename;
    if (access == GA_ReadOnly) {
        if (!d->rd.open(d->path)) { delete d; return nullptr; }
        d->info = d->rd.info();
        d->wkt = wkt_for(d->info);
        if (d->info.file_format == 3) d->type = tdx::DType::F32;
        else if (d->info.file_bits == 16) d->type = tdx::DType::I16;
        else d->type = tdx::DType::I32;
        return d;
    }
    if (!d->wr.open_update(d->path)) { fprintf(stderr, "gdal shim: %s\n", d->wr.error().c_str()); delete d; return nullptr; }
    d->writable = true; d->materialised = true;
    d->type = d->wr.type();
    d->info.nx = d->wr.nx(); d->info.ny = d->wr.ny();
    return d;
}

void GDALClose(GDALDatasetH h) {
    ShimDS* d = static_cast<ShimDS*>(h);
    if (!d) return;
    if (d->created) materialise(d);
    d->wr.close();
    d->rd.close();
    // datasets opened read-only are leaked on purpose: tiffIO keeps raw pointers (copyfh) to them
    if (d->writable) delete d;
}

GDALDriverH GDALGetDatasetDriver(GDALDatasetH) { static int drv; return &drv; }
GDALDriverH GDALGetDriverByName(const char* name) { static int drv; return (name && strcmp(name, "GTiff") == 0) ? &drv : nullptr; }

const char* GDALGetProjectionRef(GDALDatasetH h) { return static_cast<ShimDS*>(h)->wkt.c_str(); }
CPLErr GDALSetProjection(GDALDatasetH h, const char* wkt) { static_cast<ShimDS*>(h)->wkt = wkt ? wkt : ""; return CE_None; }
GDALRasterBandH GDALGetRasterBand(GDALDatasetH h, int) { return h; }
const char* GDALGetRasterUnitType(GDALRasterBandH) { return ""; }
int GDALGetRasterXSize(GDALDatasetH h) { return int(static_cast<ShimDS*>(h)->info.nx); }
int GDALGetRasterYSize(GDALDatasetH h) { return int(static_cast<ShimDS*>(h)->info.ny); }
CPLErr GDALGetGeoTransform(GDALDatasetH h, double* gt) { memcpy(gt, static_cast<ShimDS*>(h)->info.gt, 6 * sizeof(double)); return CE_None; }
CPLErr GDALSetGeoTransform(GDALDatasetH h, double* gt) {
    ShimDS* d = static_cast<ShimDS*>(h);
    memcpy(d->info.gt, gt, 6 * sizeof(double));
    d->geotransform_set = true;
    return CE_None;
}
double GDALGetRasterNoDataValue(GDALRasterBandH h, int* ok) {
    ShimDS* d = static_cast<ShimDS*>(h);
    if (ok) *ok = d->info.has_nodata ? TRUE : FALSE;
    return d->info.has_nodata ? d->info.nodata : -1e10;
}
CPLErr GDALSetRasterNoDataValue(GDALRasterBandH h, double v) {
    ShimDS* d = static_cast<ShimDS*>(h);
    d->info.has_nodata = true; d->info.nodata = v;
    return CE_None;
}
GDALDataType GDALGetRasterDataType(GDALRasterBandH h) {
    switch (static_cast<ShimDS*>(h)->type) {
        case tdx::DType::I16: return GDT_Int16;
        case tdx::DType::I32: return GDT_Int32;
        default: return GDT_Float32;
    }
}

CPLErr GDALRasterIO(GDALRasterBandH h, GDALRWFlag rw, int xoff, int yoff, int xsize, int ysize, void* data,
                    int bxsize, int bysize, GDALDataType btype, int, int) {
    ShimDS* d = static_cast<ShimDS*>(h);
    if (bxsize != xsize || bysize != ysize) { fprintf(stderr, "gdal shim: resampling RasterIO not supported\n"); return CE_Failure; }
    if (rw == GF_Read) {
        if (!d->rd.read_window(xoff, yoff, xsize, ysize, to_dtype(btype), data)) {
            fprintf(stderr, "gdal shim: %s\n", d->rd.error().c_str());
            return CE_Failure;
        }
        return CE_None;
    }
    if (d->created && !materialise(d)) return CE_Failure;
    if (to_dtype(btype) != d->type || xoff != 0 || xsize != d->info.nx) {
        fprintf(stderr, "gdal shim: write must be full-width rows of the file type\n");
        return CE_Failure;
    }
    if (!d->wr.write_rows(yoff, ysize, data)) { fprintf(stderr, "gdal shim: %s\n", d->wr.error().c_str()); return CE_Failure; }
    return CE_None;
}

void GDALFlushCache(GDALDatasetH) {}

GDALDatasetH GDALCreate(GDALDriverH, const char* filename, int nx, int ny, int, GDALDataType type, char**) {
    ShimDS* d = new ShimDS;
    d->path = filename;
    d->writable = true; d->created = true;
    d->type = to_dtype(type);
    d->info.nx = nx; d->info.ny = ny;
    return d;
}

char** CSLSetNameValue(char** list, const char*, const char*) { static char* dummy[1] = {nullptr}; return list ? list : dummy; }
const char* CPLGetLastErrorMsg(void) { return ""; }

OGRSpatialReferenceH OSRNewSpatialReference(const char* wkt) { ShimSRS* s = new ShimSRS; s->wkt = wkt ? wkt : ""; return s; }
int OSRIsGeographic(OGRSpatialReferenceH s) { return s && static_cast<ShimSRS*>(s)->wkt.rfind("GEOGCS", 0) == 0; }
int OSRIsProjected(OGRSpatialReferenceH s) { return s && static_cast<ShimSRS*>(s)->wkt.rfind("PROJCS", 0) == 0; }
double OSRGetLinearUnits(OGRSpatialReferenceH, char** name) { static char m[] = "metre"; if (name) *name = m; return 1.0; }
const char* OSRGetAttrValue(OGRSpatialReferenceH, const char*, int) { return "shim"; }

void OGRRegisterAll(void) {}
OGRDataSourceH OGROpen(const char* name, int, OGRSFDriverH*) {
    ShimLayer* l = new ShimLayer;
    std::string err;
    if (!tdx::read_outlets(name, l->x, l->y, l->id, err)) { delete l; return nullptr; }
    l->name = name;
    return l;
}
void OGR_DS_Destroy(OGRDataSourceH ds) { delete static_cast<ShimLayer*>(ds); }
int OGR_DS_GetLayerCount(OGRDataSourceH) { return 1; }
OGRLayerH OGR_DS_GetLayer(OGRDataSourceH ds, int i) { return i == 0 ? ds : nullptr; }
OGRLayerH OGR_DS_GetLayerByName(OGRDataSourceH ds, const char*) { return ds; }
const char* OGR_L_GetName(OGRLayerH l) { return static_cast<ShimLayer*>(l)->name.c_str(); }
OGRwkbGeometryType OGR_L_GetGeomType(OGRLayerH) { return wkbPoint; }
OGRSpatialReferenceH OGR_L_GetSpatialRef(OGRLayerH) { return nullptr; }
GIntBig OGR_L_GetFeatureCount(OGRLayerH l, int) { return GIntBig(static_cast<ShimLayer*>(l)->x.size()); }
OGRFeatureDefnH OGR_L_GetLayerDefn(OGRLayerH l) { return l; }
void OGR_L_ResetReading(OGRLayerH l) { static_cast<ShimLayer*>(l)->cursor = 0; }
OGRFeatureH OGR_L_GetNextFeature(OGRLayerH lh) {
    ShimLayer* l = static_cast<ShimLayer*>(lh);
    if (l->cursor >= l->x.size()) return nullptr;
    return new ShimFeature{l, l->cursor++};
}
OGRFeatureH OGR_L_GetFeature(OGRLayerH lh, GIntBig i) {
    ShimLayer* l = static_cast<ShimLayer*>(lh);
    if (i < 0 || size_t(i) >= l->x.size()) return nullptr;
    return new ShimFeature{l, size_t(i)};
}
OGRGeometryH OGR_F_GetGeometryRef(OGRFeatureH f) { return f; }
double OGR_G_GetX(OGRGeometryH g, int) { ShimFeature* f = static_cast<ShimFeature*>(g); return f->layer->x[f->idx]; }
double OGR_G_GetY(OGRGeometryH g, int) { ShimFeature* f = static_cast<ShimFeature*>(g); return f->layer->y[f->idx]; }
int OGR_F_GetFieldIndex(OGRFeatureH, const char* name) { return (name && strcmp(name, "id") == 0) ? 0 : -1; }
OGRFieldDefnH OGR_FD_GetFieldDefn(OGRFeatureDefnH, int) { return &g_field_dummy; }
OGRFieldType OGR_Fld_GetType(OGRFieldDefnH) { return OFTInteger; }
int OGR_F_GetFieldAsInteger(OGRFeatureH fh, int) { ShimFeature* f = static_cast<ShimFeature*>(fh); return f->layer->id[f->idx]; }
GIntBig OGR_F_GetFieldAsInteger64(OGRFeatureH fh, int i) { return OGR_F_GetFieldAsInteger(fh, i); }
double OGR_F_GetFieldAsDouble(OGRFeatureH fh, int i) { return OGR_F_GetFieldAsInteger(fh, i); }
const char* OGR_F_GetFieldAsString(OGRFeatureH, int) { return ""; }
void OGR_F_Destroy(OGRFeatureH f) { delete static_cast<ShimFeature*>(f); }

}  // extern "C"
