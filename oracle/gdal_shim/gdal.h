/* Test infrastructure (oracle/): a stand-in for the GDAL/OGR C API headers.
 *
 * The reference (dtarb/TauDEM) links GDAL only for raster/vector file decode and encode; GDAL is
 * a third-party dependency that is not vendored under /root/reference and is not installed here.
 * This header declares just the entry points the hot-path sources call (src/tiffIO.cpp,
 * src/aread8.cpp:74-97, src/ReadOutlets.cpp, src/commonLib.cpp:388-474); shim.cpp implements them
 * over this repository's own GeoTIFF and outlet readers so that the UNMODIFIED reference sources
 * compile into oracle/_ref/ and read/write the same files as the HIP tools.
 * Nothing here is product code and nothing in the product may include it.
 */
#ifndef TDX_GDAL_SHIM_H
#define TDX_GDAL_SHIM_H
#include <cstdlib>
#include <cstring>
#include <string>

typedef void* GDALDatasetH;
typedef void* GDALRasterBandH;
typedef void* GDALDriverH;
typedef void* OGRSpatialReferenceH;
typedef void* OGRSFDriverH;
typedef void* OGRDataSourceH;
typedef void* OGRLayerH;
typedef void* OGRFeatureDefnH;
typedef void* OGRFieldDefnH;
typedef void* OGRFeatureH;
typedef void* OGRGeometryH;

typedef enum { GDT_Unknown = 0, GDT_Byte = 1, GDT_UInt16 = 2, GDT_Int16 = 3, GDT_UInt32 = 4, GDT_Int32 = 5, GDT_Float32 = 6, GDT_Float64 = 7 } GDALDataType;
typedef enum { GA_ReadOnly = 0, GA_Update = 1 } GDALAccess;
typedef enum { GF_Read = 0, GF_Write = 1 } GDALRWFlag;
typedef enum { CE_None = 0, CE_Debug = 1, CE_Warning = 2, CE_Failure = 3, CE_Fatal = 4 } CPLErr;
typedef enum { wkbUnknown = 0, wkbPoint = 1, wkbLineString = 2, wkbPolygon = 3, wkbMultiPoint = 4, wkbMultiLineString = 5, wkbMultiPolygon = 6, wkbGeometryCollection = 7 } OGRwkbGeometryType;
typedef enum { OFTInteger = 0, OFTIntegerList = 1, OFTReal = 2, OFTRealList = 3, OFTString = 4, OFTInteger64 = 12 } OGRFieldType;
typedef long long GIntBig;

#ifndef TRUE
#define TRUE 1
#endif
#ifndef FALSE
#define FALSE 0
#endif

extern "C" {
void GDALAllRegister(void);
GDALDatasetH GDALOpen(const char* filename, GDALAccess access);
void GDALClose(GDALDatasetH ds);
GDALDriverH GDALGetDatasetDriver(GDALDatasetH ds);
GDALDriverH GDALGetDriverByName(const char* name);
const char* GDALGetProjectionRef(GDALDatasetH ds);
CPLErr GDALSetProjection(GDALDatasetH ds, const char* wkt);
GDALRasterBandH GDALGetRasterBand(GDALDatasetH ds, int band);
const char* GDALGetRasterUnitType(GDALRasterBandH band);
int GDALGetRasterXSize(GDALDatasetH ds);
int GDALGetRasterYSize(GDALDatasetH ds);
CPLErr GDALGetGeoTransform(GDALDatasetH ds, double* gt);
CPLErr GDALSetGeoTransform(GDALDatasetH ds, double* gt);
double GDALGetRasterNoDataValue(GDALRasterBandH band, int* success);
CPLErr GDALSetRasterNoDataValue(GDALRasterBandH band, double v);
GDALDataType GDALGetRasterDataType(GDALRasterBandH band);
CPLErr GDALRasterIO(GDALRasterBandH band, GDALRWFlag rw, int xoff, int yoff, int xsize, int ysize, void* data,
                    int bxsize, int bysize, GDALDataType btype, int pixelspace, int linespace);
void GDALFlushCache(GDALDatasetH ds);
GDALDatasetH GDALCreate(GDALDriverH drv, const char* filename, int nx, int ny, int nbands, GDALDataType type, char** options);
char** CSLSetNameValue(char** list, const char* name, const char* value);
const char* CPLGetLastErrorMsg(void);

OGRSpatialReferenceH OSRNewSpatialReference(const char* wkt);
int OSRIsGeographic(OGRSpatialReferenceH srs);
int OSRIsProjected(OGRSpatialReferenceH srs);
double OSRGetLinearUnits(OGRSpatialReferenceH srs, char** name);
const char* OSRGetAttrValue(OGRSpatialReferenceH srs, const char* node, int child);

void OGRRegisterAll(void);
OGRDataSourceH OGROpen(const char* name, int update, OGRSFDriverH* drv);
void OGR_DS_Destroy(OGRDataSourceH ds);
int OGR_DS_GetLayerCount(OGRDataSourceH ds);
OGRLayerH OGR_DS_GetLayer(OGRDataSourceH ds, int i);
OGRLayerH OGR_DS_GetLayerByName(OGRDataSourceH ds, const char* name);
const char* OGR_L_GetName(OGRLayerH l);
OGRwkbGeometryType OGR_L_GetGeomType(OGRLayerH l);
OGRSpatialReferenceH OGR_L_GetSpatialRef(OGRLayerH l);
GIntBig OGR_L_GetFeatureCount(OGRLayerH l, int force);
OGRFeatureDefnH OGR_L_GetLayerDefn(OGRLayerH l);
void OGR_L_ResetReading(OGRLayerH l);
OGRFeatureH OGR_L_GetNextFeature(OGRLayerH l);
OGRFeatureH OGR_L_GetFeature(OGRLayerH l, GIntBig i);
OGRGeometryH OGR_F_GetGeometryRef(OGRFeatureH f);
double OGR_G_GetX(OGRGeometryH g, int i);
double OGR_G_GetY(OGRGeometryH g, int i);
int OGR_F_GetFieldIndex(OGRFeatureH f, const char* name);
OGRFieldDefnH OGR_FD_GetFieldDefn(OGRFeatureDefnH d, int i);
OGRFieldType OGR_Fld_GetType(OGRFieldDefnH f);
int OGR_F_GetFieldAsInteger(OGRFeatureH f, int i);
GIntBig OGR_F_GetFieldAsInteger64(OGRFeatureH f, int i);
double OGR_F_GetFieldAsDouble(OGRFeatureH f, int i);
const char* OGR_F_GetFieldAsString(OGRFeatureH f, int i);
void OGR_F_Destroy(OGRFeatureH f);
}
#endif
