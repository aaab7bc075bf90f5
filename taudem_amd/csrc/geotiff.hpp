// Minimal self-contained GeoTIFF / BigTIFF reader + writer for the TauDEM hot path.
//
// Replaces the raster-I/O boundary of the reference (src/tiffIO.cpp:54-428, which delegates the
// file format to GDAL): single-band rasters, strips or tiles, compression none / LZW / Deflate /
// PackBits on read, none / LZW on write, classic TIFF or BigTIFF, GeoTIFF georeferencing tags and
// the GDAL_NODATA tag.  Semantics kept from the reference:
//   * input nodata missing  -> -9999                       (src/tiffIO.cpp:161-167)
//   * dx,dy = |geotransform[1]|, |geotransform[5]|          (src/tiffIO.cpp:96-97)
//   * geographic rasters get per-row dxc/dyc in metres      (src/tiffIO.cpp:127-151, 434-445)
//   * BIGTIFF when cellbytes*X*Y/1e9 > 4.0                  (src/tiffIO.cpp:322-330)
//   * outputs copy geotransform + projection of the input   (src/tiffIO.cpp:344-349)
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace tdx {

enum class DType : int { I16 = 0, I32 = 1, F32 = 2 };   // order of DATA_TYPE, src/commonLib.h:65-69
inline size_t dtype_size(DType t) { return t == DType::I16 ? 2 : 4; }

// Georeferencing payload carried verbatim from an input file to the outputs derived from it.
struct GeoTags {
    std::vector<double> pixel_scale;      // tag 33550
    std::vector<double> tiepoints;        // tag 33922
    std::vector<double> transform;        // tag 34264
    std::vector<uint16_t> geokeys;        // tag 34735
    std::vector<double> geodoubles;       // tag 34736
    std::string geoascii;                 // tag 34737
};

struct RasterInfo {
    int64_t nx = 0, ny = 0;
    double gt[6] = {0, 1, 0, 0, 0, -1};   // GDAL-style geotransform
    bool has_nodata = false;
    double nodata = -9999.0;              // reference default when the file declares none
    bool geographic = false;
    int file_bits = 32;                   // bits per sample in the file
    int file_format = 3;                  // 1 uint, 2 int, 3 float
    GeoTags geo;
    // derived as in tiffIO::tiffIO (src/tiffIO.cpp:96-156)
    double dlon = 1, dlat = 1, xleftedge = 0, ytopedge = 0;
    std::vector<double> dxc, dyc;         // per-row cell size in metres
    double dxA() const;
    double dyA() const;
    void derive_cell_sizes();
};

class TiffReader {
public:
    TiffReader();
    ~TiffReader();
    // returns false (and sets error()) if the file cannot be opened / parsed
    bool open(const std::string& path);
    void close();
    const RasterInfo& info() const { return info_; }
    const std::string& error() const { return err_; }
    // Read a window converting to `out_type` with GDALRasterIO's conversion rules
    // (round-to-nearest + clamp for float -> int).  dst is row-major w*h.
    bool read_window(int64_t x0, int64_t y0, int64_t w, int64_t h, DType out_type, void* dst);
    // layout details (used by the in-place updater)
    bool is_plain_strips() const;         // uncompressed, strip-organised, native endian
    uint64_t strip_offset(int64_t row, int64_t* rows_in_strip) const;
private:
    struct Impl;
    Impl* p_;
    RasterInfo info_;
    std::string err_;
};

class TiffWriter {
public:
    TiffWriter();
    ~TiffWriter();
    // Creates the file. Uncompressed files are fully laid out at creation (so that rows may be
    // written in any order, or by a later open_update()); LZW files must be written top-down.
    bool create(const std::string& path, int64_t nx, int64_t ny, DType type, double nodata,
                const RasterInfo* georef_from, bool lzw);
    // Re-open an uncompressed file made by create() for writing more rows.
    bool open_update(const std::string& path);
    bool write_rows(int64_t y0, int64_t nrows, const void* src);   // src: nrows*nx of `type`
    // The whole raster (ny*nx of `type`) with up to `threads` host threads, into a file just made by create(): uncompressed files
    // are written as row bands side by side, LZW strips are encoded side by side and appended in strip order - the bytes of the
    // file do not depend on the thread count (src/tiffIO.cpp:382-427: every rank of the reference writes its own rows).
    bool write_all(const void* src, int threads);
    bool close();
    const std::string& error() const { return err_; }
    DType type() const;
    int64_t nx() const;
    int64_t ny() const;
private:
    struct Impl;
    Impl* p_;
    std::string err_;
};

// Output file name rule of tiffIO::write (src/tiffIO.cpp:268-306): no extension -> append
// ".tif"; extension lower-cased; extension not in the driver list -> replaced by "tif".
// Returns the driver index (0 = GTiff) or -1 for a recognised non-TIFF driver we cannot write.
int resolve_output_name(std::string& filename);

}  // namespace tdx
