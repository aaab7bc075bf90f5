// Minimal GeoTIFF / BigTIFF reader + writer (see geotiff.hpp for the reference citations).
#include "geotiff.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

namespace tdx {

// ----------------------------------------------------------------------------------------------
// RasterInfo: derived quantities exactly as tiffIO::tiffIO computes them.
// ----------------------------------------------------------------------------------------------
namespace {
// WGS84 constants and the truncated PI the reference uses (src/tiffIO.h:94-96, src/commonLib.h:76)
const double kPI = 3.14159265359;
const double kElipA = 6378137.000;
const double kElipB = 6356752.314;
const double kBoa = kElipB / kElipA;

// src/tiffIO.cpp:434-445 (geotoLength)
void geo_to_length(double dlon, double dlat, double lat, double* xyc) {
    double ds2, beta, dbeta;
    dlat = dlat * kPI / 180.;
    dlon = dlon * kPI / 180.;
    lat = lat * kPI / 180.;
    beta = atan(kBoa * tan(lat));
    dbeta = dlat * kBoa * (cos(beta) / cos(lat)) * (cos(beta) / cos(lat));
    ds2 = (pow(kElipA * sin(beta), 2) + pow(kElipB * cos(beta), 2)) * pow(dbeta, 2);
    xyc[0] = kElipA * cos(beta) * std::fabs(dlon);
    xyc[1] = double(sqrt(double(ds2)));
}
}  // namespace

void RasterInfo::derive_cell_sizes() {
    dlon = std::fabs(gt[1]);
    dlat = std::fabs(gt[5]);
    xleftedge = gt[0];
    ytopedge = gt[3];
    const double yllcenter = ytopedge - (double(ny) * dlat) - dlat / 2.;
    dxc.assign(size_t(ny), dlon);
    dyc.assign(size_t(ny), dlat);
    if (geographic) {
        for (int64_t j = 0; j < ny; j++) {
            // the reference stores the row latitude in a float (src/tiffIO.cpp:132)
            float rowlat = float(yllcenter + double(ny - j - 1) * dlat);
            double xy[2];
            geo_to_length(dlon, dlat, rowlat, xy);
            dxc[size_t(j)] = xy[0];
            dyc[size_t(j)] = xy[1];
        }
    }
}
double RasterInfo::dxA() const { return ny > 0 ? std::fabs(dxc[size_t(ny / 2)]) : 0.0; }
double RasterInfo::dyA() const { return ny > 0 ? std::fabs(dyc[size_t(ny / 2)]) : 0.0; }

// ----------------------------------------------------------------------------------------------
// low-level helpers
// ----------------------------------------------------------------------------------------------
namespace {

bool pread_all(int fd, void* buf, size_t n, uint64_t off) {
    char* p = static_cast<char*>(buf);
    while (n > 0) {
        ssize_t r = ::pread(fd, p, n, off_t(off));
        if (r <= 0) return false;
        p += r; n -= size_t(r); off += uint64_t(r);
    }
    return true;
}
bool pwrite_all(int fd, const void* buf, size_t n, uint64_t off) {
    const char* p = static_cast<const char*>(buf);
    while (n > 0) {
        ssize_t r = ::pwrite(fd, p, n, off_t(off));
        if (r <= 0) return false;
        p += r; n -= size_t(r); off += uint64_t(r);
    }
    return true;
}

inline void bswap(void* p, size_t n) {
    unsigned char* b = static_cast<unsigned char*>(p);
    for (size_t i = 0; i < n / 2; i++) std::swap(b[i], b[n - 1 - i]);
}

size_t tiff_type_size(int t) {
    switch (t) {
        case 1: case 2: case 6: case 7: return 1;
        case 3: case 8: return 2;
        case 4: case 9: case 11: case 13: return 4;
        case 5: case 10: case 12: case 16: case 17: case 18: return 8;
        default: return 0;
    }
}

struct TagValue {
    int type = 0;
    uint64_t count = 0;
    std::vector<unsigned char> raw;   // native-endian elements
    double as_double(size_t i) const {
        const unsigned char* q = raw.data() + i * tiff_type_size(type);
        switch (type) {
            case 1: case 7: return double(*q);
            case 6: return double(*reinterpret_cast<const int8_t*>(q));
            case 3: { uint16_t v; memcpy(&v, q, 2); return v; }
            case 8: { int16_t v; memcpy(&v, q, 2); return v; }
            case 4: case 13: { uint32_t v; memcpy(&v, q, 4); return v; }
            case 9: { int32_t v; memcpy(&v, q, 4); return v; }
            case 11: { float v; memcpy(&v, q, 4); return v; }
            case 12: { double v; memcpy(&v, q, 8); return v; }
            case 16: case 18: { uint64_t v; memcpy(&v, q, 8); return double(v); }
            case 17: { int64_t v; memcpy(&v, q, 8); return double(v); }
            case 5: { uint32_t a, b; memcpy(&a, q, 4); memcpy(&b, q + 4, 4); return b ? double(a) / b : 0; }
            case 10: { int32_t a, b; memcpy(&a, q, 4); memcpy(&b, q + 4, 4); return b ? double(a) / b : 0; }
            default: return 0;
        }
    }
    uint64_t as_u64(size_t i) const {
        const unsigned char* q = raw.data() + i * tiff_type_size(type);
        switch (type) {
            case 1: case 7: return *q;
            case 3: { uint16_t v; memcpy(&v, q, 2); return v; }
            case 4: case 13: { uint32_t v; memcpy(&v, q, 4); return v; }
            case 16: case 18: { uint64_t v; memcpy(&v, q, 8); return v; }
            default: return uint64_t(as_double(i));
        }
    }
};

// ---- TIFF LZW (MSB-first codes, 9..12 bits, "early change" as written by libtiff/GDAL) ----------
bool lzw_decode(const unsigned char* src, size_t n, std::vector<unsigned char>& out, size_t expect) {
    out.clear();
    out.reserve(expect);
    struct Entry { int prefix; int len; unsigned char ch; unsigned char first; };
    static thread_local std::vector<Entry> tab;
    tab.resize(4096);
    for (int i = 0; i < 256; i++) tab[size_t(i)] = {-1, 1, (unsigned char)i, (unsigned char)i};
    int next = 258, bits = 9, prev = -1;
    uint64_t acc = 0; int nacc = 0; size_t pos = 0;
    while (out.size() < expect) {
        while (nacc < bits && pos < n) { acc = (acc << 8) | src[pos++]; nacc += 8; }
        if (nacc < bits) break;
        const int code = int((acc >> (nacc - bits)) & ((1u << bits) - 1));
        nacc -= bits;
        if (code == 257) break;                                   // EOI
        if (code == 256) { next = 258; bits = 9; prev = -1; continue; }   // Clear
        if (prev < 0) {                                            // first code after a clear
            if (code >= 256) return false;
            out.push_back((unsigned char)code);
            prev = code;
            continue;
        }
        if (code > next || (code >= 256 && code < 258)) return false;
        const unsigned char firstc = (code < next) ? tab[size_t(code)].first : tab[size_t(prev)].first;
        if (next < 4096) {
            tab[size_t(next)] = {prev, tab[size_t(prev)].len + 1, firstc, tab[size_t(prev)].first};
            next++;
        } else if (code >= next) return false;
        const int len = tab[size_t(code)].len;
        const size_t base = out.size();
        out.resize(base + size_t(len));
        int cur = code;
        for (int i = len - 1; i >= 0; i--) { out[base + size_t(i)] = tab[size_t(cur)].ch; cur = tab[size_t(cur)].prefix; }
        prev = code;
        if (next >= (1 << bits) - 1 && bits < 12) bits++;
    }
    return true;
}

void lzw_encode(const unsigned char* src, size_t n, std::vector<unsigned char>& out) {
    out.clear();
    uint64_t acc = 0; int nacc = 0;
    auto put = [&](int code, int nb) {
        acc = (acc << nb) | uint64_t(code); nacc += nb;
        while (nacc >= 8) { out.push_back((unsigned char)((acc >> (nacc - 8)) & 0xff)); nacc -= 8; }
    };
    const int HSIZE = 9001;
    static thread_local std::vector<int> hkey, hval;
    hkey.assign(HSIZE, -1); hval.assign(HSIZE, 0);
    int next = 258, bits = 9;
    put(256, bits);
    if (n > 0) {
        int cur = src[0];
        for (size_t i = 1; i < n; i++) {
            const int c = src[i];
            const int key = (cur << 8) | c;
            int h = key % HSIZE;
            bool found = false;
            while (hkey[size_t(h)] != -1) {
                if (hkey[size_t(h)] == key) { cur = hval[size_t(h)]; found = true; break; }
                if (++h == HSIZE) h = 0;
            }
            if (found) continue;
            put(cur, bits);
            hkey[size_t(h)] = key; hval[size_t(h)] = next++;
            if (next == 4094) {                 // table full: clear and restart
                put(256, bits);
                hkey.assign(HSIZE, -1);
                next = 258; bits = 9;
            } else if (next > (1 << bits) - 1) bits++;
            cur = c;
        }
        put(cur, bits);
        next++;
        if (next == 4094) { put(256, bits); bits = 9; }
        else if (next > (1 << bits) - 1) bits++;
    }
    put(257, bits);
    if (nacc) out.push_back((unsigned char)((acc << (8 - nacc)) & 0xff));
}

bool packbits_decode(const unsigned char* src, size_t n, std::vector<unsigned char>& out, size_t expect) {
    out.clear(); out.reserve(expect);
    size_t i = 0;
    while (i < n && out.size() < expect) {
        int c = (signed char)src[i++];
        if (c >= 0) { size_t k = size_t(c) + 1; if (i + k > n) return false; out.insert(out.end(), src + i, src + i + k); i += k; }
        else if (c != -128) { size_t k = size_t(1 - c); if (i >= n) return false; out.insert(out.end(), k, src[i]); i++; }
    }
    return true;
}

inline int16_t to_i16(double v, bool from_float) {
    if (from_float) {
        if (v != v) return 0;
        v = (v >= 0) ? std::floor(v + 0.5) : std::ceil(v - 0.5);   // GDALCopyWords rounding
    }
    if (v > 32767.0) return 32767;
    if (v < -32768.0) return -32768;
    return int16_t(v);
}
inline int32_t to_i32(double v, bool from_float) {
    if (from_float) {
        if (v != v) return 0;
        v = (v >= 0) ? std::floor(v + 0.5) : std::ceil(v - 0.5);
    }
    if (v > 2147483647.0) return 2147483647;
    if (v < -2147483648.0) return int32_t(-2147483647 - 1);
    return int32_t(v);
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// Reader
// ----------------------------------------------------------------------------------------------
struct TiffReader::Impl {
    int fd = -1;
    bool big = false, swap = false;
    std::map<int, TagValue> tags;
    int compression = 1, predictor = 1, planar = 1, spp = 1;
    bool tiled = false;
    int64_t tile_w = 0, tile_h = 0, rows_per_strip = 0;
    std::vector<uint64_t> offsets, counts;
    // decoded-chunk cache
    int64_t cached_chunk = -1;
    std::vector<unsigned char> chunk, scratch;
};

TiffReader::TiffReader() : p_(new Impl) {}
TiffReader::~TiffReader() { close(); delete p_; }
void TiffReader::close() { if (p_->fd >= 0) { ::close(p_->fd); p_->fd = -1; } }

bool TiffReader::open(const std::string& path) {
    Impl& m = *p_;
    close();
    m.tags.clear(); m.cached_chunk = -1;
    m.fd = ::open(path.c_str(), O_RDONLY);
    if (m.fd < 0) { err_ = "cannot open " + path; return false; }
    unsigned char hdr[16];
    if (!pread_all(m.fd, hdr, 8, 0)) { err_ = "short file"; return false; }
    bool le;
    if (hdr[0] == 'I' && hdr[1] == 'I') le = true;
    else if (hdr[0] == 'M' && hdr[1] == 'M') le = false;
    else { err_ = "not a TIFF file: " + path; return false; }
    const uint16_t probe = 1;
    const bool host_le = *reinterpret_cast<const unsigned char*>(&probe) == 1;
    m.swap = (le != host_le);
    auto rd16 = [&](const unsigned char* q) { uint16_t v; memcpy(&v, q, 2); if (m.swap) bswap(&v, 2); return v; };
    auto rd32 = [&](const unsigned char* q) { uint32_t v; memcpy(&v, q, 4); if (m.swap) bswap(&v, 4); return v; };
    auto rd64 = [&](const unsigned char* q) { uint64_t v; memcpy(&v, q, 8); if (m.swap) bswap(&v, 8); return v; };
    uint16_t magic = rd16(hdr + 2);
    uint64_t ifd_off;
    if (magic == 42) { m.big = false; ifd_off = rd32(hdr + 4); }
    else if (magic == 43) {
        m.big = true;
        if (!pread_all(m.fd, hdr, 16, 0)) { err_ = "short BigTIFF header"; return false; }
        ifd_off = rd64(hdr + 8);
    } else { err_ = "bad TIFF magic"; return false; }

    uint64_t nent;
    unsigned char cb[8];
    if (m.big) { if (!pread_all(m.fd, cb, 8, ifd_off)) { err_ = "bad IFD"; return false; } nent = rd64(cb); ifd_off += 8; }
    else { if (!pread_all(m.fd, cb, 2, ifd_off)) { err_ = "bad IFD"; return false; } nent = rd16(cb); ifd_off += 2; }
    const size_t esz = m.big ? 20 : 12;
    std::vector<unsigned char> ents(size_t(nent) * esz);
    if (!pread_all(m.fd, ents.data(), ents.size(), ifd_off)) { err_ = "bad IFD entries"; return false; }
    for (uint64_t e = 0; e < nent; e++) {
        const unsigned char* q = ents.data() + e * esz;
        int tag = rd16(q), type = rd16(q + 2);
        uint64_t count = m.big ? rd64(q + 4) : rd32(q + 4);
        size_t tsz = tiff_type_size(type);
        if (tsz == 0) continue;
        uint64_t bytes = count * tsz;
        TagValue tv; tv.type = type; tv.count = count; tv.raw.resize(size_t(bytes));
        const size_t inl = m.big ? 8 : 4;
        const unsigned char* vp = q + (m.big ? 12 : 8);
        if (bytes <= inl) memcpy(tv.raw.data(), vp, size_t(bytes));
        else {
            uint64_t off = m.big ? rd64(vp) : rd32(vp);
            if (!pread_all(m.fd, tv.raw.data(), size_t(bytes), off)) { err_ = "bad tag data"; return false; }
        }
        if (m.swap && tsz > 1) {
            size_t unit = (type == 5 || type == 10) ? 4 : tsz;
            for (size_t i = 0; i + unit <= tv.raw.size(); i += unit) bswap(tv.raw.data() + i, unit);
        }
        m.tags[tag] = std::move(tv);
    }
    auto has = [&](int t) { return m.tags.count(t) > 0; };
    auto geti = [&](int t, int64_t def) { return has(t) && m.tags[t].count ? int64_t(m.tags[t].as_u64(0)) : def; };

    info_ = RasterInfo();
    info_.nx = geti(256, 0); info_.ny = geti(257, 0);
    info_.file_bits = int(geti(258, 1));
    info_.file_format = int(geti(339, 1));
    m.compression = int(geti(259, 1));
    m.predictor = int(geti(317, 1));
    m.planar = int(geti(284, 1));
    m.spp = int(geti(277, 1));
    if (info_.nx <= 0 || info_.ny <= 0) { err_ = "bad raster size"; return false; }
    if (m.spp != 1 && m.planar != 2) { err_ = "multi-sample pixel-interleaved TIFF not supported"; return false; }
    if (info_.file_format == 4) info_.file_format = 1;   // "undefined" -> treat as uint
    if (!(info_.file_bits == 8 || info_.file_bits == 16 || info_.file_bits == 32 || info_.file_bits == 64)) {
        err_ = "unsupported bits per sample"; return false;
    }
    if (has(322)) {
        m.tiled = true; m.tile_w = geti(322, 0); m.tile_h = geti(323, 0);
        if (!has(324) || !has(325)) { err_ = "missing tile offsets"; return false; }
        const TagValue &o = m.tags[324], &c = m.tags[325];
        m.offsets.resize(size_t(o.count)); m.counts.resize(size_t(c.count));
        for (size_t i = 0; i < o.count; i++) m.offsets[i] = o.as_u64(i);
        for (size_t i = 0; i < c.count; i++) m.counts[i] = c.as_u64(i);
    } else {
        m.tiled = false;
        m.rows_per_strip = geti(278, info_.ny);
        if (m.rows_per_strip <= 0 || m.rows_per_strip > info_.ny) m.rows_per_strip = info_.ny;
        if (!has(273)) { err_ = "missing strip offsets"; return false; }
        const TagValue& o = m.tags[273];
        m.offsets.resize(size_t(o.count));
        for (size_t i = 0; i < o.count; i++) m.offsets[i] = o.as_u64(i);
        if (has(279)) {
            const TagValue& c = m.tags[279];
            m.counts.resize(size_t(c.count));
            for (size_t i = 0; i < c.count; i++) m.counts[i] = c.as_u64(i);
        } else {
            m.counts.assign(m.offsets.size(), uint64_t(m.rows_per_strip) * uint64_t(info_.nx) * uint64_t(info_.file_bits / 8));
        }
    }
    // georeferencing
    auto getd = [&](int t, std::vector<double>& v) { if (has(t)) { v.resize(size_t(m.tags[t].count)); for (size_t i = 0; i < v.size(); i++) v[i] = m.tags[t].as_double(i); } };
    getd(33550, info_.geo.pixel_scale); getd(33922, info_.geo.tiepoints); getd(34264, info_.geo.transform);
    getd(34736, info_.geo.geodoubles);
    if (has(34735)) { const TagValue& g = m.tags[34735]; info_.geo.geokeys.resize(size_t(g.count)); for (size_t i = 0; i < g.count; i++) info_.geo.geokeys[i] = uint16_t(g.as_u64(i)); }
    if (has(34737)) { const TagValue& g = m.tags[34737]; info_.geo.geoascii.assign(reinterpret_cast<const char*>(g.raw.data()), g.raw.size()); }
    bool pixel_is_point = false;
    const std::vector<uint16_t>& gk = info_.geo.geokeys;
    if (gk.size() >= 4) {
        size_t nk = gk[3];
        for (size_t k = 0; k < nk && 4 + 4 * k + 3 < gk.size(); k++) {
            uint16_t id = gk[4 + 4 * k], loc = gk[5 + 4 * k], val = gk[7 + 4 * k];
            if (id == 1024 && loc == 0 && val == 2) info_.geographic = true;
            if (id == 1025 && loc == 0 && val == 2) pixel_is_point = true;
        }
    }
    if (info_.geo.pixel_scale.size() >= 2 && info_.geo.tiepoints.size() >= 6) {
        const double sx = info_.geo.pixel_scale[0], sy = info_.geo.pixel_scale[1];
        const std::vector<double>& tp = info_.geo.tiepoints;
        info_.gt[0] = tp[3] - tp[0] * sx; info_.gt[1] = sx; info_.gt[2] = 0;
        info_.gt[3] = tp[4] + tp[1] * sy; info_.gt[4] = 0; info_.gt[5] = -sy;
        if (pixel_is_point) { info_.gt[0] -= 0.5 * sx; info_.gt[3] += 0.5 * sy; }
    } else if (info_.geo.transform.size() >= 16) {
        const std::vector<double>& t = info_.geo.transform;
        info_.gt[0] = t[3]; info_.gt[1] = t[0]; info_.gt[2] = t[1];
        info_.gt[3] = t[7]; info_.gt[4] = t[4]; info_.gt[5] = t[5];
    }
    if (has(42113)) {
        const TagValue& nd = m.tags[42113];
        std::string s(reinterpret_cast<const char*>(nd.raw.data()), nd.raw.size());
        char* endp = nullptr;
        double v = strtod(s.c_str(), &endp);
        if (endp != s.c_str()) { info_.has_nodata = true; info_.nodata = v; }
    }
    info_.derive_cell_sizes();
    return true;
}

bool TiffReader::is_plain_strips() const {
    const Impl& m = *p_;
    return !m.tiled && m.compression == 1 && !m.swap && m.spp == 1;
}
uint64_t TiffReader::strip_offset(int64_t row, int64_t* rows_in_strip) const {
    const Impl& m = *p_;
    int64_t s = row / m.rows_per_strip;
    if (rows_in_strip) *rows_in_strip = std::min<int64_t>(m.rows_per_strip, info_.ny - s * m.rows_per_strip);
    return m.offsets[size_t(s)];
}

namespace {
// decode one chunk (strip or tile) into `out` as native-endian samples, row-major cw x ch
bool decode_chunk(int fd, bool swap, int compression, int predictor, int bytes_ps, int file_format,
                  uint64_t off, uint64_t cnt, int64_t cw, int64_t ch,
                  std::vector<unsigned char>& scratch, std::vector<unsigned char>& out, std::string& err) {
    const size_t expect = size_t(cw) * size_t(ch) * size_t(bytes_ps);
    if (compression == 1) {
        out.resize(expect);
        size_t n = std::min<uint64_t>(cnt, expect);
        if (n && !pread_all(fd, out.data(), n, off)) { err = "short read"; return false; }
        if (n < expect) memset(out.data() + n, 0, expect - n);
    } else {
        scratch.resize(size_t(cnt));
        if (cnt && !pread_all(fd, scratch.data(), size_t(cnt), off)) { err = "short read"; return false; }
        if (compression == 5) {
            if (!lzw_decode(scratch.data(), scratch.size(), out, expect)) { err = "LZW decode error"; return false; }
        } else if (compression == 8 || compression == 32946) {
            out.resize(expect);
            uLongf dl = uLongf(expect);
            int rc = uncompress(out.data(), &dl, scratch.data(), uLong(scratch.size()));
            if (rc != Z_OK && rc != Z_BUF_ERROR) { err = "deflate decode error"; return false; }
            if (dl < expect) memset(out.data() + dl, 0, expect - dl);
        } else if (compression == 32773) {
            if (!packbits_decode(scratch.data(), scratch.size(), out, expect)) { err = "PackBits decode error"; return false; }
        } else { err = "unsupported TIFF compression " + std::to_string(compression); return false; }
        if (out.size() < expect) out.resize(expect, 0);
    }
    const size_t rowbytes = size_t(cw) * size_t(bytes_ps);
    if (predictor == 3 && file_format == 3) {
        // floating point predictor: undo byte differencing, then de-interleave the byte planes
        std::vector<unsigned char> row(rowbytes);
        for (int64_t r = 0; r < ch; r++) {
            unsigned char* q = out.data() + size_t(r) * rowbytes;
            for (size_t i = 1; i < rowbytes; i++) q[i] = (unsigned char)(q[i] + q[i - 1]);
            memcpy(row.data(), q, rowbytes);
            for (int64_t x = 0; x < cw; x++)
                for (int b = 0; b < bytes_ps; b++)   // planes are stored most-significant byte first
                    q[size_t(x) * size_t(bytes_ps) + size_t(bytes_ps - 1 - b)] = row[size_t(b) * size_t(cw) + size_t(x)];
        }
        return true;   // now native little-endian
    }
    if (swap && bytes_ps > 1)
        for (size_t i = 0; i + size_t(bytes_ps) <= out.size(); i += size_t(bytes_ps)) bswap(out.data() + i, size_t(bytes_ps));
    if (predictor == 2) {
        for (int64_t r = 0; r < ch; r++) {
            unsigned char* q = out.data() + size_t(r) * rowbytes;
            if (bytes_ps == 1) for (int64_t x = 1; x < cw; x++) q[x] = (unsigned char)(q[x] + q[x - 1]);
            else if (bytes_ps == 2) { uint16_t* v = reinterpret_cast<uint16_t*>(q); for (int64_t x = 1; x < cw; x++) v[x] = uint16_t(v[x] + v[x - 1]); }
            else if (bytes_ps == 4) { uint32_t* v = reinterpret_cast<uint32_t*>(q); for (int64_t x = 1; x < cw; x++) v[x] = v[x] + v[x - 1]; }
            else if (bytes_ps == 8) { uint64_t* v = reinterpret_cast<uint64_t*>(q); for (int64_t x = 1; x < cw; x++) v[x] = v[x] + v[x - 1]; }
        }
    }
    return true;
}

inline double sample_as_double(const unsigned char* q, int bits, int fmt) {
    switch (bits) {
        case 8: return fmt == 2 ? double(*reinterpret_cast<const int8_t*>(q)) : double(*q);
        case 16: { if (fmt == 2) { int16_t v; memcpy(&v, q, 2); return v; } uint16_t v; memcpy(&v, q, 2); return v; }
        case 32: {
            if (fmt == 3) { float v; memcpy(&v, q, 4); return v; }
            if (fmt == 2) { int32_t v; memcpy(&v, q, 4); return v; }
            uint32_t v; memcpy(&v, q, 4); return v;
        }
        default: {
            if (fmt == 3) { double v; memcpy(&v, q, 8); return v; }
            if (fmt == 2) { int64_t v; memcpy(&v, q, 8); return double(v); }
            uint64_t v; memcpy(&v, q, 8); return double(v);
        }
    }
}
}  // namespace

bool TiffReader::read_window(int64_t x0, int64_t y0, int64_t w, int64_t h, DType out_type, void* dst) {
    Impl& m = *p_;
    if (m.fd < 0) { err_ = "file not open"; return false; }
    if (x0 < 0 || y0 < 0 || x0 + w > info_.nx || y0 + h > info_.ny) { err_ = "window outside raster"; return false; }
    const int bps = info_.file_bits / 8;
    const int fmt = info_.file_format;
    const bool same = (out_type == DType::F32 && fmt == 3 && bps == 4) || (out_type == DType::I16 && fmt == 2 && bps == 2) ||
                      (out_type == DType::I32 && fmt == 2 && bps == 4);
    const size_t osz = dtype_size(out_type);
    // fast path: plain strips of the requested type
    if (same && is_plain_strips()) {
        for (int64_t r = 0; r < h; r++) {
            int64_t row = y0 + r;
            int64_t s = row / m.rows_per_strip;
            uint64_t off = m.offsets[size_t(s)] + (uint64_t(row - s * m.rows_per_strip) * uint64_t(info_.nx) + uint64_t(x0)) * osz;
            if (!pread_all(m.fd, static_cast<char*>(dst) + size_t(r) * size_t(w) * osz, size_t(w) * osz, off)) { err_ = "short read"; return false; }
        }
        return true;
    }
    const int64_t cw = m.tiled ? m.tile_w : info_.nx;
    const int64_t chh = m.tiled ? m.tile_h : m.rows_per_strip;
    const int64_t tiles_x = m.tiled ? (info_.nx + cw - 1) / cw : 1;
    for (int64_t ty = y0 / chh; ty * chh < y0 + h; ty++) {
        for (int64_t tx = x0 / cw; tx * cw < x0 + w; tx++) {
            int64_t idx = ty * tiles_x + tx;
            if (idx < 0 || size_t(idx) >= m.offsets.size()) { err_ = "chunk index out of range"; return false; }
            int64_t ch_rows = m.tiled ? chh : std::min<int64_t>(chh, info_.ny - ty * chh);
            if (m.cached_chunk != idx) {
                if (!decode_chunk(m.fd, m.swap, m.compression, m.predictor, bps, fmt, m.offsets[size_t(idx)], m.counts[size_t(idx)],
                                  cw, ch_rows, m.scratch, m.chunk, err_)) return false;
                m.cached_chunk = idx;
            }
            int64_t rx0 = std::max(x0, tx * cw), rx1 = std::min(x0 + w, (tx + 1) * cw);
            int64_t ry0 = std::max(y0, ty * chh), ry1 = std::min(y0 + h, ty * chh + ch_rows);
            for (int64_t yy = ry0; yy < ry1; yy++) {
                const unsigned char* srow = m.chunk.data() + (size_t(yy - ty * chh) * size_t(cw) + size_t(rx0 - tx * cw)) * size_t(bps);
                char* drow = static_cast<char*>(dst) + (size_t(yy - y0) * size_t(w) + size_t(rx0 - x0)) * osz;
                if (same) { memcpy(drow, srow, size_t(rx1 - rx0) * osz); continue; }
                for (int64_t xx = 0; xx < rx1 - rx0; xx++) {
                    double v = sample_as_double(srow + size_t(xx) * size_t(bps), info_.file_bits, fmt);
                    if (out_type == DType::F32) reinterpret_cast<float*>(drow)[xx] = float(v);
                    else if (out_type == DType::I16) reinterpret_cast<int16_t*>(drow)[xx] = to_i16(v, fmt == 3);
                    else reinterpret_cast<int32_t*>(drow)[xx] = to_i32(v, fmt == 3);
                }
            }
        }
    }
    return true;
}

// ----------------------------------------------------------------------------------------------
// Writer
// ----------------------------------------------------------------------------------------------
struct TiffWriter::Impl {
    int fd = -1;
    bool big = false, lzw = false, update = false;
    int64_t nx = 0, ny = 0, rps = 1;
    DType type = DType::F32;
    std::vector<uint64_t> offsets, counts;
    uint64_t data_end = 0;
    // LZW state
    int64_t next_row = 0;
    std::vector<unsigned char> stripbuf, enc;
    // deferred IFD (LZW)
    double nodata = 0;
    GeoTags geo;
    bool have_geo = false;
};

TiffWriter::TiffWriter() : p_(new Impl) {}
TiffWriter::~TiffWriter() { close(); delete p_; }
DType TiffWriter::type() const { return p_->type; }
int64_t TiffWriter::nx() const { return p_->nx; }
int64_t TiffWriter::ny() const { return p_->ny; }

namespace {
struct OutTag { int tag; int type; uint64_t count; std::vector<unsigned char> data; };
template <class T> void push_vals(OutTag& t, const T* v, size_t n) { t.data.resize(n * sizeof(T)); memcpy(t.data.data(), v, n * sizeof(T)); }

std::string format_nodata(double nd, DType type) {
    char buf[64];
    if (type != DType::F32) { snprintf(buf, sizeof buf, "%.0f", nd); return buf; }
    if (nd != nd) return "nan";
    // shortest representation that round-trips the double
    for (int prec = 1; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*g", prec, nd);
        if (strtod(buf, nullptr) == nd) break;
    }
    return buf;
}

// Serialises an IFD at file offset `ifd_off`; returns bytes (IFD followed by out-of-line values).
std::vector<unsigned char> build_ifd(bool big, uint64_t ifd_off, std::vector<OutTag>& tags) {
    std::sort(tags.begin(), tags.end(), [](const OutTag& a, const OutTag& b) { return a.tag < b.tag; });
    const size_t esz = big ? 20 : 12, inl = big ? 8 : 4;
    size_t ifd_bytes = (big ? 8 : 2) + tags.size() * esz + (big ? 8 : 4);
    std::vector<unsigned char> out(ifd_bytes, 0);
    uint64_t extra_off = ifd_off + ifd_bytes;
    if (extra_off & 1) { out.push_back(0); extra_off++; }
    if (big) { uint64_t n = tags.size(); memcpy(out.data(), &n, 8); } else { uint16_t n = uint16_t(tags.size()); memcpy(out.data(), &n, 2); }
    size_t pos = big ? 8 : 2;
    for (OutTag& t : tags) {
        uint16_t tg = uint16_t(t.tag), ty = uint16_t(t.type);
        memcpy(&out[pos], &tg, 2); memcpy(&out[pos + 2], &ty, 2);
        if (big) { uint64_t c = t.count; memcpy(&out[pos + 4], &c, 8); } else { uint32_t c = uint32_t(t.count); memcpy(&out[pos + 4], &c, 4); }
        size_t vpos = pos + (big ? 12 : 8);
        if (t.data.size() <= inl) memcpy(&out[vpos], t.data.data(), t.data.size());
        else {
            if (big) { uint64_t o = extra_off; memcpy(&out[vpos], &o, 8); } else { uint32_t o = uint32_t(extra_off); memcpy(&out[vpos], &o, 4); }
            out.insert(out.end(), t.data.begin(), t.data.end());
            extra_off += t.data.size();
            if (extra_off & 1) { out.push_back(0); extra_off++; }
        }
        pos += esz;
    }
    return out;
}

void make_tags(std::vector<OutTag>& tags, bool big, int64_t nx, int64_t ny, DType type, bool lzw, int64_t rps,
               const std::vector<uint64_t>& offsets, const std::vector<uint64_t>& counts, double nodata,
               const GeoTags* geo) {
    auto add_short = [&](int tag, uint16_t v) { OutTag t{tag, 3, 1, {}}; push_vals(t, &v, 1); tags.push_back(t); };
    auto add_long = [&](int tag, uint32_t v) { OutTag t{tag, 4, 1, {}}; push_vals(t, &v, 1); tags.push_back(t); };
    add_long(256, uint32_t(nx)); add_long(257, uint32_t(ny));
    add_short(258, uint16_t(dtype_size(type) * 8));
    add_short(259, lzw ? 5 : 1);
    add_short(262, 1);
    add_short(277, 1);
    add_long(278, uint32_t(rps));
    add_short(284, 1);
    add_short(339, type == DType::F32 ? 3 : 2);
    if (big) {
        OutTag o{273, 16, offsets.size(), {}}; push_vals(o, offsets.data(), offsets.size()); tags.push_back(o);
        OutTag c{279, 16, counts.size(), {}}; push_vals(c, counts.data(), counts.size()); tags.push_back(c);
    } else {
        std::vector<uint32_t> o32(offsets.begin(), offsets.end()), c32(counts.begin(), counts.end());
        OutTag o{273, 4, o32.size(), {}}; push_vals(o, o32.data(), o32.size()); tags.push_back(o);
        OutTag c{279, 4, c32.size(), {}}; push_vals(c, c32.data(), c32.size()); tags.push_back(c);
    }
    if (geo) {
        auto add_d = [&](int tag, const std::vector<double>& v) { if (v.empty()) return; OutTag t{tag, 12, v.size(), {}}; push_vals(t, v.data(), v.size()); tags.push_back(t); };
        add_d(33550, geo->pixel_scale); add_d(33922, geo->tiepoints); add_d(34264, geo->transform); add_d(34736, geo->geodoubles);
        if (!geo->geokeys.empty()) { OutTag t{34735, 3, geo->geokeys.size(), {}}; push_vals(t, geo->geokeys.data(), geo->geokeys.size()); tags.push_back(t); }
        if (!geo->geoascii.empty()) { OutTag t{34737, 2, geo->geoascii.size(), {}}; push_vals(t, geo->geoascii.data(), geo->geoascii.size()); tags.push_back(t); }
    }
    std::string nds = format_nodata(nodata, type);
    OutTag nt{42113, 2, nds.size() + 1, {}}; nt.data.assign(nds.begin(), nds.end()); nt.data.push_back(0); tags.push_back(nt);
}
}  // namespace

bool TiffWriter::create(const std::string& path, int64_t nx, int64_t ny, DType type, double nodata,
                        const RasterInfo* georef_from, bool lzw) {
    Impl& m = *p_;
    close();
    m = Impl();
    m.nx = nx; m.ny = ny; m.type = type; m.lzw = lzw; m.nodata = nodata;
    if (georef_from) { m.geo = georef_from->geo; m.have_geo = true; }
    const size_t cb = dtype_size(type);
    const double fileGB = double(cb) * double(nx) * double(ny) / 1000000000.0;   // src/tiffIO.cpp:324-325
    m.big = fileGB > 4.0;
    const uint64_t rowbytes = uint64_t(nx) * cb;
    m.rps = std::max<int64_t>(1, std::min<int64_t>(ny, int64_t((1u << 20) / std::max<uint64_t>(rowbytes, 1))));
    const int64_t nstrips = (ny + m.rps - 1) / m.rps;
    m.fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (m.fd < 0) { err_ = "cannot create " + path; return false; }
    m.offsets.assign(size_t(nstrips), 0); m.counts.assign(size_t(nstrips), 0);
    const uint64_t hdr = m.big ? 16 : 8;
    if (!lzw) {
        // layout: header | IFD + values | data
        for (int64_t s = 0; s < nstrips; s++) m.counts[size_t(s)] = uint64_t(std::min<int64_t>(m.rps, ny - s * m.rps)) * rowbytes;
        std::vector<OutTag> tags;
        make_tags(tags, m.big, nx, ny, type, false, m.rps, m.offsets, m.counts, nodata, m.have_geo ? &m.geo : nullptr);
        std::vector<unsigned char> ifd = build_ifd(m.big, hdr, tags);
        uint64_t data0 = (hdr + ifd.size() + 15) & ~uint64_t(15);
        for (int64_t s = 0; s < nstrips; s++) m.offsets[size_t(s)] = data0 + uint64_t(s) * uint64_t(m.rps) * rowbytes;
        tags.clear();
        make_tags(tags, m.big, nx, ny, type, false, m.rps, m.offsets, m.counts, nodata, m.have_geo ? &m.geo : nullptr);
        ifd = build_ifd(m.big, hdr, tags);
        unsigned char h[16] = {'I', 'I'};
        if (m.big) { uint16_t v = 43, b = 8, z = 0; memcpy(h + 2, &v, 2); memcpy(h + 4, &b, 2); memcpy(h + 6, &z, 2); uint64_t o = hdr; memcpy(h + 8, &o, 8); }
        else { uint16_t v = 42; memcpy(h + 2, &v, 2); uint32_t o = uint32_t(hdr); memcpy(h + 4, &o, 4); }
        if (!pwrite_all(m.fd, h, size_t(hdr), 0) || !pwrite_all(m.fd, ifd.data(), ifd.size(), hdr)) { err_ = "write error"; return false; }
        m.data_end = data0 + uint64_t(ny) * rowbytes;
        if (ftruncate(m.fd, off_t(m.data_end)) != 0) { err_ = "cannot size file"; return false; }
    } else {
        unsigned char h[16] = {0};
        if (!pwrite_all(m.fd, h, size_t(hdr), 0)) { err_ = "write error"; return false; }
        m.data_end = hdr;
        m.next_row = 0;
        m.stripbuf.clear();
    }
    return true;
}

bool TiffWriter::open_update(const std::string& path) {
    Impl& m = *p_;
    close();
    m = Impl();
    TiffReader rd;
    if (!rd.open(path)) { err_ = rd.error(); return false; }
    if (!rd.is_plain_strips()) { err_ = "file is not an uncompressed strip TIFF: " + path; return false; }
    const RasterInfo& ri = rd.info();
    m.nx = ri.nx; m.ny = ri.ny;
    if (ri.file_format == 3 && ri.file_bits == 32) m.type = DType::F32;
    else if (ri.file_format == 2 && ri.file_bits == 16) m.type = DType::I16;
    else if (ri.file_format == 2 && ri.file_bits == 32) m.type = DType::I32;
    else { err_ = "unsupported sample type for update"; return false; }
    int64_t rin = 0;
    rd.strip_offset(0, &rin);
    m.rps = rin;
    const int64_t nstrips = (m.ny + m.rps - 1) / m.rps;
    m.offsets.resize(size_t(nstrips));
    for (int64_t s = 0; s < nstrips; s++) m.offsets[size_t(s)] = rd.strip_offset(s * m.rps, nullptr);
    rd.close();
    m.update = true; m.lzw = false;
    m.fd = ::open(path.c_str(), O_RDWR);
    if (m.fd < 0) { err_ = "cannot open for update " + path; return false; }
    return true;
}

bool TiffWriter::write_rows(int64_t y0, int64_t nrows, const void* src) {
    Impl& m = *p_;
    if (m.fd < 0) { err_ = "file not open"; return false; }
    if (y0 < 0 || y0 + nrows > m.ny) { err_ = "rows outside raster"; return false; }
    const size_t cb = dtype_size(m.type);
    const uint64_t rowbytes = uint64_t(m.nx) * cb;
    const char* s = static_cast<const char*>(src);
    if (!m.lzw) {
        int64_t r = 0;
        while (r < nrows) {
            int64_t row = y0 + r;
            int64_t st = row / m.rps, in = row - st * m.rps;
            int64_t n = std::min<int64_t>(nrows - r, std::min<int64_t>(m.rps, m.ny - st * m.rps) - in);
            if (!pwrite_all(m.fd, s + uint64_t(r) * rowbytes, size_t(uint64_t(n) * rowbytes), m.offsets[size_t(st)] + uint64_t(in) * rowbytes)) { err_ = "write error"; return false; }
            r += n;
        }
        return true;
    }
    if (y0 != m.next_row) { err_ = "LZW output must be written top-down"; return false; }
    for (int64_t r = 0; r < nrows; r++) {
        m.stripbuf.insert(m.stripbuf.end(), s + uint64_t(r) * rowbytes, s + uint64_t(r + 1) * rowbytes);
        m.next_row++;
        int64_t st = (m.next_row - 1) / m.rps;
        bool strip_done = (m.next_row % m.rps == 0) || m.next_row == m.ny;
        if (strip_done) {
            lzw_encode(m.stripbuf.data(), m.stripbuf.size(), m.enc);
            if (!pwrite_all(m.fd, m.enc.data(), m.enc.size(), m.data_end)) { err_ = "write error"; return false; }
            m.offsets[size_t(st)] = m.data_end; m.counts[size_t(st)] = m.enc.size();
            m.data_end += m.enc.size();
            if (m.data_end & 1) { unsigned char z = 0; pwrite_all(m.fd, &z, 1, m.data_end); m.data_end++; }
            m.stripbuf.clear();
        }
    }
    return true;
}

bool TiffWriter::write_all(const void* src, int threads) {
    Impl& m = *p_;
    if (m.fd < 0) { err_ = "file not open"; return false; }
    const size_t cb = dtype_size(m.type);
    const uint64_t rowbytes = uint64_t(m.nx) * cb;
    const char* s = static_cast<const char*>(src);
    const int64_t nstrips = (m.ny + m.rps - 1) / m.rps;
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(threads, nstrips)));
    if (!m.lzw) {
        // the file is laid out: bands of whole strips, one per thread (pwrite on a shared descriptor; the layout tables are only read)
        std::vector<char> failed(size_t(nt), 0);
        auto band = [&](int k) {
            const int64_t s0 = nstrips * k / nt, s1 = nstrips * (k + 1) / nt;
            for (int64_t st = s0; st < s1; st++) {
                const int64_t y = st * m.rps, n = std::min<int64_t>(m.rps, m.ny - y);
                if (!pwrite_all(m.fd, s + uint64_t(y) * rowbytes, size_t(uint64_t(n) * rowbytes), m.offsets[size_t(st)])) { failed[size_t(k)] = 1; return; }
            }
        };
        if (nt == 1) band(0);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back(band, k);
            for (auto& t : th) t.join();
        }
        for (char f : failed) if (f) { err_ = "write error"; return false; }
        return true;
    }
    if (m.next_row != 0) { err_ = "LZW output must be written top-down"; return false; }
    // chunks of strips: encoded side by side (the encoder keeps its tables per thread), appended in order - bounded memory for any raster
    const int64_t chunk = int64_t(nt) * 8;
    std::vector<std::vector<unsigned char>> enc(static_cast<size_t>(std::min<int64_t>(chunk, nstrips)));
    for (int64_t c0 = 0; c0 < nstrips; c0 += chunk) {
        const int64_t c1 = std::min<int64_t>(nstrips, c0 + chunk);
        auto work = [&](int k) {
            for (int64_t st = c0 + k; st < c1; st += nt) {
                const int64_t y = st * m.rps, n = std::min<int64_t>(m.rps, m.ny - y);
                lzw_encode(reinterpret_cast<const unsigned char*>(s + uint64_t(y) * rowbytes), size_t(uint64_t(n) * rowbytes), enc[size_t(st - c0)]);
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back(work, k);
            for (auto& t : th) t.join();
        }
        for (int64_t st = c0; st < c1; st++) {
            const std::vector<unsigned char>& e = enc[size_t(st - c0)];
            if (!pwrite_all(m.fd, e.data(), e.size(), m.data_end)) { err_ = "write error"; return false; }
            m.offsets[size_t(st)] = m.data_end; m.counts[size_t(st)] = e.size();
            m.data_end += e.size();
            if (m.data_end & 1) { unsigned char z = 0; pwrite_all(m.fd, &z, 1, m.data_end); m.data_end++; }
        }
    }
    m.next_row = m.ny;
    return true;
}

bool TiffWriter::close() {
    Impl& m = *p_;
    if (m.fd < 0) return true;
    bool ok = true;
    if (m.lzw && !m.update) {
        if (!m.big && m.data_end > 0xFFFF0000ull) { err_ = "LZW output exceeded classic TIFF limits"; ok = false; }
        std::vector<OutTag> tags;
        make_tags(tags, m.big, m.nx, m.ny, m.type, true, m.rps, m.offsets, m.counts, m.nodata, m.have_geo ? &m.geo : nullptr);
        std::vector<unsigned char> ifd = build_ifd(m.big, m.data_end, tags);
        ok = ok && pwrite_all(m.fd, ifd.data(), ifd.size(), m.data_end);
        unsigned char h[16] = {'I', 'I'};
        const uint64_t hdr = m.big ? 16 : 8;
        if (m.big) { uint16_t v = 43, b = 8, z = 0; memcpy(h + 2, &v, 2); memcpy(h + 4, &b, 2); memcpy(h + 6, &z, 2); uint64_t o = m.data_end; memcpy(h + 8, &o, 8); }
        else { uint16_t v = 42; memcpy(h + 2, &v, 2); uint32_t o = uint32_t(m.data_end); memcpy(h + 4, &o, 4); }
        ok = ok && pwrite_all(m.fd, h, size_t(hdr), 0);
        if (!ok && err_.empty()) err_ = "write error";
    }
    ::close(m.fd);
    m.fd = -1;
    return ok;
}

int resolve_output_name(std::string& filename) {
    static const char* ext_list[6] = {".tif", ".img", ".sdat", ".bil", ".bin", ".tiff"};
    size_t dot = filename.rfind('.');
    if (dot == std::string::npos) { filename += ".tif"; return 0; }
    for (size_t i = dot; i < filename.size(); i++) filename[i] = char(tolower((unsigned char)filename[i]));
    std::string ext = filename.substr(dot);
    for (int i = 0; i < 6; i++)
        if (ext == ext_list[i]) return (i == 0 || i == 5) ? 0 : -1;
    filename = filename.substr(0, dot + 1) + "tif";
    return 0;
}

}  // namespace tdx
