#include "outlets.hpp"

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace tdx {
namespace {
std::string lower_ext(const std::string& p) {
    size_t d = p.rfind('.');
    if (d == std::string::npos) return "";
    std::string e = p.substr(d);
    for (char& c : e) c = char(tolower((unsigned char)c));
    return e;
}
uint32_t be32(const unsigned char* q) { return (uint32_t(q[0]) << 24) | (uint32_t(q[1]) << 16) | (uint32_t(q[2]) << 8) | q[3]; }
int32_t le32(const unsigned char* q) { uint32_t v = uint32_t(q[0]) | (uint32_t(q[1]) << 8) | (uint32_t(q[2]) << 16) | (uint32_t(q[3]) << 24); return int32_t(v); }
double le64f(const unsigned char* q) { double v; memcpy(&v, q, 8); return v; }   // host is little-endian

bool read_shp(const std::string& path, std::vector<double>& x, std::vector<double>& y, std::vector<int>& id, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.size() < 100 || be32(buf.data()) != 9994) { err = "not a shapefile: " + path; return false; }
    int shptype = le32(buf.data() + 32);
    if (!(shptype == 1 || shptype == 11 || shptype == 21)) { err = "shapefile layer is not a point layer"; return false; }
    size_t pos = 100;
    while (pos + 8 <= buf.size()) {
        uint32_t recno = be32(buf.data() + pos);
        size_t clen = size_t(be32(buf.data() + pos + 4)) * 2;
        pos += 8;
        if (pos + clen > buf.size()) break;
        if (clen >= 20) {
            int t = le32(buf.data() + pos);
            if (t == 1 || t == 11 || t == 21) {
                x.push_back(le64f(buf.data() + pos + 4));
                y.push_back(le64f(buf.data() + pos + 12));
                id.push_back(int(recno));
            }
        }
        pos += clen;
    }
    return true;
}

// Minimal GeoJSON scan: every "coordinates": [x, y ...] whose enclosing geometry type is Point.
bool read_geojson(const std::string& path, std::vector<double>& x, std::vector<double>& y, std::vector<int>& id, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "cannot open " + path; return false; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string s = ss.str();
    size_t pos = 0;
    while ((pos = s.find("\"coordinates\"", pos)) != std::string::npos) {
        size_t b = s.find('[', pos);
        if (b == std::string::npos) break;
        size_t q = b + 1;
        while (q < s.size() && isspace((unsigned char)s[q])) q++;
        if (q < s.size() && s[q] != '[') {   // a flat [x, y] pair => Point
            char* e1 = nullptr;
            double vx = strtod(s.c_str() + q, &e1);
            const char* c = e1;
            while (*c && (isspace((unsigned char)*c) || *c == ',')) c++;
            char* e2 = nullptr;
            double vy = strtod(c, &e2);
            if (e1 != s.c_str() + q && e2 != c) { x.push_back(vx); y.push_back(vy); id.push_back(int(x.size())); }
        }
        pos = b + 1;
    }
    return true;
}

bool read_text(const std::string& path, std::vector<double>& x, std::vector<double>& y, std::vector<int>& id, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "cannot open " + path; return false; }
    std::string line;
    while (std::getline(f, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line.resize(h);
        for (char& c : line) if (c == ',' || c == ';' || c == '\t') c = ' ';
        std::istringstream is(line);
        double vx, vy;
        if (!(is >> vx >> vy)) continue;
        double vid;
        int iid = int(x.size()) + 1;
        if (is >> vid) iid = int(vid);
        x.push_back(vx); y.push_back(vy); id.push_back(iid);
    }
    return true;
}
}  // namespace

bool read_outlets(const std::string& path, std::vector<double>& x, std::vector<double>& y, std::vector<int>& id, std::string& err) {
    x.clear(); y.clear(); id.clear();
    const std::string e = lower_ext(path);
    if (e == ".shp") return read_shp(path, x, y, id, err);
    if (e == ".json" || e == ".geojson") return read_geojson(path, x, y, id, err);
    return read_text(path, x, y, id, err);
}

void geo_to_global_xy(double geoX, double geoY, double xleftedge, double ytopedge, double dlon, double dlat, int& gx, int& gy) {
    gx = (int)((geoX - xleftedge) / dlon);
    gy = (int)((ytopedge - geoY) / dlat);
}
}  // namespace tdx
