// Outlet point reader: replaces readoutlets() (src/ReadOutlets.cpp:49-198), which uses OGR.
// Supported sources: ESRI shapefile point layers (.shp: Point / PointZ / PointM), GeoJSON Point
// features (.json/.geojson) and plain text ("x y [id]" per line, '#' comments, ',' or blanks).
// Like the reference, only x,y (and a best-effort id) are returned; ids default to index+1
// (src/ReadOutlets.cpp:176-181).
#pragma once
#include <string>
#include <vector>

namespace tdx {
bool read_outlets(const std::string& path, std::vector<double>& x, std::vector<double>& y,
                  std::vector<int>& id, std::string& err);
// tiffIO::geoToGlobalXY (src/tiffIO.cpp:580-588): (int) truncation of (x-xleft)/dlon, (ytop-y)/dlat
void geo_to_global_xy(double geoX, double geoY, double xleftedge, double ytopedge, double dlon, double dlat,
                      int& gx, int& gy);
}  // namespace tdx
