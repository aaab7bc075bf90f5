/* test infrastructure: see gdal.h */
#include "gdal.h"
