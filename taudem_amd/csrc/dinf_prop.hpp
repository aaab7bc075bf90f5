// prop() of the D-infinity flow model (src/commonLib.cpp:76-91) on the device, shared by the D-infinity accumulation tools.
// atan2(dy, dx) enters only through the per-row value a2, computed on the host with the host libm; everything else is exactly
// rounded +, -, / in fp64 (the library is built with -ffp-contract=off like the x86-64 reference build).
#pragma once
#include "device_common.hpp"

#define TDX_PI 3.14159265359   /* src/commonLib.h:76 */

struct RowProp { double a2; double dx; };   // a2 = atan2(dyc[j], dxc[j]) from the host libm

// aref[i] of prop() (src/commonLib.cpp:78-79)
__device__ __forceinline__ double aref_at(int i, double a2) {
    switch (i) {
        case 0: return -a2;
        case 1: return 0.;
        case 2: return a2;
        case 3: return (double)(0.5 * TDX_PI);
        case 4: return TDX_PI - a2;
        case 5: return (double)TDX_PI;
        case 6: return TDX_PI + a2;
        case 7: return (double)(1.5 * TDX_PI);
        case 8: return 2. * TDX_PI - a2;
        default: return (double)(2. * TDX_PI);
    }
}

// prop() (src/commonLib.cpp:76-91)
__device__ __forceinline__ double prop_dev(float a, int k, double a2) {
    double p = 0.;
    if (k <= 0) k = k + 8;
    if (k == 1 && a > TDX_PI) a = (float)(a - 2.0 * TDX_PI);
    const double lo = aref_at(k - 1, a2), mid = aref_at(k, a2), hi = aref_at(k + 1, a2);
    if (a > lo && a < hi) {
        if (a > mid) p = (hi - a) / (hi - mid);
        else p = (a - lo) / (mid - lo);
    }
    if (p < 1e-5) return -1.;
    return p;
}

__device__ __forceinline__ int dinf_sector(float ang, double a2) {   // number of aref[1..8] that are <= ang, at least 1
    int sector = 0;
#pragma unroll
    for (int j = 1; j <= 8; j++) sector += (double(ang) >= aref_at(j, a2)) ? 1 : 0;
    return sector < 1 ? 1 : sector;
}

