// prop() of the D-infinity flow model (src/commonLib.cpp:76-91) on the device, shared by the D-infinity accumulation tools.
// atan2(dy, dx) enters only through the per-row value a2, computed on the host with the host libm; everything else is exactly
// rounded +, -, / in fp64 (the library is built with -ffp-contract=off like the x86-64 reference build).
#pragma once
#include "device_common.hpp"

#define TDX_PI 3.14159265359   /* src/commonLib.h:76 */

struct RowProp { double a2; double dx; };   // a2 = atan2(dyc[j], dxc[j]) from the host libm

// aref[i] of prop() (src/commonLib.cpp:78-79): { -a2, 0, a2, pi/2, pi - a2, pi, pi + a2, 3pi/2, 2pi - a2, 2pi }, i = 0 .. 9.
// Without a branch: as a switch this became a tree of divergent branches (the index differs from lane to lane), ~15 exec-mask blocks per
// call and 24 calls per lane in the set-up of a reverse-sweep activation - 5.5 us of its ~25 (round 6).  Even i: a constant base plus or
// minus a2 (x - a2 and x + (-a2) are the same operation; -a2 itself for i = 0, a2 for i = 2), odd i: one of five constants.
__device__ __forceinline__ double aref_at(int i, double a2) {
    const int h = i >> 1;
    const double t = (h & 1) ? a2 : -a2;
    const double base = h < 4 ? (double)TDX_PI : (double)(2. * TDX_PI);
    const double even = h < 2 ? t : base + t;
    const double odd = h == 0 ? 0. : (h == 1 ? (double)(0.5 * TDX_PI) : (h == 2 ? (double)TDX_PI : (h == 3 ? (double)(1.5 * TDX_PI) : (double)(2. * TDX_PI))));
    return (i & 1) ? odd : even;
}

// prop() (src/commonLib.cpp:76-91)
__device__ __forceinline__ double prop_dev(float a, int k, double a2) {
    double p = 0.;
    if (k <= 0) k = k + 8;
    if (k == 1 && a > TDX_PI) a = (float)(a - 2.0 * TDX_PI);
    const double lo = aref_at(k - 1, a2), mid = aref_at(k, a2), hi = aref_at(k + 1, a2);
    if (a > lo && a < hi) {   // (one division: numerator and denominator selected first - lanes on either side of `mid` share it)
        const bool upper = a > mid;
        const double num = upper ? hi - a : a - lo, den = upper ? hi - mid : mid - lo;
        p = num / den;
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


// ---- the outflow of a cell as one byte (pass 1 of the setup stencils of the D-infinity sweeps) ------------------------------
// prop() is positive only for the two directions that bracket the angle (src/commonLib.cpp:83-88): s1 = dinf_sector(angle) and
// s1 % 8 + 1.  Whether neighbour k drains into a cell - prop(angle_n, (k + 4) % 8) > 0, evaluated eight times per cell by
// initNeighborDinfup (src/commonLib.cpp:99-131) - is therefore a property of the NEIGHBOUR's own two proportions: pass 1 computes
// them once per cell (two fp64 divisions instead of ten) and leaves this byte, pass 2 is a byte stencil.
//   0xFF: no angle (nodata)      else [0:3) s1 - 1, [3] prop(s1) > 0, [4] prop(s1 % 8 + 1) > 0, [5] the cell participates,
//   [6] the cell has no angle in the ORIGINAL raster: an outlet placed on such a cell takes part as a pure sink (TDX_ANG_SINK, outlets
//       mode), but its neighbours still see a cell without angle - the reference's contamination tests read the original angles
//       (src/areadinf.cpp:196-199, src/DinfConcLimAccum.cpp:243, src/DinfTransLimAccum.cpp:246)
constexpr unsigned DINF_CODE_NODATA = 0xFFu, DINF_CODE_P1 = 8u, DINF_CODE_P2 = 16u, DINF_CODE_PART = 32u, DINF_CODE_NOANGLE = 64u;
constexpr float TDX_ANG_OUTSIDE = 100.0f, TDX_ANG_SINK = 200.0f;   // re-coded angles of outlets mode (dinf_outlets.hpp)
// the neighbour with code `c` counts as "missing" for the contamination test of a cell beside it
__device__ __forceinline__ bool dinf_code_missing(unsigned c) { return c == DINF_CODE_NODATA || (c & DINF_CODE_NOANGLE) != 0u; }
__device__ __forceinline__ unsigned dinf_code(float ang, bool nodata, bool participates, double a2, double* p1, double* p2) {
    *p1 = 0.; *p2 = 0.;
    if (nodata) return DINF_CODE_NODATA;
    unsigned c = ang == TDX_ANG_SINK ? DINF_CODE_NOANGLE : 0u;
    if (participates) {
        c |= DINF_CODE_PART;
        const int s1 = dinf_sector(ang, a2);
        c |= unsigned(s1 - 1);
        const double q1 = prop_dev(ang, s1, a2), q2 = prop_dev(ang, s1 % 8 + 1, a2);
        if (q1 > 0.0) { c |= DINF_CODE_P1; *p1 = q1; }
        if (q2 > 0.0) { c |= DINF_CODE_P2; *p2 = q2; }
    }
    return c;
}
// does the cell with code `c` send flow in direction kk (1..8)?  0: no, 1: with its first proportion, 2: with its second
__device__ __forceinline__ int dinf_code_sends(unsigned c, int kk) {
    if (c == DINF_CODE_NODATA) return 0;
    const int s1 = int(c & 7u) + 1, s2 = s1 % 8 + 1;
    if (s1 == kk && (c & DINF_CODE_P1)) return 1;
    if (s2 == kk && (c & DINF_CODE_P2)) return 2;
    return 0;
}

// pass 1 alone (no proportions kept): `inert` = an angle value whose cells have a valid angle but neither send nor participate
// (TDX_ANG_OUTSIDE of outlets mode; any value no raster holds otherwise)
static __global__ __launch_bounds__(256) void dinf_code_kernel(const float* __restrict__ ANG, size_t n, int nx, float nodata, float inert,
                                                               const double* __restrict__ a2row, uint8_t* __restrict__ code) {
    const size_t idx = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n) return;
    const float ang = ANG[idx];
    const bool nd = tdxk::is_nodata_f(ang, nodata);
    double p1, p2;
    code[idx] = uint8_t(dinf_code(ang, nd, !(nd || ang == inert), a2row[idx / size_t(nx)], &p1, &p2));
}
// the 8 neighbour codes of (x, y): loads first (clamped), cells outside the raster read as "no angle"
__device__ __forceinline__ void dinf_code_window(const uint8_t* __restrict__ code, int nx, int ny, int x, int y, unsigned (&c)[9]) {
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        const int xn = x + tdxk::d1(k), yn = y + tdxk::d2(k);
        const bool in = xn >= 0 && xn < nx && yn >= 0 && yn < ny;
        c[k] = code[size_t(in ? yn : y) * size_t(nx) + size_t(in ? xn : x)];
        if (!in) c[k] = DINF_CODE_NODATA;
    }
    c[0] = code[size_t(y) * size_t(nx) + size_t(x)];
}
