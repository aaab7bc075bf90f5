// DinfFlowDir on gfx950: replaces the compute part of setdir() (src/dinf.cpp:156-243).
//
//   dinf_slope_kernel   setPosDirDinf + SET2 + VSLOPE (src/dinf.cpp:530-595, 317-373, 286-313): eight
//                       triangular facets per cell in fp64, steepest facet -> angle (f32) and slope (f32)
//   flat resolution     same incfall/incrise breadth-first sweeps as D8 (flats.hpp) with the D-infinity
//                       tests: "has a direction" = angle >= 0 (src/dinf.cpp:678), flat = angle < 0 and not
//                       nodata (src/dinf.cpp:636-641), and the reference's dontCross that compares the
//                       FLOAT angle of the cardinal neighbours with 2,4,6,8 (src/dinf.cpp:58-105, sic)
//   dinf_set2flat_kernel the 4-case SET2 overload on real / artificial elevations (src/dinf.cpp:375-528)
//
// fp64 arithmetic is +,-,*,/,sqrt (exactly rounded; the library is built with -ffp-contract=off like
// the x86-64 reference build) plus atan2.  AD = atan2(D2,D1) depends only on the row's cell size and is
// taken from a host table computed with the host libm; the per-facet atan2(S2,S1) uses the device libm,
// which can differ from glibc in the last ulp of the DOUBLE - visible in the float32 angle only when that
// ulp straddles a float32 rounding boundary.  Observed: none - every digest (2048^2 ... 16384^2) and the
// full-raster comparison at 32768^2 (tests/test_gpu_fullsize.py) match bit for bit; only the small
// golden-raster tests (tests/test_gpu_dinf.py: check_angles) would tolerate single float32 ulps on `ang`.
// Facet choice and slope are compared exactly everywhere.
#include "context.hpp"
#include "device_common.hpp"
#include "flats.hpp"

namespace {
using namespace tdxk;

#define TDX_PI 3.14159265359   /* src/commonLib.h:76 */

// facet tables (src/dinf.cpp:328-335) as arithmetic on packed 2-bit fields: a lookup in a __constant__ array with a
// run-time index is a global load (device_common.hpp, d1/d2)
//   ID1 {1,2,2,1,1,2,2,1}  I1 {0,-1,-1,0,0,1,1,0}  I2 {-1,-1,-1,-1,1,1,1,1}  J1 {1,0,0,-1,-1,0,0,1}  J2 {1,1,-1,-1,-1,-1,1,1}
//   ANGC {0,1,1,2,2,3,3,4} = K / 2      ANGF {1,-1,1,-1,1,-1,1,-1}
__host__ __device__ constexpr int fI1(int K) { return int((0x1a505u >> (2 * K)) & 3u) - 1; }
__host__ __device__ constexpr int fI2(int K) { return int((0x2a801u >> (2 * K)) & 3u) - 1; }
__host__ __device__ constexpr int fJ1(int K) { return int((0x25059u >> (2 * K)) & 3u) - 1; }
__host__ __device__ constexpr int fJ2(int K) { return int((0x28029u >> (2 * K)) & 3u) - 1; }
__host__ __device__ constexpr bool fID1_is1(int K) { return ((0x132u >> K) & 1u) != 0u; }
__host__ __device__ constexpr double fANGC(int K) { return double(K / 2); }
__host__ __device__ constexpr double fANGF(int K) { return (K & 1) ? 1.0 : -1.0; }
__device__ __forceinline__ double fANGC_rt(int K) { return double(K >> 1); }
__device__ __forceinline__ double fANGF_rt(int K) { return (K & 1) ? 1.0 : -1.0; }
static_assert(fI1(3) == -1 && fI2(5) == 1 && fJ1(4) == -1 && fJ2(8) == 1 && fID1_is1(1) && !fID1_is1(2) && fID1_is1(8), "facet tables");

// per-row geometry: DXX[1]=dx, DXX[2]=dy, DD, AD12 = atan2(dy,dx) [D1=dx,D2=dy], AD21 = atan2(dx,dy)
struct RowGeom { double dx, dy, dd, ad12, ad21; };

// VSLOPE (src/dinf.cpp:286-313); AD supplied from the host table
__device__ __forceinline__ void vslope(double E0, double E1, double E2, double D1, double D2, double DD, double AD, double& S, double& A) {
    double S1 = 0, S2 = 0;
    if (D1 != 0) S1 = (E0 - E1) / D1;
    if (D2 != 0) S2 = (E1 - E2) / D2;
    if (S2 == 0 && S1 == 0) A = 0;
    else A = atan2(S2, S1);
    if (A < 0.) { A = 0.; S = S1; }
    else if (A > AD) { A = AD; S = (E0 - E2) / DD; }
    else S = sqrt(S1 * S1 + S2 * S2);
}

__device__ __forceinline__ void facet_geom(const RowGeom& g, int K, double& D1, double& D2, double& AD) {
    const bool one = fID1_is1(K);
    D1 = one ? g.dx : g.dy;
    D2 = one ? g.dy : g.dx;
    AD = one ? g.ad12 : g.ad21;
}

// value number i of four (a run-time index into a register array would go through scratch memory)
template <class T>
__device__ __forceinline__ T sel4(int i, T a0, T a1, T a2, T a3) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }
// the cells of facet K (src/dinf.cpp:328-335): E1 is the cardinal neighbour E N N W W S S E, E2 the diagonal one NE NE NW NW SW SW SE SE
__device__ __forceinline__ int facet_cardinal(int K) { return (K & 7) >> 1; }   // 0 E, 1 N, 2 W, 3 S
__device__ __forceinline__ int facet_diagonal(int K) { return (K - 1) >> 1; }   // 0 NE, 1 NW, 2 SW, 3 SE

// VSLOPE's slope without its atan2.  Which of the three branches a facet takes is a question about the SIGN of S2 and about
// atan2(S2, S1) > atan2(D2, D1), i.e. S2 * D1 > S1 * D2 for positive S1 - decided exactly by that product unless the two sides
// agree to nine digits, where the rounded atan2 values themselves decide as in the reference.  The angle is needed for the
// WINNING facet only: 8 fp64 atan2 per cell become at most one.
//   returns S; *kind = 0: A = 0, 1: A = AD, 2: A = atan2(S2, S1) (interior)
__device__ __forceinline__ double vslope_s(double E0, double E1, double E2, double D1, double D2, double DD, double AD, int* kind, double* s1o, double* s2o) {
    const double S1 = (E0 - E1) / D1, S2 = (E1 - E2) / D2;   // cell sizes are never 0
    *s1o = S1; *s2o = S2;
    int kd;
    if (S2 == 0 && S1 == 0) kd = 2;                       // A = 0 through the else-branch: S = sqrt(0) = 0 (kind 2 with atan2 skipped: A = 0)
    else if (S2 < 0) kd = 0;                              // atan2 < 0
    else if (S2 == 0) kd = S1 > 0 ? 2 : 1;                // atan2(+0, S1): 0 for S1 > 0 (not < 0, not > AD), pi for S1 < 0
    else if (S1 <= 0) kd = 1;                             // angle >= pi/2 > AD
    else {
        const double l = S2 * D1, r = S1 * D2;
        if (fabs(l - r) > 1e-9 * (l + r)) kd = l > r ? 1 : 2;
        // in the band (a gradient along the facet's diagonal edge: tests/test_gpu_dinf.py::test_ramps_along_the_facet_diagonals) two ROUNDED angles decide, as
        // in the reference - both from the same libm there, so both from the device's here (AD, the host's value, is what the angle itself is built from)
        else kd = atan2(S2, S1) > atan2(D2, D1) ? 1 : 2;
    }
    *kind = kd;
    if (kd == 0) return S1;
    if (kd == 1) return (E0 - E2) / DD;
    return sqrt(S1 * S1 + S2 * S2);
}

// setPosDirDinf + SET2 + VSLOPE as a streaming 3x3 stencil: a 256-thread block covers 64 x 64 cells, a lane walks a 16-row
// column segment with the window in registers (3 coalesced row loads per output row).  Facets whose two corners are both not
// lower than the centre cannot exceed SMAX = 0 and are skipped before any division.
// (Round 4: SQ_WAIT_INST_ANY is 45 % of this kernel's wave cycles - its 16 x 8 unrolled facet bodies are 270 KB of code - yet both compact forms, the row
// loop alone rolled up (17 KB, 123 VGPRs) and rows and facets rolled up (7 KB, 119 VGPRs), run SLOWER: 7.3 / 7.2 instead of 6.05 ms at 16384^2; held to
// the unrolled form's 76 VGPRs they spill 41 registers.  Six waves per SIMD hide the fetch stalls better than four waves avoid them.  The list kernel
// dinf_set2flat_kernel is the opposite case and uses the run-time facet loop.)
constexpr int DSLOPE_ROWS = 16;
__global__ __launch_bounds__(256) void dinf_slope_kernel(const float* __restrict__ Z, int nx, int ny, int y_own0, int y_own1, float nodata,
                                                         const RowGeom* __restrict__ geom, float* __restrict__ ANG,
                                                         float* __restrict__ SLP, unsigned long long* __restrict__ nflat) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ybase = y_own0 + blockIdx.y * (4 * DSLOPE_ROWS) + (threadIdx.x >> 6) * DSLOPE_ROWS;
    const bool colok = x < nx;
    const int xc = colok ? x : nx - 1, xm = xc > 0 ? xc - 1 : xc, xp = xc < nx - 1 ? xc + 1 : xc;
    auto ldrow = [&](int y, float& a, float& b, float& c) {
        if (y >= 0 && y < ny) {
            const float* r = Z + size_t(y) * size_t(nx);
            a = r[xm]; b = r[xc]; c = r[xp];
        } else { a = b = c = nodata; }
    };
    // window w[dy + 1][dx + 1]
    float n0, n1, n2, c0, c1, c2, s0, s1, s2;
    ldrow(ybase - 1, n0, n1, n2);
    ldrow(ybase, c0, c1, c2);
    unsigned nfl = 0;
#pragma unroll
    for (int r = 0; r < DSLOPE_ROWS; r++) {
        const int y = ybase + r;
        ldrow(y + 1, s0, s1, s2);
        if (colok && y < y_own1) {
            const size_t idx = size_t(y) * size_t(nx) + size_t(x);
            float ang = TDX_ANG_NODATA, slp = -1.0f;
            const float z0 = c1;
            const bool edge = (x == 0 || y == 0 || x == nx - 1 || y == ny - 1);
            if (!edge && !is_nodata_f(z0, nodata)) {
                const bool con = is_nodata_f(n0, nodata) || is_nodata_f(n1, nodata) || is_nodata_f(n2, nodata) || is_nodata_f(c0, nodata) ||
                                 is_nodata_f(c2, nodata) || is_nodata_f(s0, nodata) || is_nodata_f(s1, nodata) || is_nodata_f(s2, nodata);
                if (!con) {
                    const RowGeom g = geom[y];
                    const float w[3][3] = {{n0, n1, n2}, {c0, c1, c2}, {s0, s1, s2}};
                    double SMAX = 0., S1W = 0., S2W = 0.;
                    int KD = 0, KINDW = 0;
                    const double a = (double)z0;
#pragma unroll
                    for (int K = 1; K <= 8; K++) {
                        // SET2(I=row, J=col): E1 at (row+I1, col+J1), E2 at (row+I2, col+J2)
                        const float e1 = w[1 + fI1(K)][1 + fJ1(K)], e2 = w[1 + fI2(K)][1 + fJ2(K)];
                        if (!(e1 < z0 || e2 < z0)) continue;   // S1 <= 0 and (E0 - E2) <= 0: every branch of VSLOPE gives S <= 0
                        double D1, D2, AD, sa, sb;
                        facet_geom(g, K, D1, D2, AD);
                        int kind;
                        const double S = vslope_s(a, (double)e1, (double)e2, D1, D2, g.dd, AD, &kind, &sa, &sb);
                        if (S > SMAX) { SMAX = S; KD = K; KINDW = kind; S1W = sa; S2W = sb; }
                    }
                    ang = -1.f;
                    if (KD > 0) {
                        double AD, D1, D2;
                        facet_geom(g, KD, D1, D2, AD);
                        const double AMAX = KINDW == 0 ? 0. : (KINDW == 1 ? AD : ((S2W == 0 && S1W == 0) ? 0. : atan2(S2W, S1W)));
                        ang = (float)(fANGC(KD) * (TDX_PI / 2) + fANGF(KD) * AMAX);
                    }
                    slp = (float)SMAX;
                    if (ang == -1.f) nfl++;
                }
            }
            ANG[idx] = ang;
            SLP[idx] = slp;
        }
        n0 = c0; n1 = c1; n2 = c2;
        c0 = s0; c1 = s1; c2 = s2;
    }
    (void)block_reserve(nfl, nflat);   // one atomic per block
}

// (Round 5 also built the slope pass as TWO kernels - an fp32 candidate mask, then fp64 on the candidates: bit-identical, slower (8.4 + 15.3 against 22.3 ms at
// 32768^2: the candidate pass is itself ~260 VALU instructions per cell); docs/experiments_r05.md section 2b.  Retired in round 6.)

struct DinfTraits {
    const float* ANG;
    // dinf.cpp:58-105: float angle of the cardinal neighbours compared with the D8 codes 2,4,6,8
    __device__ __forceinline__ bool dont_cross(size_t c, int nx, int k) const {
        switch (k) {
            case 2: return ANG[c + 1] == 4 || ANG[c - nx] == 8;
            case 4: return ANG[c - nx] == 6 || ANG[c - 1] == 2;
            case 6: return ANG[c + nx] == 4 || ANG[c - 1] == 8;
            case 8: return ANG[c + 1] == 6 || ANG[c + nx] == 2;
            default: return false;
        }
    }
    __device__ __forceinline__ bool has_direction(size_t n) const { return ANG[n] >= 0.0f; }
};

__device__ __forceinline__ bool dinf_is_flat(float a) { return !is_nodata_f(a, TDX_ANG_NODATA) && a < 0.0f; }

// The angle raster as the one-hot codes of flatk::classify_stream_kernel: bit 0 = in the queue (a flat cell: angle -1), bits 1 .. 8 = has a direction
// (angle >= 0), and - the reference's dontCross compares the FLOAT angle of the cardinal neighbours with the D8 codes 2, 4, 6, 8 (src/dinf.cpp:58-105,
// DinfTraits::dont_cross) - an angle that IS 2, 4, 6 or 8 sets that code's bit, any other angle bit 1; nothing for nodata.
struct DinfCodes {
    using Raw = float;
    const float* ANG;
    __device__ __forceinline__ float load(size_t o) const { return ANG[o]; }
    static __device__ __forceinline__ unsigned onehot(float a) {
        if (is_nodata_f(a, TDX_ANG_NODATA)) return 0u;
        if (a < 0.0f) return 1u;
        return a == 2.0f ? 1u << 2 : (a == 4.0f ? 1u << 4 : (a == 6.0f ? 1u << 6 : (a == 8.0f ? 1u << 8 : 1u << 1)));
    }
};

// flat queue + markers (8 cells per lane, one atomic per block)
template <class LV>
__global__ __launch_bounds__(256) void dinf_collect_flats_kernel(const float* __restrict__ ANG, size_t first, size_t n, LV* __restrict__ lvl,
                                                                 LV* __restrict__ rq, uint32_t* __restrict__ list,
                                                                 unsigned long long* __restrict__ counter) {
    // a thread takes EIGHT CONSECUTIVE cells (and block_reserve hands out slots in thread order), so the list is in raster order: the list kernels of
    // flat resolution gather the 3 x 3 windows of 64 consecutive entries per wave - with a thread's cells 256 apart, as they were, consecutive entries
    // alternated between eight distant places
    const size_t c0 = first + (size_t(blockIdx.x) * 256 + threadIdx.x) * 8;   // cells [first, n): the owned rows
    unsigned mask = 0;
    float a[8];
    const bool whole = c0 + 8 <= n && (reinterpret_cast<uintptr_t>(ANG + c0) & 15u) == 0;   // (16-byte loads: the caller's raster may start anywhere)
    if (whole) {
        const float4 a0 = *reinterpret_cast<const float4*>(ANG + c0), a1 = *reinterpret_cast<const float4*>(ANG + c0 + 4);
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = c0 + i < n ? ANG[c0 + i] : TDX_ANG_NODATA;
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (c0 + i < n && dinf_is_flat(a[i])) mask |= 1u << i;
    if (sizeof(LV) == 2 && c0 + 8 <= n && ((reinterpret_cast<uintptr_t>(lvl + c0) | reinterpret_cast<uintptr_t>(rq + c0)) & 15u) == 0) {   // eight int16 markers = one 16-byte store
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = (((mask >> (2 * j)) & 1u) ? 0u : 0xFFFFu) | (((mask >> (2 * j + 1)) & 1u) ? 0u : 0xFFFF0000u);   // 0 in the queue, -1 outside
        *reinterpret_cast<uint4*>(lvl + c0) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(rq + c0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (c0 + i < n) { const LV m = ((mask >> i) & 1u) ? LV(0) : LV(-1); lvl[c0 + i] = m; rq[c0 + i] = m; }
    }
    unsigned long long pos = block_reserve(unsigned(__popc(mask)), counter);
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (mask & (1u << i)) list[pos++] = uint32_t(c0 + size_t(i));
}

template <class LV>
__global__ __launch_bounds__(256) void dinf_mark_pits_kernel(const uint32_t* __restrict__ list, unsigned long long nq,
                                                             const LV* __restrict__ lvl, float* __restrict__ ANG) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const size_t c = list[q];
    if (lvl[c] == 0) ANG[c] = TDX_ANG_NODATA;   // src/dinf.cpp:723
}

// SET2 overload (src/dinf.cpp:375-528) for every cell of the flat list.
// The reference walks the eight facets with an early exit and reads what each facet needs as it goes - on the GPU that was up to two
// dozen DEPENDENT memory round trips per cell (marker -> elevation or level, facet after facet; 10 ms for 82 M flat cells at 16384^2).
// Here the whole 3 x 3 neighbourhood of the three arrays is requested up front (27 loads in flight together: one latency), the facet
// loop is unrolled over registers with a `done` flag in place of the break, and VSLOPE's atan2 is evaluated once, for the facet that
// wins (vslope_s decides its three branches without the angle, as in dinf_slope_kernel).
template <class LV>
__global__ __launch_bounds__(256) void dinf_set2flat_kernel(const float* __restrict__ Z, int nx, const RowGeom* __restrict__ geom,
                                                            const uint32_t* __restrict__ list, unsigned long long nq,
                                                            const LV* __restrict__ lvl, const LV* __restrict__ rq,
                                                            FlatLevels fl, float* __restrict__ ANG) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const size_t c0 = list[q];
    const int y = int(c0 / size_t(nx));
    // window w[(di + 1) * 3 + (dj + 1)]: a flat cell is an interior cell, all nine cells exist
    float zw[9];
    LV lw[9], rw[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const size_t n = size_t(ptrdiff_t(c0) + ptrdiff_t(i / 3 - 1) * nx + (i % 3 - 1));
        zw[i] = Z[n]; lw[i] = lvl[n]; rw[i] = rq[n];
    }
    const float ang0 = ANG[c0];
    const RowGeom g = geom[y];
    int e2w[9];   // elev2 + s of the nine cells (src/d8.cpp:545,640-645; only read where the cell is a marked flat cell)
#pragma unroll
    for (int i = 0; i < 9; i++) e2w[i] = int(flat_elev2<LV>(lw[i], rw[i], fl));
    double SMAX = 0.0;
    int KD = 0, KINDW = 0;          // winning facet; how its angle follows: 0: A = 0, 1: A = AD, 2: A = atan2(S2W, S1W)
    double S1W = 0., S2W = 0.;
    bool diagOutFound = false, done = false;
    const double a = (double)zw[4];
    const int a1 = e2w[4];
    // One facet body with run-time K (not eight unrolled ones with four inlined VSLOPEs each: 70 KB of code, 63 % of the wave cycles waiting for
    // instruction fetch): the four cases of src/dinf.cpp:399-500 differ in WHICH three numbers go into VSLOPE and in how its result is taken.
#pragma unroll 1
    for (int K = 1; K <= 8 && !done; K++) {
        const int j1 = facet_cardinal(K), j2 = facet_diagonal(K);
        const float zb = sel4(j1, zw[5], zw[1], zw[3], zw[7]), zc = sel4(j2, zw[2], zw[0], zw[6], zw[8]);
        const bool in1 = sel4(j1, rw[5], rw[1], rw[3], rw[7]) > 0, in2 = sel4(j2, rw[2], rw[0], rw[6], rw[8]) > 0;   // dn > 0
        const int l1 = sel4(j1, e2w[5], e2w[1], e2w[3], e2w[7]), l2 = sel4(j2, e2w[2], e2w[0], e2w[6], e2w[8]);
        const double b = (double)zb, cc = (double)zc;
        if (!in1 && in2 && a >= b) { KD = K; KINDW = 0; done = true; continue; }                         // ANGLE[K] = 0
        if (in1 && !in2 && a >= cc) { if (!diagOutFound) { KD = K; KINDW = 1; diagOutFound = true; } continue; }   // ANGLE[K] = atan2(DXX[ID2], DXX[ID1]) = AD
        const bool real = !in1 && !in2;          // both corners outside the marked flat: the real elevations; else the artificial surface
        const double E0 = real ? a : (double)a1;
        const double E1 = real ? b : (double)(in1 ? l1 : (a1 > l2 ? a1 : l2));
        const double E2 = real ? cc : (double)(in2 ? l2 : (a1 > l1 ? a1 : l1));
        double D1, D2, AD, sa, sb;
        facet_geom(g, K, D1, D2, AD);
        int kind;
        const double S = vslope_s(E0, E1, E2, D1, D2, g.dd, AD, &kind, &sa, &sb);
        if (real) {
            if (S >= 0.0) {
                if (b > a) { if (!diagOutFound) { diagOutFound = true; KD = K; KINDW = kind; S1W = sa; S2W = sb; } }
                else { KD = K; KINDW = kind; S1W = sa; S2W = sb; done = true; }
            }
        } else if (S > SMAX) { SMAX = S; KD = K; KINDW = kind; S1W = sa; S2W = sb; }
    }
    float ang = ang0;
    if (!is_nodata_f(ang, TDX_ANG_NODATA)) ang = -1.f;
    if (KD > 0) {
        double D1, D2, AD;
        facet_geom(g, KD, D1, D2, AD);
        const double AKD = KINDW == 0 ? 0. : (KINDW == 1 ? AD : ((S2W == 0 && S1W == 0) ? 0. : atan2(S2W, S1W)));
        const float t = (float)(fANGC(KD) * (TDX_PI / 2) + fANGF(KD) * AKD);
        if (t >= 0.0f) ang = t;
    }
    ANG[c0] = ang;
}

__global__ __launch_bounds__(256) void dinf_recollect_kernel(const float* __restrict__ ANG, const uint32_t* __restrict__ list,
                                                             unsigned long long nq, uint32_t* __restrict__ out,
                                                             unsigned long long* __restrict__ counter) {
    const unsigned long long q0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 8;   // eight consecutive entries per thread: the list keeps its (raster) order
    uint32_t keep[8];
    unsigned cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned long long q = q0 + (unsigned long long)i;
        if (q < nq) {
            const uint32_t c = list[q];
            if (dinf_is_flat(ANG[c])) keep[cnt++] = c;
        }
    }
    unsigned long long pos = block_reserve(cnt, counter);
    for (unsigned i = 0; i < cnt; i++) out[pos + i] = keep[i];
}

}  // namespace

// One strip of setdir() (src/dinf.cpp:156-243); the same strip protocol as d8flowdir_impl (d8flowdir.hip).
template <class LV>
static int dinfflowdir_levels(tdx_context* ctx, const Strip& st, float* d_fel, float fel_nodata, const double* dxc, const double* dyc, float* d_ang,
                            float* d_slp, tdx_stats* stats) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = st.nx, iny = st.ny_arr;
    const size_t n = size_t(inx) * size_t(iny);
    // per-row geometry on the host (sqrt and atan2 from the host libm, src/dinf.cpp:575-576, 300, 466)
    std::vector<RowGeom> geom;
    geom.resize(size_t(iny));
    for (int j = 0; j < iny; j++) {
        RowGeom g;
        g.dx = dxc[j]; g.dy = dyc[j];
        g.dd = sqrt(dxc[j] * dxc[j] + dyc[j] * dyc[j]);
        g.ad12 = atan2(dyc[j], dxc[j]);
        g.ad21 = atan2(dxc[j], dyc[j]);
        geom[size_t(j)] = g;
    }
    RowGeom* d_geom = static_cast<RowGeom*>(ctx->scratch(TDX_S_J, geom.size() * sizeof(RowGeom)));
    if (!d_geom) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_geom, geom.data(), geom.size() * sizeof(RowGeom), hipMemcpyHostToDevice, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);

    ctx->begin_call(stats);
    strip_mark(ctx, st, "dinfflowdir");
    ctx->phase = "slope pass";
    int rc = strip_exchange<float>(ctx, st, d_fel, fel_nodata);   // elevation halo rows
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        dim3 grid((inx + 63) / 64, (st.y1 - st.y0 + 4 * DSLOPE_ROWS - 1) / (4 * DSLOPE_ROWS));
        hipLaunchKernelGGL(dinf_slope_kernel, grid, dim3(256), 0, s, d_fel, inx, iny, st.y0, st.y1, fel_nodata, d_geom, d_ang, d_slp, d_cnt);   // all eight facets in fp64 in one kernel
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
    unsigned long long nq = ctx->h_mail[0];   // flats of this strip
    int64_t total = int64_t(nq);               // flats of the whole raster
    rc = strip_allreduce(ctx, st, &total, 1, TDX_OP_SUM);
    if (rc != TDX_OK) return rc;
    if (stats) { stats->flats_initial = total; stats->flats_left = total; }

    if (total > 0) {
        LV* lvl = static_cast<LV*>(ctx->scratch(TDX_S_A, n * sizeof(LV)));
        LV* rq = static_cast<LV*>(ctx->scratch(TDX_S_B, n * sizeof(LV)));
        uint32_t* qlist = static_cast<uint32_t*>(ctx->scratch(TDX_S_C, size_t(nq) * 4));
        uint32_t* qnext = static_cast<uint32_t*>(ctx->scratch(TDX_S_D, size_t(nq) * 4));
        if (!lvl || !rq || !qlist || !qnext) return TDX_ERR_NOMEM;
        float* zwork = nullptr;
        const float* zcur = d_fel;
        FlatBuffersT<LV> fbuf{lvl, rq};
        rc = strip_exchange<float>(ctx, st, d_ang, TDX_ANG_NODATA);   // angles of the neighbours' boundary rows
        if (rc != TDX_OK) return rc;
        TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
        const size_t own_first = size_t(st.y0) * size_t(inx), own_end = size_t(st.y1) * size_t(inx);
        hipLaunchKernelGGL((dinf_collect_flats_kernel<LV>), dim3(tdx_blocks_for(own_end - own_first, 2048)), dim3(256), 0, s, d_ang, own_first, own_end, lvl, rq,
                           qlist, d_cnt);
        rc = strip_exchange<LV>(ctx, st, lvl, LV(-1));   // queue membership of the neighbours' boundary rows
        if (rc != TDX_OK) return rc;
        rc = strip_exchange<LV>(ctx, st, rq, LV(-1));
        if (rc != TDX_OK) return rc;
        int64_t last = total;
        bool first = true;
        unsigned long long nq_old = 0;      // cells of the previous iteration's queue (in qnext after the swap)
        for (int iteration = 1;; iteration++) {
            const FlatPhases& ph = flat_phases(iteration);
            ctx->phase = ph.classify;
            if (!first) { rc = flats_reset_markers_after(ctx, st, qnext, nq_old, qlist, nq, lvl, rq); if (rc != TDX_OK) return rc; }
            first = false;
            FlatLevels fl;
            DinfTraits tr{d_ang};
            // A dense queue (the first iteration: a third of the raster at BASELINE.json configs[2]) is classified by the streaming pass D8FlowDir uses - markers and
            // masks of every owned cell from the 3 x 3 windows of the elevations and the angles - which also marks the tiles that are not full, so that the level
            // fields' open water runs as blocks (flats.hpp); later iterations work from the list.  The choice is this rank's own: both forms write the same values.
            const float* zc = zcur;
            const StreamClassifyFn classify = [&](const tilek::TileGeom& g, uint8_t* fmask, uint8_t* rmask, uint32_t* tile_flags, uint8_t* tile_masked, uint8_t* notfull) {
                const int nbx = (st.nx + flatk::CLS_COLS - 1) / flatk::CLS_COLS;
                const dim3 grid(tdx_xcd_grid_x(unsigned(nbx)), (st.y1 - st.y0 + 4 * flatk::CLS_ROWS - 1) / (4 * flatk::CLS_ROWS));
                hipLaunchKernelGGL((flatk::classify_stream_kernel<LV, DinfCodes>), grid, dim3(256), 0, s, zc, DinfCodes{d_ang}, st.nx, st.ny_arr, st.y0, st.y1, g.tiles_x, lvl, rq,
                                   fmask, rmask, tile_flags, tile_masked, nbx, tdx_xcd_map() ? 1 : 0, notfull);
            };
            const bool force_list = getenv("TDX_FLATS_LIST") != nullptr;   // (test hook, read per call: the list classification for a dense queue too)
            const bool dense = !force_list && nq > n / 16;
            rc = flats_bfs<DinfTraits, LV>(ctx, tr, zcur, st, qlist, nq, fbuf, &fl, stats, dense ? &classify : nullptr, iteration);
            if (rc != TDX_OK) return rc;
            ctx->phase = ph.directions;
            {
                TdxSpan sp(ctx, TDX_K_FLATDIR);
                // (the queue-length counter of the next iteration is word 4 of the stage counters: cleared by flatk::prepare_kernel of THIS iteration, untouched since)
                unsigned long long* d_next = d_cnt + 4;
                if (nq) {
                    if (fl.has_pits)
                        hipLaunchKernelGGL((dinf_mark_pits_kernel<LV>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, s, qlist, nq, lvl, d_ang);
                    hipLaunchKernelGGL((dinf_set2flat_kernel<LV>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, s, zcur, inx, d_geom, qlist, nq, lvl, rq, fl, d_ang);
                    hipLaunchKernelGGL(dinf_recollect_kernel, dim3(tdx_blocks_for(nq, 2048)), dim3(256), 0, s, d_ang, qlist, nq, qnext, d_next);
                }
                TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_next, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
                if (stats) stats->launches[TDX_K_FLATDIR] += 2 + (fl.has_pits ? 1 : 0);
            }
            const unsigned long long nleft = ctx->h_mail[0];
            total = int64_t(nleft);
            rc = strip_allreduce(ctx, st, &total, 1, TDX_OP_SUM);
            if (rc != TDX_OK) return rc;
            rc = strip_exchange<float>(ctx, st, d_ang, TDX_ANG_NODATA);   // the neighbours' new angles
            if (rc != TDX_OK) return rc;
            if (stats) { stats->flat_iterations++; stats->flats_left = total; }
            if (!(total > 0 && total < last)) break;     // src/dinf.cpp:230
            ctx->phase = ph.next;
            if (!zwork) { zwork = static_cast<float*>(ctx->scratch(TDX_S_I, n * 4)); if (!zwork) return TDX_ERR_NOMEM; }
            // src/dinf.cpp:822-828 (only where the next iteration reads it when few flats are left)
            rc = nleft <= n / 32 ? flats_overwrite_elevation_sparse(ctx, inx, qnext, nleft, lvl, rq, fl, zwork) : flats_overwrite_elevation(ctx, n, lvl, rq, fl, zwork);
            if (rc != TDX_OK) return rc;
            zcur = zwork;
            nq_old = nq;
            std::swap(qlist, qnext);
            nq = nleft; last = total;
        }
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    ctx->end_call();
    return TDX_OK;
}

// The level fields are int16 like the reference's elev2 / dn / s partitions (src/dinf.cpp:614-617); a flat deeper than they hold (32 766 levels - the
// reference's short counters wrap there, so nothing there is defined to be equal to) starts the call over on int32 fields.  TDX_LEVELS_INT32=1 (test
// hook, read per call): int32 fields from the start.
static int dinfflowdir_impl(tdx_context* ctx, const Strip& st, float* d_fel, float fel_nodata, const double* dxc, const double* dyc, float* d_ang,
                            float* d_slp, tdx_stats* stats) {
    int rc = getenv("TDX_LEVELS_INT32") ? TDX_FLATS_TOO_DEEP : dinfflowdir_levels<int16_t>(ctx, st, d_fel, fel_nodata, dxc, dyc, d_ang, d_slp, stats);
    if (rc == TDX_FLATS_TOO_DEEP) rc = dinfflowdir_levels<int32_t>(ctx, st, d_fel, fel_nodata, dxc, dyc, d_ang, d_slp, stats);
    return rc;
}

extern "C" int tdx_dinfflowdir_dev(tdx_context* ctx, const float* d_fel, int64_t nx, int64_t ny, float fel_nodata,
                                   const double* dxc, const double* dyc, float* d_ang, float* d_slp, tdx_stats* stats) {
    if (!ctx || !d_fel || !d_ang || !d_slp || !dxc || !dyc || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfflowdir_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return dinfflowdir_impl(ctx, strip_single(int(nx), int(ny)), const_cast<float*>(d_fel), fel_nodata, dxc, dyc, d_ang, d_slp, stats);
}

extern "C" int tdx_dinfflowdir_strip(tdx_context* ctx, const tdx_comm* comm, float* d_fel, int64_t nx, int64_t ny_local, float fel_nodata,
                                     const double* dxc, const double* dyc, float* d_ang, float* d_slp, tdx_stats* stats) {
    if (!ctx || !d_fel || !d_ang || !d_slp || !dxc || !dyc || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfflowdir_strip: bad argument");
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return dinfflowdir_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_fel, fel_nodata, dxc, dyc, d_ang, d_slp, stats);
}

extern "C" int tdx_dinfflowdir(tdx_context* ctx, const float* fel, int64_t nx, int64_t ny, float fel_nodata,
                               const double* dxc, const double* dyc, float* ang, float* slp, tdx_stats* stats) {
    if (!ctx || !fel || !ang || !slp || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfflowdir: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_z = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_s = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    if (!d_z || !d_a || !d_s) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_z, fel, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = tdx_dinfflowdir_dev(ctx, d_z, nx, ny, fel_nodata, dxc, dyc, d_a, d_s, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ang, d_a, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(slp, d_s, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
