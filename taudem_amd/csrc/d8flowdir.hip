// D8FlowDir on gfx950: replaces the compute part of setdird8() (src/d8.cpp:227-320).
//
//   d8_slope_kernel      setPosDir + setFlow + calcSlope (src/d8.cpp:359-409, 103-177) as one 3x3
//                        stencil: fel -> p, sd8.  dontCross() cannot fire in the first pass and the
//                        "neighbour points back" branch is unreachable, so the
//                        pass is a pure per-cell function of the 3x3 window.
//   flat resolution      resolveflats (src/d8.cpp:459-680).  The reference re-sweeps every flat cell
//                        once per level (N^1.5); both of its relaxations are breadth-first level fields -
//                        fixed points of a monotone operator - and run on the tile relaxation engine
//                        (flats.hpp, tile_relax.hpp: 64 x 64 tiles to their local fixed point, rounds):
//                          incfall: elev2(c) = level at which c stops incrementing
//                                   level 1  = a non-crossing neighbour is <= and has a direction
//                                   level 2 += an equal, non-crossing neighbour that is NOT a flat cell
//                                              and has no direction (its elev2 stays 1 < st for st>=2)
//                                   level t  = 1 + min level over equal, non-crossing flat neighbours
//                          incrise: q(c) = BFS level from cells with a strictly higher neighbour through
//                                   8-connected flat cells; s(c) = Tr - q(c) + 1
//                        The sweep counts T, Tr of the reference's loops are reconstructed from the
//                        level counts (they leak into the artificial elevations of pits and into s).
//   classification       flatk::classify_stream_kernel<LV, D8Codes> (a dense queue: one pass over the strip) / flatk::classify_kernel
//                        (a list): seeds of both fields, eligibility masks, activation flags of the tiles
//   d8_setflow2_kernel   setFlow2 (src/d8.cpp:412-454) per flat cell on the artificial surface
//                        (d8_setflow2_stream_kernel for a dense queue; it also writes the next iteration's list)
//   outer iteration      while flats decrease: overwrite the WHOLE elevation grid with (float)elev2
//                        (src/d8.cpp:669-675) and repeat (src/d8.cpp:300-317)
#include <cstring>

#include "context.hpp"
#include "device_common.hpp"
#include "flats.hpp"

namespace {

using namespace tdxk;

constexpr int SLOPE_ROWS = 16;   // rows per lane: a 64 x 64 cell tile per 256-thread block

// One lane walks down a column segment keeping the 3x3 window in registers.  A wave covers 62 output columns: lanes 0 and 63
// only carry the columns beside them, so the west / east neighbours of a row are DPP lane shifts of the ONE value each lane
// loads per row (v_mov_b32_dpp wave_shr:1 / wave_shl:1) - a third of the load instructions and of the L1 traffic of three
// overlapping row loads per lane.
// Flat cells (no positive slope) leave ONE BIT each: lane (x, band of 16 rows) stores its 16-bit row mask to flatbits[band][x].
// (Round 2 appended them to the flat queue here, with one returning atomic per block on ONE counter: ~45 000 same-address atomics at
// ~90 M/s were 0.5 ms of the kernel's 0.95 ms, and every block ended on three barriers waiting for its slot.  The queue is now built
// from the bit masks by two small kernels and a block scan - no atomics anywhere: flat_count_kernel / flat_list_kernel below.)
// Nodata tests: each lane tests the 18 values it loaded once and keeps them as a bit mask; the neighbours' flags are lane shifts of
// the mask (two per 16 rows instead of nine tests per cell).  A cell on the edge of the raster needs no test of its own: a neighbour
// outside the raster reads as nodata (src/linearpart.h:470-483), which makes the cell contaminated just like src/d8.cpp:383-386 does.
constexpr int SLOPE_COLS = 62;
using flatmask_t = uint32_t;    // one bit per row of a lane's segment
// SLOPE_SEG = rows per lane: 32 (34 loads for 32 rows: 6 % overlap; the default) or 16 (12.5 %; TDX_SLOPE_SEG=16, A/B hook)
template <int SLOPE_SEG>
__global__ __launch_bounds__(256) void d8_slope_kernel(const float* __restrict__ Z, int nx, int ny, int y_own0, int y_own1, float nodata,
                                                       const double* __restrict__ fact, int16_t* __restrict__ P,
                                                       float* __restrict__ SD8, flatmask_t* __restrict__ flatbits, int nbx, int xmap) {
    using tilek::lane_left;
    using tilek::lane_right;
    const int bx = tdxk::xcd_block_x(nbx, xmap);
    if (bx < 0) return;
    const int lx = threadIdx.x & 63;
    const int x = bx * SLOPE_COLS - 1 + lx;
    const int band = __builtin_amdgcn_readfirstlane(int(blockIdx.y) * 4 + int(threadIdx.x >> 6));
    const int ybase = y_own0 + band * SLOPE_SEG;
    const bool mine = lx >= 1 && lx <= SLOPE_COLS && x < nx;
    const bool inx = x >= 0 && x < nx;
    const int xc = x < 0 ? 0 : (x >= nx ? nx - 1 : x);
    // all 18 row loads of the lane are issued back to back (rows clamped into the array; validity applied afterwards)
    float z[SLOPE_SEG + 2];
#pragma unroll
    for (int j = 0; j < SLOPE_SEG + 2; j++) {
        const int y = ybase - 1 + j, yc = y < 0 ? 0 : (y >= ny ? ny - 1 : y);
        z[j] = Z[size_t(yc) * size_t(nx) + size_t(xc)];
    }
    unsigned long long nd = 0;   // bit j: window row j of this column is nodata or outside the raster
#pragma unroll
    for (int j = 0; j < SLOPE_SEG + 2; j++) {
        const int y = ybase - 1 + j;
        if (!inx || y < 0 || y >= ny || is_nodata_f(z[j], nodata)) nd |= 1ull << j;
    }
    // this column or one beside it (two 32-bit halves travel through the lane shifts; what a missing lane would hold does not matter:
    // lanes 0 and 63 are not output lanes)
    const unsigned ndlo = unsigned(nd), ndhi = unsigned(nd >> 32);
    const unsigned long long nd3 = nd | ((unsigned long long)(lane_left(ndhi, 0u) | lane_right(ndhi, 0u)) << 32) | (unsigned long long)(lane_left(ndlo, 0u) | lane_right(ndlo, 0u));
    flatmask_t flatmask = 0;
#pragma unroll
    for (int r = 0; r < SLOPE_SEG; r++) {
        const int y = ybase + r;
        const float n1 = z[r], c1 = z[r + 1], s1 = z[r + 2];
        const float n0 = lane_left(n1, nodata), n2 = lane_right(n1, nodata);
        const float c0 = lane_left(c1, nodata), c2 = lane_right(c1, nodata);
        const float s0 = lane_left(s1, nodata), s2 = lane_right(s1, nodata);
        if (mine && y < y_own1) {
            const size_t idx = size_t(y) * size_t(nx) + size_t(x);
            int16_t p = TDX_P_NODATA;
            float sd = -1.0f;
            const float z0 = c1;
            if (((nd3 >> r) & 7ull) == 0ull) {   // the cell and its eight neighbours hold data
                const double* f = fact + size_t(y) * 9;
                // fact[j][k]: 1/dx for E,W; 1/dy for N,S; 1/diag for the diagonals (src/d8.cpp:375)
                const double fE = f[1], fN = f[3], fD = f[2];
                float smax = 0.f;
                int dir = 0;
                // candidate order 1,3,5,7 then 2,4,6,8; strict '>' keeps the first maximum (src/d8.cpp:113-148).
                // calcSlope re-evaluates elevDiff*fact[j][dir] = the winning slope itself; dir == 0 -> 0.
#define TDX_D8_TRY(K, F, ZN)                                           \
    {                                                                  \
        const float slope = (float)((F) * (double)(z0 - (ZN)));        \
        if (slope > smax) { smax = slope; dir = K; }                   \
    }
                TDX_D8_TRY(1, fE, c2) TDX_D8_TRY(3, fN, n1) TDX_D8_TRY(5, fE, c0) TDX_D8_TRY(7, fN, s1)
                TDX_D8_TRY(2, fD, n2) TDX_D8_TRY(4, fD, n0) TDX_D8_TRY(6, fD, s0) TDX_D8_TRY(8, fD, s2)
#undef TDX_D8_TRY
                p = int16_t(dir);
                if (dir == 0) flatmask |= (1u << r);
                sd = smax;
            }
            P[idx] = p;
            if (SD8) SD8[idx] = sd;
        }
    }
    if (mine) flatbits[size_t(band) * size_t(nx) + size_t(x)] = flatmask;
}

// ---- the flat queue from the bit masks: count per block of 2048 masks, scan the block sums, write the cells (no atomics) ----
constexpr int FLATQ_PER_BLOCK = 2048;
__global__ __launch_bounds__(256) void flat_count_kernel(const flatmask_t* __restrict__ bits, size_t nmasks, unsigned* __restrict__ blocksum) {
    const size_t base = size_t(blockIdx.x) * FLATQ_PER_BLOCK + threadIdx.x;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < FLATQ_PER_BLOCK / 256; i++) {
        const size_t m = base + size_t(i) * 256;
        if (m < nmasks) c += unsigned(__popc(bits[m]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    __shared__ unsigned sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blocksum[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
// exclusive scan of the block sums in place (one workgroup; a strip has at most a few tens of thousands of blocks); *total = their sum
__global__ __launch_bounds__(1024) void flat_scan_kernel(unsigned* __restrict__ blocksum, unsigned nblocks, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long sw[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (unsigned b0 = 0; b0 < nblocks; b0 += 1024) {
        const unsigned i = b0 + threadIdx.x;
        const unsigned long long own = i < nblocks ? blocksum[i] : 0u;
        unsigned long long v = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long t = __shfl_up(v, off, 64);
            if (int(threadIdx.x & 63) >= off) v += t;
        }
        if ((threadIdx.x & 63) == 63) sw[threadIdx.x >> 6] = v;
        __syncthreads();
        unsigned long long wave_off = carry;
        for (unsigned w = 0; w < (threadIdx.x >> 6); w++) wave_off += sw[w];
        // offsets are 32-bit: a strip has fewer than 2^32 cells
        if (i < nblocks) blocksum[i] = unsigned(wave_off + v - own);
        __syncthreads();
        if (threadIdx.x == 1023) carry = wave_off + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void flat_list_kernel(const flatmask_t* __restrict__ bits, size_t nmasks, int nx, int y_own0, int seg, const unsigned* __restrict__ blockoff,
                                                        uint32_t* __restrict__ list) {
    // The block's 2048 masks are 32 groups of 64 neighbouring columns (chunk i, wave w); the list takes them group by group and, within a group, ROW BY ROW, so that
    // consecutive entries are neighbouring cells of a raster row (mask by mask - a column's rows - every entry of a wave of the list kernels sat in a row of its own).
    constexpr int NCH = FLATQ_PER_BLOCK / 256;
    const size_t base = size_t(blockIdx.x) * FLATQ_PER_BLOCK + threadIdx.x;
    const int lane = int(threadIdx.x & 63), w = int(threadIdx.x >> 6);
    unsigned mk[NCH];
    __shared__ unsigned scnt[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const size_t m = base + size_t(i) * 256;
        mk[i] = m < nmasks ? bits[m] : 0u;
        unsigned c = unsigned(__popc(mk[i]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        if (lane == 0) scnt[i][w] = c;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        unsigned any = mk[i];   // rows in which some column of the group has a flat cell (wave-uniform)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) any |= __shfl_xor(any, off, 64);
        if (!any) continue;
        unsigned pos = blockoff[blockIdx.x];
        for (int j = 0; j < i; j++) pos += scnt[j][0] + scnt[j][1] + scnt[j][2] + scnt[j][3];
        for (int u = 0; u < w; u++) pos += scnt[i][u];
        const size_t m = base + size_t(i) * 256, mc = m < nmasks ? m : nmasks - 1;
        const unsigned band = unsigned(mc / size_t(nx)), x = unsigned(mc - size_t(band) * size_t(nx));
        const size_t c0 = size_t(y_own0 + int(band) * seg) * size_t(nx) + size_t(x);
        for (unsigned rows = any; rows; rows &= rows - 1u) {
            const int r = __ffs(int(rows)) - 1;
            const bool mine = (mk[i] >> r) & 1u;
            const unsigned long long b = __ballot(mine);
            if (mine) list[pos + unsigned(__popcll(b & below))] = uint32_t(c0 + size_t(r) * size_t(nx));
            pos += unsigned(__popcll(b));
        }
    }
}

// D8 dontCross (src/d8.cpp:54-100) for an interior cell at linear index c
__device__ __forceinline__ bool dont_cross_d8(const int16_t* __restrict__ P, size_t c, int nx, int k) {
    switch (k) {
        case 2: return P[c + 1] == 4 || P[c - nx] == 8;
        case 4: return P[c - nx] == 6 || P[c - 1] == 2;
        case 6: return P[c + nx] == 4 || P[c - 1] == 8;
        case 8: return P[c + 1] == 6 || P[c + nx] == 2;
        default: return false;
    }
}

// (the streaming classification of flat resolution is flatk::classify_stream_kernel, flats.hpp: shared with DinfFlowDir)
// the direction codes in one-hot form: bit c for a code c in 0 .. 8, nothing for nodata / any other value
struct D8Codes {
    using Raw = int;
    const int16_t* P;
    __device__ __forceinline__ int load(size_t o) const { return P[o]; }
    static __device__ __forceinline__ unsigned onehot(int p) { return (1u << min(unsigned(p), 31u)) & 0x1FFu; }
};

struct D8Traits {
    const int16_t* P;
    __device__ __forceinline__ bool dont_cross(size_t c, int nx, int k) const { return dont_cross_d8(P, c, nx, k); }
    // "adjacent cell drains": flowDir in 1..8 (src/d8.cpp:537)
    __device__ __forceinline__ bool has_direction(size_t n) const { const int16_t v = P[n]; return v > 0 && v < 9; }
};

// setFlow2 (src/d8.cpp:412-454) for every cell of the flat list
template <class LV>
__global__ __launch_bounds__(256) void d8_setflow2_kernel(const float* __restrict__ Z, int nx, const double* __restrict__ fact,
                                                          const uint32_t* __restrict__ list, unsigned long long nq,
                                                          const LV* __restrict__ lvl, const LV* __restrict__ rq,
                                                          FlatLevels fl, int16_t* __restrict__ P) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const size_t c = list[q];
    const int y = int(c / size_t(nx));
    const double* f = fact + size_t(y) * 9;
    const int e2c = int(flat_elev2<LV>(lvl[c], rq[c], fl));
    const float z0 = Z[c];
    float smax = 0.f;
    int16_t dir = P[c];   // 0, or nodata for a cell marked as pit in this iteration
    const int order[8] = {1, 3, 5, 7, 2, 4, 6, 8};
    for (int o = 0; o < 8; o++) {
        const int k = order[o];
        const size_t n = size_t(ptrdiff_t(c) + ptrdiff_t(d2(k)) * nx + d1(k));
        if (rq[n] > 0) {   // dn > 0: neighbour is a marked flat cell
            const int e2n = int(flat_elev2<LV>(lvl[n], rq[n], fl));
            const float slope = (float)(f[k] * (double)(e2c - e2n));
            if (slope > smax) { dir = int16_t(k); smax = slope; }
        } else {
            const float ed = z0 - Z[n];
            if (ed >= 0) { dir = int16_t(k); break; }
        }
    }
    P[c] = dir;
}

// Q' = cells of Q still 0 (8 list entries per lane, one atomic per block)
__global__ __launch_bounds__(256) void d8_recollect_kernel(const int16_t* __restrict__ P, const uint32_t* __restrict__ list,
                                                           unsigned long long nq, uint32_t* __restrict__ out,
                                                           unsigned long long* __restrict__ counter) {
    const unsigned long long q0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 8;   // eight consecutive entries per thread: the list keeps its order
    uint32_t keep[8];
    unsigned cnt = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned long long q = q0 + (unsigned long long)i;
        if (q < nq) {
            const uint32_t c = list[q];
            if (P[c] == 0) keep[cnt++] = c;
        }
    }
    unsigned long long pos = block_reserve(cnt, counter);
    for (unsigned i = 0; i < cnt; i++) out[pos + i] = keep[i];
}

template <class LV>
__global__ __launch_bounds__(256) void d8_mark_pits_kernel(const uint32_t* __restrict__ list, unsigned long long nq,
                                                           const LV* __restrict__ lvl, int16_t* __restrict__ P) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const size_t c = list[q];
    if (lvl[c] == 0) P[c] = TDX_P_NODATA;   // never stopped incrementing: enclosed pit (src/d8.cpp:559-585)
}

// mark_pits + setFlow2 + re-collection of the cells that are still flat as ONE streaming pass over the strip (used
// while a large share of the raster is in the queue; the list kernels above are cheaper for a sparse queue).
// A cell's new direction depends on the elevations and level markers of its neighbours, never on their directions,
// so P can be rewritten in place.
// A wave covers 62 columns x 16 rows: lanes 0 and 63 only carry the columns beside them, so every neighbour value is a
// DPP lane shift of a register (3 coalesced loads per window row instead of 9), and only the window rows next to a row
// that holds a flat cell are loaded at all (wave-uniform branches; flats are clustered).
constexpr int SF2_COLS = 62;
template <class LV>
__global__ __launch_bounds__(256) void d8_setflow2_stream_kernel(const float* __restrict__ Z, int nx, int ny, int y_own0, int y_own1,
                                                                 const double* __restrict__ fact, const LV* __restrict__ lvl,
                                                                 const LV* __restrict__ rq, FlatLevels fl, int16_t* __restrict__ P,
                                                                 uint32_t* __restrict__ qnext, unsigned long long* __restrict__ counter, int nbx, int xmap,
                                                                 LV* __restrict__ lvl_next, LV* __restrict__ rq_next) {
    using tilek::lane_left;
    using tilek::lane_right;
    const int bx = tdxk::xcd_block_x(nbx, xmap);
    if (bx < 0) return;
    const int lx = threadIdx.x & 63;
    const int x = bx * SF2_COLS - 1 + lx;
    const int ybase = __builtin_amdgcn_readfirstlane(y_own0 + blockIdx.y * (4 * SLOPE_ROWS) + (threadIdx.x >> 6) * SLOPE_ROWS);
    const bool mine = lx >= 1 && lx <= SF2_COLS && x < nx;
    const int xc = x < 0 ? 0 : (x >= nx ? nx - 1 : x);
    int16_t p[SLOPE_ROWS];
#pragma unroll
    for (int r = 0; r < SLOPE_ROWS; r++) {
        const int y = ybase + r, yc = y >= ny ? ny - 1 : y;
        p[r] = P[size_t(yc) * size_t(nx) + size_t(xc)];
    }
    unsigned flat = 0;      // this lane's flat cells
    unsigned rows0 = 0;     // wave-uniform: rows with a flat cell
#pragma unroll
    for (int r = 0; r < SLOPE_ROWS; r++) {
        const bool f0 = mine && ybase + r < y_own1 && p[r] == 0;
        if (f0) flat |= 1u << r;
        if (__ballot(f0) != 0ull) rows0 |= 1u << r;
    }
    unsigned keep = 0;
    if (rows0) {
        // window row j = raster row ybase - 1 + j; row r looks at window rows r, r + 1, r + 2
        const unsigned need = rows0 | (rows0 << 1) | (rows0 << 2);
        float z[SLOPE_ROWS + 2];
        int e2[SLOPE_ROWS + 2], rr[SLOPE_ROWS + 2];
#pragma unroll
        for (int j = 0; j < SLOPE_ROWS + 2; j++) {
            z[j] = 0.f; e2[j] = -1; rr[j] = -1;
            if ((need >> j) & 1u) {   // (flat cells are interior cells: the rows and columns they look at exist; clamping only keeps halo lanes in bounds)
                const int y = ybase - 1 + j, yc = y < 0 ? 0 : (y >= ny ? ny - 1 : y);
                const size_t o = size_t(yc) * size_t(nx) + size_t(xc);
                z[j] = Z[o]; e2[j] = lvl[o]; rr[j] = rq[o];
            }
        }
        unsigned pit = 0;       // window rows whose cell never stopped incrementing (lvl == 0)
#pragma unroll
        for (int j = 0; j < SLOPE_ROWS + 2; j++) {
            if (e2[j] == 0) pit |= 1u << j;
            e2[j] = int(flat_elev2<LV>(e2[j], rr[j], fl));
        }
#pragma unroll
        for (int r = 0; r < SLOPE_ROWS; r++) {
            if (!((rows0 >> r) & 1u)) continue;
            const int y = ybase + r;
            // neighbours k = 1..8 (E NE N NW W SW S SE): elevation, rq marker, elev2
            const float zk[9] = {0.f, lane_right(z[r + 1], 0.f), lane_right(z[r], 0.f), z[r], lane_left(z[r], 0.f), lane_left(z[r + 1], 0.f),
                                 lane_left(z[r + 2], 0.f), z[r + 2], lane_right(z[r + 2], 0.f)};
            const int rk[9] = {0, lane_right(rr[r + 1], -1), lane_right(rr[r], -1), rr[r], lane_left(rr[r], -1), lane_left(rr[r + 1], -1),
                               lane_left(rr[r + 2], -1), rr[r + 2], lane_right(rr[r + 2], -1)};
            const int ek[9] = {0, lane_right(e2[r + 1], 0), lane_right(e2[r], 0), e2[r], lane_left(e2[r], 0), lane_left(e2[r + 1], 0),
                               lane_left(e2[r + 2], 0), e2[r + 2], lane_right(e2[r + 2], 0)};
            if ((flat >> r) & 1u) {
                const double* f = fact + size_t(y) * 9;
                const int e2c = e2[r + 1];
                const float z0 = z[r + 1];
                float smax = 0.f;
                int16_t dir = (fl.has_pits && ((pit >> (r + 1)) & 1u)) ? TDX_P_NODATA : int16_t(0);   // enclosed pit (src/d8.cpp:559-585)
                bool done = false;
                // candidate order 1,3,5,7,2,4,6,8; the first non-marked neighbour that is not higher ends the search (src/d8.cpp:444-451)
                // (as straight-line selects instead of branches the compiler interleaves all sixteen rows' fp64 products: 242 VGPRs, twice the instructions - not kept)
#define TDX_SF2(K)                                                                     \
    if (!done) {                                                                        \
        const float slope = (float)(f[K] * (double)(e2c - ek[K]));                      \
        const bool marked = rk[K] > 0;                                                  \
        const bool take = marked && slope > smax;                                       \
        const bool stop = !marked && z0 - zk[K] >= 0;                                   \
        dir = (take || stop) ? int16_t(K) : dir;                                        \
        smax = take ? slope : smax;                                                     \
        done = stop;                                                                    \
    }
                TDX_SF2(1) TDX_SF2(3) TDX_SF2(5) TDX_SF2(7) TDX_SF2(2) TDX_SF2(4) TDX_SF2(6) TDX_SF2(8)
#undef TDX_SF2
                P[size_t(y) * size_t(nx) + size_t(x)] = dir;
                if (dir == 0) keep |= 1u << r;
            }
        }
    }
    // The markers of the NEXT iteration (0 on the cells that are still flat, -1 elsewhere) go to a second pair of rasters on the way: the pass visits every owned
    // cell anyway, and two fills of the level rasters plus the list pass that re-marked the queue were 0.27 ms of the second iteration at 16384^2
    if (lvl_next != nullptr) {
#pragma unroll
        for (int r = 0; r < SLOPE_ROWS; r++) {
            if (mine && ybase + r < y_own1) {
                const size_t idx = size_t(ybase + r) * size_t(nx) + size_t(x);
                const LV m = ((keep >> r) & 1u) ? LV(0) : LV(-1);
                lvl_next[idx] = m;
                rq_next[idx] = m;
            }
        }
    }
    // The wave's slots are filled row by row, so that neighbouring list entries are neighbouring cells of a raster row: the kernels of the next
    // iteration gather the 3 x 3 windows of 64 consecutive entries per wave (lane by lane - a lane's column segment - every entry of a wave sat in
    // a row of its own).
    const unsigned long long pos = block_reserve(unsigned(__popc(keep)), counter);
    const unsigned long long wbase = (unsigned long long)(unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(pos))))) |
                                     ((unsigned long long)(unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(pos >> 32))))) << 32);   // lane 0's slot = the wave's first
    const unsigned long long below = (1ull << lx) - 1ull;
    unsigned rowoff = 0;
#pragma unroll
    for (int r = 0; r < SLOPE_ROWS; r++) {
        const unsigned long long b = __ballot((keep >> r) & 1u);
        if ((keep >> r) & 1u) qnext[wbase + rowoff + unsigned(__popcll(b & below))] = uint32_t(size_t(ybase + r) * size_t(nx) + size_t(x));
        rowoff += unsigned(__popcll(b));
    }
}

}  // namespace

int tdx_build_fact_table(tdx_context* ctx, int64_t ny, const double* dxc, const double* dyc, double** d_fact_out) {
    // fact[j][k] = 1/sqrt((d1*dx)^2 + (d2*dy)^2) in double on the host (src/d8.cpp:369-377); index 0 = 0
    static const int hd1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1};
    static const int hd2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};
    // (the slot is this table's alone, so a table built from the same cell sizes is still there: 0.6 ms of host work per call at 16384 rows otherwise)
    const size_t rows = size_t(ny);
    if (ctx->fact_dxc.size() == rows && ctx->fact_dyc.size() == rows && memcmp(ctx->fact_dxc.data(), dxc, rows * 8) == 0 &&
        memcmp(ctx->fact_dyc.data(), dyc, rows * 8) == 0) {
        *d_fact_out = static_cast<double*>(ctx->scratch(TDX_S_FACT, rows * 9 * sizeof(double)));
        return *d_fact_out ? TDX_OK : TDX_ERR_NOMEM;
    }
    ctx->fact_dxc.clear(); ctx->fact_dyc.clear();
    std::vector<double> fact(rows * 9, 0.0);
    for (int64_t m = 0; m < ny; m++)
        for (int k = 1; k <= 8; k++)
            fact[size_t(m) * 9 + size_t(k)] = (double)(1. / sqrt(hd1[k] * hd1[k] * dxc[m] * dxc[m] + hd2[k] * hd2[k] * dyc[m] * dyc[m]));
    double* d_fact = static_cast<double*>(ctx->scratch(TDX_S_FACT, fact.size() * sizeof(double)));
    if (!d_fact) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_fact, fact.data(), fact.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // `fact` is a local
    ctx->fact_dxc.assign(dxc, dxc + rows);
    ctx->fact_dyc.assign(dyc, dyc + rows);
    *d_fact_out = d_fact;
    return TDX_OK;
}

// One strip of setdird8() (src/d8.cpp:227-320).  Counts that steer the outer loop (flats left) are summed
// over the ranks, so every rank takes the same branches (src/d8.cpp:294-316).
template <class LV>
static int d8flowdir_levels(tdx_context* ctx, const Strip& st, float* d_fel, float fel_nodata, const double* dxc, const double* dyc, int16_t* d_p,
                          float* d_sd8, tdx_stats* stats) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = st.nx;
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    double* d_fact = nullptr;
    int rc = tdx_build_fact_table(ctx, st.ny_arr, dxc, dyc, &d_fact);
    if (rc != TDX_OK) return rc;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    // flat-resolution markers and the flat queue are produced by the slope pass itself
    LV* lvl = static_cast<LV*>(ctx->scratch(TDX_S_A, n * sizeof(LV)));
    LV* rq = static_cast<LV*>(ctx->scratch(TDX_S_B, n * sizeof(LV)));
    uint32_t* qlist = static_cast<uint32_t*>(ctx->scratch(TDX_S_C, n * 4));
    if (!lvl || !rq || !qlist) return TDX_ERR_NOMEM;

    ctx->begin_call(stats);
    strip_mark(ctx, st, "d8flowdir");
    ctx->phase = "slope pass";
    rc = strip_exchange<float>(ctx, st, d_fel, fel_nodata);   // elevation halo rows
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    // one bit per owned cell: bands of SLOPE_SEG rows x nx columns of row masks (the slope pass's lane = one mask)
    static const int seg = (getenv("TDX_SLOPE_SEG") && atoi(getenv("TDX_SLOPE_SEG")) == 16) ? 16 : 32;   // rows per lane of the slope pass
    const int nband = ((st.y1 - st.y0 + 4 * seg - 1) / (4 * seg)) * 4;
    const size_t nmasks = size_t(nband) * size_t(inx);
    const unsigned nqblocks = unsigned((nmasks + FLATQ_PER_BLOCK - 1) / FLATQ_PER_BLOCK);
    flatmask_t* flatbits = static_cast<flatmask_t*>(ctx->scratch(TDX_S_N, nmasks * sizeof(flatmask_t)));
    unsigned* blocksum = static_cast<unsigned*>(ctx->scratch(TDX_S_O, size_t(nqblocks) * 4));
    if (!flatbits || !blocksum) return TDX_ERR_NOMEM;
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        const int nbx = (inx + SLOPE_COLS - 1) / SLOPE_COLS, xmap = tdx_xcd_map() ? 1 : 0;
        dim3 grid(tdx_xcd_grid_x(unsigned(nbx)), unsigned(nband / 4));
        if (seg == 16) hipLaunchKernelGGL(d8_slope_kernel<16>, grid, dim3(256), 0, s, d_fel, inx, st.ny_arr, st.y0, st.y1, fel_nodata, d_fact, d_p, d_sd8, flatbits, nbx, xmap);
        else hipLaunchKernelGGL(d8_slope_kernel<32>, grid, dim3(256), 0, s, d_fel, inx, st.ny_arr, st.y0, st.y1, fel_nodata, d_fact, d_p, d_sd8, flatbits, nbx, xmap);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    {
        TdxSpan sp(ctx, TDX_K_MISC);
        hipLaunchKernelGGL(flat_count_kernel, dim3(nqblocks), dim3(256), 0, s, flatbits, nmasks, blocksum);
        hipLaunchKernelGGL(flat_scan_kernel, dim3(1), dim3(1024), 0, s, blocksum, nqblocks, d_cnt);
    }
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
    unsigned long long nq = ctx->h_mail[0];        // flats of this strip
    int64_t total = int64_t(nq);                    // flats of the whole raster
    rc = strip_allreduce(ctx, st, &total, 1, TDX_OP_SUM);
    if (rc != TDX_OK) return rc;
    if (stats) { stats->flats_initial = total; stats->flats_left = total; }

    // The first queue as a LIST is only built when something will read it: a dense queue (more than 1/16 of the strip; a third of the
    // raster at BASELINE.json configs[1]) is classified, evaluated (level statistics) and re-directed by streaming passes, and the marker
    // reset of a later iteration rewrites the whole strip anyway (flats_reset_markers_after).  The choice is this rank's own: nothing that
    // is exchanged depends on it.
    const bool force_list = getenv("TDX_FLATS_LIST") != nullptr;   // (test hook, read per call: the first queue as a list for a dense queue too)
    const bool have_list = force_list || nq <= n / 16;
    if (total > 0 && nq > 0 && have_list) {
        TdxSpan sp(ctx, TDX_K_MISC);
        hipLaunchKernelGGL(flat_list_kernel, dim3(nqblocks), dim3(256), 0, s, flatbits, nmasks, inx, st.y0, seg, blocksum, qlist);
    }
    if (total > 0) {
        rc = strip_exchange<int16_t>(ctx, st, d_p, TDX_P_NODATA);
        if (rc != TDX_OK) return rc;
        // working storage for flat resolution
        uint32_t* qnext = static_cast<uint32_t*>(ctx->scratch(TDX_S_D, size_t(nq) * 4));
        if (!qnext) return TDX_ERR_NOMEM;
        float* zwork = nullptr;            // allocated only if a second iteration is needed
        const float* zcur = d_fel;
        FlatBuffersT<LV> fbuf{lvl, rq};

        // first call of resolveflats: queue = cells with flowDir == 0 (src/d8.cpp:492-503) = qlist from the slope pass
        int64_t last = total;
        const int64_t total0 = total;
        bool sparse = false;                // this iteration works from lists (few flats left): no pass over the whole raster
        unsigned long long nq_old = 0;      // cells of the previous iteration's queue (in qnext after the swap)
        bool old_list_valid = false;        // ... and whether qnext really holds them
        LV *lvl_next = nullptr, *rq_next = nullptr;   // the next iteration's markers, written by the streaming setFlow2 (rasters of their own)
        bool markers_ready = false;
        for (int iteration = 1;; iteration++) {
            const FlatPhases& ph = flat_phases(iteration);
            ctx->phase = ph.classify;
            // every call re-creates elev2 / dn (src/d8.cpp:483-486): the streaming classification rewrites all markers
            FlatLevels fl;
            D8Traits tr{d_p};
            const float* zc = zcur;
            const StreamClassifyFn classify = [&](const tilek::TileGeom& g, uint8_t* fmask, uint8_t* rmask, uint32_t* tile_flags, uint8_t* tile_masked, uint8_t* notfull) {
                const int nbx = (st.nx + flatk::CLS_COLS - 1) / flatk::CLS_COLS;
                const dim3 grid(tdx_xcd_grid_x(unsigned(nbx)), (st.y1 - st.y0 + 4 * flatk::CLS_ROWS - 1) / (4 * flatk::CLS_ROWS));
                hipLaunchKernelGGL((flatk::classify_stream_kernel<LV, D8Codes>), grid, dim3(256), 0, s, zc, D8Codes{d_p}, st.nx, st.ny_arr, st.y0, st.y1, g.tiles_x, lvl, rq, fmask,
                                   rmask, tile_flags, tile_masked, nbx, tdx_xcd_map() ? 1 : 0, notfull);
            };
            if (sparse && markers_ready) {
                // the streaming setFlow2 of the previous iteration left this iteration's markers in the second pair of rasters
                std::swap(lvl, lvl_next);
                std::swap(rq, rq_next);
                fbuf = FlatBuffersT<LV>{lvl, rq};
                rc = strip_exchange<LV>(ctx, st, lvl, LV(-1));   // queue membership of the neighbours' boundary rows (flats_reset_markers)
                if (rc != TDX_OK) return rc;
                rc = strip_exchange<LV>(ctx, st, rq, LV(-1));
                if (rc != TDX_OK) return rc;
            } else if (sparse) {
                // the previous queue exists as a list only if it was built (first queue of a dense strip: bit masks only) - without one every marker is rewritten
                rc = old_list_valid ? flats_reset_markers_after(ctx, st, qnext, nq_old, qlist, nq, lvl, rq) : flats_reset_markers(ctx, st, qlist, nq, lvl, rq);
                if (rc != TDX_OK) return rc;
            }
            markers_ready = false;
            // (the first queue of a dense strip exists as bit masks only: no list)
            rc = flats_bfs<D8Traits, LV>(ctx, tr, zcur, st, (nq_old == 0 && !have_list) ? nullptr : qlist, nq, fbuf, &fl, stats, sparse ? nullptr : &classify, iteration);
            if (rc != TDX_OK) return rc;
            ctx->phase = ph.directions;
            {
                TdxSpan sp(ctx, TDX_K_FLATDIR);
                // (the queue-length counter of the next iteration is word 4 of the stage counters: cleared by flatk::prepare_kernel of THIS iteration, untouched since)
                unsigned long long* d_next = d_cnt + 4;
                if (nq > n / 32) {   // dense queue: one streaming pass
                    const int nbx = (st.nx + SF2_COLS - 1) / SF2_COLS;
                    const dim3 grid(tdx_xcd_grid_x(unsigned(nbx)), (st.y1 - st.y0 + 4 * SLOPE_ROWS - 1) / (4 * SLOPE_ROWS));
                    static const bool no_next = getenv("TDX_FLATS_NO_NEXT_MARKERS") != nullptr;   // (A/B hook)
                    if (!no_next && !lvl_next) {
                        lvl_next = static_cast<LV*>(ctx->scratch(TDX_S_P, n * sizeof(LV)));
                        rq_next = static_cast<LV*>(ctx->scratch(TDX_S_R, n * sizeof(LV)));
                        if (!lvl_next || !rq_next) return TDX_ERR_NOMEM;
                    }
                    hipLaunchKernelGGL((d8_setflow2_stream_kernel<LV>), grid, dim3(256), 0, s, zcur, inx, st.ny_arr, st.y0, st.y1, d_fact, lvl, rq, fl, d_p, qnext,
                                       d_next, nbx, tdx_xcd_map() ? 1 : 0, lvl_next, rq_next);
                    markers_ready = lvl_next != nullptr;
                } else if (nq) {
                    if (fl.has_pits)
                        hipLaunchKernelGGL((d8_mark_pits_kernel<LV>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, s, qlist, nq, lvl, d_p);
                    hipLaunchKernelGGL((d8_setflow2_kernel<LV>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, s, zcur, inx, d_fact, qlist, nq, lvl, rq, fl, d_p);
                    hipLaunchKernelGGL(d8_recollect_kernel, dim3(tdx_blocks_for(nq, 2048)), dim3(256), 0, s, d_p, qlist, nq, qnext, d_next);
                }
                TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_next, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
                if (stats) stats->launches[TDX_K_FLATDIR] += 2 + (fl.has_pits ? 1 : 0);
            }
            const unsigned long long nleft = ctx->h_mail[0];
            total = int64_t(nleft);
            rc = strip_allreduce(ctx, st, &total, 1, TDX_OP_SUM);
            if (rc != TDX_OK) return rc;
            rc = strip_exchange<int16_t>(ctx, st, d_p, TDX_P_NODATA);   // directions of the neighbours' boundary rows
            if (rc != TDX_OK) return rc;
            if (stats) { stats->flat_iterations++; stats->flats_left = total; }
            if (!(total > 0 && total < last)) break;     // src/d8.cpp:307
            ctx->phase = ph.next;
            // another iteration: elevDEM := (float)elev2 for ALL cells (src/d8.cpp:669-675)
            if (!zwork) {
                zwork = static_cast<float*>(ctx->scratch(TDX_S_I, n * 4));
                if (!zwork) return TDX_ERR_NOMEM;
            }
            // (the choice must be the same on every rank - the list path exchanges two more halo rows -, hence global counts)
            static const bool no_sparse = getenv("TDX_FLATS_DENSE") != nullptr;
            sparse = !no_sparse && (total * 8 <= total0 || total <= (int64_t(1) << 20));
            rc = sparse ? flats_overwrite_elevation_sparse(ctx, inx, qnext, nleft, lvl, rq, fl, zwork) : flats_overwrite_elevation(ctx, n, lvl, rq, fl, zwork);
            if (rc != TDX_OK) return rc;
            zcur = zwork;
            old_list_valid = nq_old != 0 || have_list;   // (the list this iteration worked from: every queue after the first is a list)
            nq_old = nq;
            std::swap(qlist, qnext);
            nq = nleft;
            last = total;
        }
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    ctx->end_call();
    return TDX_OK;
}

// The level fields are int16 like the reference's elev2 / dn / s partitions (src/d8.cpp:483,486,595); a flat deeper than they hold (32 766 levels - the
// reference's short counters wrap there, so nothing there is defined to be equal to) starts the call over on int32 fields: the slope pass is repeated, the
// inputs are untouched (elevDEM := elev2 works on a copy).  Every rank sees the same level maxima, so every rank takes the same path.
// TDX_LEVELS_INT32=1 (test hook, read per call): int32 fields from the start.
static int d8flowdir_impl(tdx_context* ctx, const Strip& st, float* d_fel, float fel_nodata, const double* dxc, const double* dyc, int16_t* d_p,
                          float* d_sd8, tdx_stats* stats) {
    int rc = getenv("TDX_LEVELS_INT32") ? TDX_FLATS_TOO_DEEP : d8flowdir_levels<int16_t>(ctx, st, d_fel, fel_nodata, dxc, dyc, d_p, d_sd8, stats);
    if (rc == TDX_FLATS_TOO_DEEP) rc = d8flowdir_levels<int32_t>(ctx, st, d_fel, fel_nodata, dxc, dyc, d_p, d_sd8, stats);
    return rc;
}

extern "C" int tdx_d8flowdir_dev(tdx_context* ctx, const float* d_fel, int64_t nx, int64_t ny, float fel_nodata,
                                 const double* dxc, const double* dyc, int16_t* d_p, float* d_sd8, tdx_stats* stats) {
    if (!ctx || !d_fel || !d_p || !dxc || !dyc || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_d8flowdir_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return d8flowdir_impl(ctx, strip_single(int(nx), int(ny)), const_cast<float*>(d_fel), fel_nodata, dxc, dyc, d_p, d_sd8, stats);
}

extern "C" int tdx_d8flowdir_strip(tdx_context* ctx, const tdx_comm* comm, float* d_fel, int64_t nx, int64_t ny_local, float fel_nodata,
                                   const double* dxc, const double* dyc, int16_t* d_p, float* d_sd8, tdx_stats* stats) {
    if (!ctx || !d_fel || !d_p || !dxc || !dyc || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_d8flowdir_strip: bad argument");
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return d8flowdir_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_fel, fel_nodata, dxc, dyc, d_p, d_sd8, stats);
}

extern "C" int tdx_d8flowdir(tdx_context* ctx, const float* fel, int64_t nx, int64_t ny, float fel_nodata,
                             const double* dxc, const double* dyc, int16_t* p, float* sd8, tdx_stats* stats) {
    if (!ctx || !fel || !p || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_d8flowdir: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_z = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    int16_t* d_p = static_cast<int16_t*>(ctx->scratch(TDX_S_IO1, n * 2));
    float* d_s = sd8 ? static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4)) : nullptr;
    if (!d_z || !d_p || (sd8 && !d_s)) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_z, fel, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = tdx_d8flowdir_dev(ctx, d_z, nx, ny, fel_nodata, dxc, dyc, d_p, d_s, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(p, d_p, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (sd8) TDX_HIP_CHECK(ctx, hipMemcpyAsync(sd8, d_s, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
