// Tile dependency sweep for the D8 flow algebra: weighted AreaD8, D8FlowPathExtremeUp (aread8.hip) and GridNet (gridnet.hip).
//
// Same idea as the D-infinity sweep of areadinf.hip (dsweep): a cell's value depends only on the values of the cells that
// drain into it (the reference folds them in k = 1..8 order: src/aread8.cpp:231-256, src/D8flowpathextremeup.cpp:167-199,
// src/gridnet.cpp:380-426), never on the schedule.  The atomic pull walks pay two device-scope memory round trips per cell of
// the longest flow path and one device-scope atomic per flow link; here a 1024-thread workgroup stages a 64 x 64 tile + ring of
// the result(s) and the per-cell info word in LDS, counts the pending contributors of every pending cell FROM THE STAGED VALUES
// (no global counters), evaluates ready cells in lockstep sweeps while the tile is fresh, then lets lanes walk chains
// downstream inside LDS (one returning LDS atomic per hop; D8 has one target per cell, so a chain never forks).  Finished
// cells are written back; a tile whose cells drain into a neighbouring tile raises that tile's flag for the next ROUND
// (the schedule of tile_relax.hpp).  Rounds = tile crossings of the longest flow path; strips exchange the boundary rows of
// the result between runs of rounds and re-activate the tiles that see a changed halo cell.
//
// Per-cell info word (d8sweep::setup_kernel, one streaming pass over the direction grid):
//   [0:8)   neighbour k is a contributor for the DEPENDENCY count (in-degree of initNeighborD8up, src/commonLib.cpp:251-282:
//           its direction code is 0..8 and code - k == +-4 - which includes a p == 0 cell at k == 4, the reference's quirk)
//   [8]     a neighbour is missing (off the raster or nodata): edge contamination (src/aread8.cpp:241-242)
//   [9:13)  the cell's own direction code (0..8; 15: none / sink)
//   [13]    the cell participates      [14] the cell can never become ready (a p == 0 contributor is counted but never drains)
//   [16:24) neighbour k contributes to the VALUE (GridNet: its code is > 0, its mask value passes and it drains into the cell;
//           the AreaD8 family: same as [0:8))
//   [24]    the cell's own mask value passes (GridNet: an unmasked cell completes without being evaluated)
#pragma once
#include "context.hpp"
#include "device_common.hpp"
#include "flats.hpp"
#include "strips.hpp"
#include "tile_relax.hpp"

#include <cstring>

namespace d8sweep {
using namespace tdxk;
// Two tile geometries over the same arrays (the state of a sweep is the work array alone): 32 x 32 tiles (256 threads, a few tens
// of KB of LDS: several tiles per CU) carry the BULK rounds, in which every tile of the raster is active and a tile's time is its
// longest in-tile chain - what counts is how many tiles a CU works on at once; 64 x 64 tiles (1024 threads, one tile per CU) carry
// the TAIL, where a round is one tile crossing of the longest flow path and fewer crossings win.
template <int TSZ>
struct Dim {
    static constexpr int TS = TSZ, LH = TSZ + 2, NT = TSZ == 64 ? 1024 : 256, RPL = TSZ * TSZ / NT, NSTAGE = (LH * LH + NT - 1) / NT;
};
constexpr int LH = tilek::TS + 2;   // (row pitch of the staged window of the 64 x 64 geometry: not used by the engine itself)
constexpr int BULK_SWEEPS = 12;   // (the D-infinity limited accumulations: dinflim.hip)
constexpr int BULK_SWEEPS_D8 = 3;  // one-receiver policies: the sources and the first confluences in lockstep, the rest by walks (measured at 16384^2, weighted AreaD8 / GridNet:
                                   // 1: 49.8 / 55.5 ms, 2: 49.8 / 54.6, 3: 49.6 / 54.7, 4: 50.1 / 54.9, 6: 50.3 / 56.1, 8: 51.1 / 57.3, 12: 52.4 / 58.0, 16: 53.2 / 58.7)
constexpr uint32_t PENDING_BITS = 0x7FC0DEADu;   // a quiet NaN no arithmetic produces: "participating, not evaluated yet"
constexpr unsigned INFO_CON = 1u << 8, INFO_PART = 1u << 13, INFO_DEAD = 1u << 14, INFO_OWNMASK = 1u << 24;
constexpr int16_t P_OUTSIDE = 16, P_SINK = 32;   // re-coded directions of outlets mode (aread8.hip)

__device__ __forceinline__ bool pending(float v) { return __float_as_uint(v) == PENDING_BITS; }

// gn_mask: GridNet's mask grid (cells with mask >= thresh are evaluated), nullptr otherwise; gridnet != 0 selects GridNet's
// value-contributor rule.  participates(p): code 0..8 or P_SINK (AreaD8 family) / any non-nodata code inside `reach` (GridNet).
static __global__ __launch_bounds__(256) void setup_kernel(const int16_t* __restrict__ P, int nx, int ny, int16_t nodata, int gridnet,
                                                    const int32_t* __restrict__ gn_mask, int thresh, const int32_t* __restrict__ reach,
                                                    uint32_t* __restrict__ info) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    const int16_t p = P[idx];
    unsigned inf = 0;
    bool part;
    if (gridnet) part = !is_nodata_s(p, nodata) && (!reach || reach[idx] == 1);
    else part = !is_nodata_s(p, nodata) && ((p >= 0 && p <= 8) || p == P_SINK);
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        const int xn = x + d1(k), yn = y + d2(k);
        if (xn < 0 || xn >= nx || yn < 0 || yn >= ny) { inf |= INFO_CON; continue; }
        const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
        const int16_t pn = P[n];
        if (is_nodata_s(pn, nodata)) { inf |= INFO_CON; continue; }
        const bool drains = (pn - k == 4 || pn - k == -4);
        if (!drains) continue;
        if (gridnet) {
            // in-degree: every non-nodata neighbour that drains into the cell (src/gridnet.cpp:238-267; with outlets: :285-300)
            inf |= 1u << (k - 1);
            if (pn == 0) inf |= INFO_DEAD;
            if (pn > 0 && (!gn_mask || gn_mask[n] >= thresh)) inf |= 1u << (16 + k - 1);
        } else if (pn >= 0 && pn <= 8) {
            inf |= (1u << (k - 1)) | (1u << (16 + k - 1));
            if (pn == 0) inf |= INFO_DEAD;   // counted in the in-degree, never decremented (src/aread8.cpp:262)
        }
    }
    const unsigned code = (p >= 0 && p <= 8) ? unsigned(p) : 15u;
    inf |= code << 9;
    if (part) inf |= INFO_PART;
    if (!gn_mask || gn_mask[idx] >= thresh) inf |= INFO_OWNMASK;
    info[idx] = inf;
}

// rim bit (tilek::RES_* numbering) of the neighbouring tile that holds the cell (nx2, ny2), which lies outside this tile
template <int TS>
__device__ __forceinline__ int rim_bit(int nx2, int ny2) {
    return ny2 < 0 ? (nx2 < 0 ? 16 : (nx2 >= TS ? 32 : 1)) : (ny2 >= TS ? (nx2 < 0 ? 64 : (nx2 >= TS ? 128 : 2)) : (nx2 < 0 ? 4 : 8));
}

// the same for a whole set of neighbours (bit k - 1 of `m`: neighbour k), without a loop or a branch: a loop over the set bits with d1 / d2 / rim_bit per
// neighbour was ~40 instructions per bit, run by every wave (each holds cells of the tile's first and last column) for the largest set in the wave.
// Neighbours to the north are k = 2, 3, 4, south 6, 7, 8, west 4, 5, 6, east 1, 2, 8.
template <int TS>
__device__ __forceinline__ int rim_bits_of_mask(unsigned m, int lx, int ly) {
    const unsigned mN = ly == 0 ? m & 0x0Eu : 0u, mS = ly == TS - 1 ? m & 0xE0u : 0u, mW = lx == 0 ? m & 0x38u : 0u, mE = lx == TS - 1 ? m & 0x83u : 0u;
    const unsigned ns = mN | mS, we = mW | mE;
    return ((mN & ~we) ? 1 : 0) | ((mS & ~we) ? 2 : 0) | ((mW & ~ns) ? 4 : 0) | ((mE & ~ns) ? 8 : 0) | ((mN & mW) ? 16 : 0) | ((mN & mE) ? 32 : 0) | ((mS & mW) ? 64 : 0) |
           ((mS & mE) ? 128 : 0);
}

// ---- value policies -------------------------------------------------------------------------------------------------------
// A policy defines the per-cell record (`Cell`) that lives in the work array and, tile + ring, in LDS.  Its first 32 bits carry
// the pending pattern.  A record is read and written with ONE load / store instruction, so a tile that stages its ring while the
// neighbouring tile writes back (same round) sees either the old record (pending) or the complete new one - never a mixture;
// that is why GridNet's three results travel as one 16-byte record and are split into the three rasters afterwards.
struct SumMaxMin {   // AreaD8 with weights, D8FlowPathExtremeUp (aread8.hip: D8Expr)
    using Cell = float;
    using Aux = float;                          // weight grid / the grid whose extreme is sought (may be absent for unit weights)
    static constexpr bool HAS_AUX = true;
    static constexpr bool HAS_DIST = false;
    static constexpr bool HAS_ROWS = false;
    static constexpr int kBulkSweeps = BULK_SWEEPS_D8;
    static constexpr unsigned kBulkUntil = 16;     // rounds run on 32 x 32 tiles until this few are active (measured at 16384^2: 6000 -> 16 is 2-3 % faster for every forward tool)
    static constexpr int kMinWaves32 = 4;
    static constexpr int kMaxRelease = 1;          // cells a finished cell can release (<= 2: their in-tile indices are made once per activation, see Lds::tw)
    // cells whose pending count includes this one: the cell it drains to
    static __device__ __forceinline__ unsigned rel_mask(unsigned inf) { const unsigned code = (inf >> 9) & 15u; return (code >= 1u && code <= 8u) ? 1u << (code - 1u) : 0u; }
    int mode;            // 0 sum, 1 max, 2 min
    float out_nodata;
    float w_nodata;
    int contcheck;
    bool has_aux;
    static __device__ __forceinline__ float head(float c) { return c; }
    static __host__ __device__ __forceinline__ float outside() { return -1.0f; }
    template <class L>
    __device__ __forceinline__ void eval(L& S, int c, int cl, int ly, unsigned inf, const Cell (&nb)[9]) const {
        float a;
        const float aux = has_aux ? S.aux[c] : 1.0f;
        if (mode != 0) a = aux;                                                     // src/D8flowpathextremeup.cpp:170
        else if (has_aux) a = is_nodata_f(aux, w_nodata) ? TDX_AREA_NODATA : aux;   // a nodata weight keeps the initial -1
        else a = 1.0f;
        bool con = (inf & INFO_CON) != 0u;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (!((inf >> (16 + k - 1)) & 1u)) continue;
            const float v = nb[k];
            if (is_nodata_f(v, out_nodata)) con = true;
            else if (mode == 0) a = a + v;
            else if (mode == 1) { if (v > a) a = v; }
            else { if (v < a) a = v; }
        }
        if (con && contcheck == 1) a = out_nodata;
        S.v[cl] = a;
    }
};

struct GridNetAlg {   // src/gridnet.cpp:380-426; record = {plen, tlen, gord (int bits), -}
    using Cell = float4;
    using Aux = float;
    static constexpr bool HAS_AUX = false;
    static constexpr bool HAS_DIST = true;
    static constexpr bool HAS_ROWS = false;
    static constexpr int kBulkSweeps = BULK_SWEEPS_D8;
    static constexpr unsigned kBulkUntil = 16;
    static constexpr int kMinWaves32 = 4;
    static constexpr int kMaxRelease = 1;
    static __device__ __forceinline__ unsigned rel_mask(unsigned inf) { const unsigned code = (inf >> 9) & 15u; return (code >= 1u && code <= 8u) ? 1u << (code - 1u) : 0u; }
    static __device__ __forceinline__ float head(const float4& c) { return c.x; }
    static __host__ __device__ __forceinline__ float4 outside() { const int m1 = -1; float z; memcpy(&z, &m1, 4); return make_float4(-1.0f, -1.0f, z, 0.f); }
    template <class L>
    __device__ __forceinline__ void eval(L& S, int c, int cl, int ly, unsigned inf, const Cell (&nb)[9]) const {
        float4 me = S.v[cl];
        if (!(inf & INFO_OWNMASK)) { me.x = -1.0f; S.v[cl] = me; return; }   // completes without a value: tlen / gord keep what they had
        float tl = 0.0f, pl = 0.0f;
        int a1 = 0, a2 = 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (!((inf >> (16 + k - 1)) & 1u)) continue;
            const int g = __float_as_int(nb[k].z);        // Strahler order (src/gridnet.cpp:404-411)
            if (g >= a1) { a2 = a1; a1 = g; }
            else if (g > a2) a2 = g;
            const float dd = S.dist[ly * 9 + ((k + 3) % 8 + 1)];   // dist[j][sdir]: the row of the evaluated cell, the CODE of the neighbour (= k +- 4)
            const float ld = nb[k].x + dd;
            tl = tl + (float)(nb[k].y + dd);
            if (ld > pl) pl = ld;
        }
        S.v[cl] = make_float4(pl, tl, __int_as_float((a2 + 1 > a1) ? a2 + 1 : a1), 0.f);
    }
};

constexpr int QFWD = 128;   // forward policies: only a fork whose branches become ready at once uses the queue
template <class Alg, int TSZ>
struct Lds {
    static constexpr int TS = Dim<TSZ>::TS, LH = Dim<TSZ>::LH;
    static constexpr int QLEN = Alg::kMaxRelease <= 2 ? QFWD : 1;   // (reverse policies run sweep_tile_rev: no walks, no queue; their Lds serves the verifier)
    typename Alg::Cell v[LH * LH];
    typename Alg::Aux aux[Alg::HAS_AUX ? TS * TS : 1];
    float dist[Alg::HAS_DIST ? TS * 9 : 1];
    double rows[Alg::HAS_ROWS ? LH : 1];   // per-row value of the tile's rows and the ring rows, window row ly + 1 (D-infinity: a2 = atan2(dy, dx))
    uint32_t info[TS * TS];
    uint32_t cnt[TS * TS / 4];   // one byte per cell: contributors still pending (255: not a pending cell of this rank)
    // forward policies (a finished cell releases one or two cells): the cells it releases inside the tile, made once per activation - index t0 [0:12)
    // t1 [12:24), "is in the tile" bits 24 / 25, "outside the tile" bits 26 / 27 - so that a hop of the walks has no direction arithmetic and no branch
    // around its decrements (a cell that is not there decrements the lane's spare word)
    uint32_t tw[Alg::kMaxRelease <= 2 ? TS * TS : 1];
    uint32_t spare[Alg::kMaxRelease <= 2 ? 64 : 1];
    uint16_t q[2][QLEN];   // ready cells handed on to the next phase (a finished cell may release several)
    unsigned nq[2];
    int rim;
    int over;                    // the queue overflowed: the tile runs again (ready cells are re-discovered from the values)
    int wrote;                   // some lane wrote a result back in this activation
    int vote[3];                 // "did anybody evaluate a cell in this sweep": one word per sweep, three in rotation (clearing one never meets its setters or readers)
};

template <class Alg>
struct Arrays {   // global arrays of one sweep
    typename Alg::Cell* v;                  // the work array (AreaD8 family: the result raster itself)
    const typename Alg::Aux* aux;           // per-cell input record (weights, angle ...; may be null)
    const float* dist;                      // GridNet: [row][9]
    const double* rows;                     // per array row (may be null)
    const uint32_t* info;
};

// Stages tile + ring of the work array, the tile's info words and the policy's per-cell / per-row inputs in LDS: every load is issued
// before the first LDS store (addresses clamped, validity applied afterwards).  The caller synchronises.
template <class Alg, int TSZ>
__device__ __forceinline__ void stage_tile(const tilek::TileGeom& g, int tile, Lds<Alg, TSZ>& S, const Arrays<Alg>& A) {
    using Cell = typename Alg::Cell;
    constexpr int TS = Dim<TSZ>::TS, LH = Dim<TSZ>::LH, NT = Dim<TSZ>::NT, RPL = Dim<TSZ>::RPL, NSTAGE = Dim<TSZ>::NSTAGE;
    const int tid = threadIdx.x, lx = tid % TS, ry0 = (tid / TS) * RPL;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    const int x0 = tx * TS, y0 = ty * TS;
    {
        Cell s0[NSTAGE];
        uint32_t si[RPL];
        typename Alg::Aux sa[Alg::HAS_AUX ? RPL : 1];
        unsigned ok = 0;
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) {
            const int e = tid + i * NT, ec = e < LH * LH ? e : LH * LH - 1;
            const int wy = ec / LH, wx = ec - wy * LH;
            const int gx = x0 - 1 + wx, gy = y0 - 1 + wy;
            if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny) ok |= 1u << i;
            const int gxc = gx < 0 ? 0 : (gx >= g.nx ? g.nx - 1 : gx), gyc = gy < 0 ? 0 : (gy >= g.ny ? g.ny - 1 : gy);
            s0[i] = A.v[size_t(gyc) * size_t(g.nx) + size_t(gxc)];
        }
        unsigned oki = 0;
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            const int gx = x0 + lx, gy = y0 + ry0 + r;
            if (gx < g.nx && gy < g.ny) oki |= 1u << r;
            const size_t idx = size_t(gy >= g.ny ? g.ny - 1 : gy) * size_t(g.nx) + size_t(gx >= g.nx ? g.nx - 1 : gx);
            si[r] = A.info[idx];
            if (Alg::HAS_AUX) { if (A.aux) sa[r] = A.aux[idx]; else sa[r] = typename Alg::Aux{}; }
        }
        double srow = 0.;
        if (Alg::HAS_ROWS && tid < LH) { const int gy = y0 - 1 + tid; srow = A.rows[gy < 0 ? 0 : (gy >= g.ny ? g.ny - 1 : gy)]; }
        constexpr int NDIST = (TS * 9 + NT - 1) / NT;   // (32 x 32 tiles: 288 table entries for 256 threads)
        float sdist[NDIST];
#pragma unroll
        for (int i = 0; i < NDIST; i++) {
            const int e = tid + i * NT, ec = e < TS * 9 ? e : TS * 9 - 1;
            const int gy = y0 + ec / 9;
            sdist[i] = Alg::HAS_DIST ? A.dist[size_t(gy >= g.ny ? g.ny - 1 : gy) * 9 + size_t(ec % 9)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) {
            const int e = tid + i * NT;
            if (e < LH * LH) S.v[e] = ((ok >> i) & 1u) ? s0[i] : Alg::outside();   // (never read: cells outside the raster are nobody's contributor)
        }
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            const unsigned inf = ((oki >> r) & 1u) ? si[r] : 0u;
            S.info[(ry0 + r) * TS + lx] = inf;
            if (Alg::HAS_AUX) S.aux[(ry0 + r) * TS + lx] = sa[r];
            if constexpr (Alg::kMaxRelease <= 2) {
                unsigned tw = 0u, m = Alg::rel_mask(inf);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int k = m ? __ffs(int(m)) : 0;
                    m &= m - 1u;
                    if (!k) continue;
                    const int nx2 = lx + d1(k), ny2 = ry0 + r + d2(k);
                    if (nx2 < 0 || nx2 >= TS || ny2 < 0 || ny2 >= TS) tw |= 1u << (26 + t);
                    else tw |= (unsigned(ny2 * TS + nx2) << (12 * t)) | (1u << (24 + t));
                }
                S.tw[(ry0 + r) * TS + lx] = tw;
            }
        }
        if (Alg::HAS_DIST) {
#pragma unroll
            for (int i = 0; i < NDIST; i++) if (tid + i * NT < TS * 9) S.dist[tid + i * NT] = sdist[i];
        }
        if (Alg::HAS_ROWS && tid < LH) S.rows[tid] = srow;
    }
}

#ifdef TDX_REV_CLOCKS
static __device__ unsigned long long g_rev_clk[40];   // phase clocks of the tile routines (a build-time probe: -DTDX_REV_CLOCKS on one object file)
#endif
template <class Alg, int TSZ>
__device__ __forceinline__ int sweep_tile(const Alg& alg, const tilek::TileGeom& g, int tile, bool full, Lds<Alg, TSZ>& S, const Arrays<Alg>& A, bool clk_on = true) {
#ifdef TDX_REV_CLOCKS
    const unsigned long long fc0 = wall_clock64();
    unsigned long long fc1 = 0, fc2 = 0, fc3 = 0, fc4 = 0;
#endif
    using Cell = typename Alg::Cell;
    constexpr int TS = Dim<TSZ>::TS, LH = Dim<TSZ>::LH, NT = Dim<TSZ>::NT, RPL = Dim<TSZ>::RPL;
    const int tid = threadIdx.x, lx = tid % TS, ry0 = (tid / TS) * RPL;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    const int x0 = tx * TS, y0 = ty * TS;
    if (tid == 0) { S.rim = 0; S.over = 0; S.nq[0] = 0u; S.nq[1] = 0u; S.wrote = 0; S.vote[0] = 0; S.vote[1] = 0; S.vote[2] = 0; }
    stage_tile<Alg, TSZ>(g, tile, S, A);
    __syncthreads();
#ifdef TDX_REV_CLOCKS
    fc1 = wall_clock64();
#endif
    unsigned pendmask = 0;   // own cells that are pending (participating, owned by this rank, not evaluated yet) and can become ready
#pragma unroll
    for (int r = 0; r < RPL; r++) {
        const int gx = x0 + lx, gy = y0 + ry0 + r;
        if (gx < g.nx && gy >= g.y_own0 && gy < g.y_own1 && pending(Alg::head(S.v[(ry0 + r + 1) * LH + lx + 1])) && !(S.info[(ry0 + r) * TS + lx] & INFO_DEAD))
            pendmask |= 1u << r;
    }
    const unsigned pend0 = pendmask;
    int rim = 0;
    auto out_of_tile = [&](unsigned inf, int cx, int ly) {   // the tiles a finished cell releases cells in have to look again
        for (unsigned m = Alg::rel_mask(inf); m; m &= m - 1u) {
            const int k = __ffs(int(m));
            const int nx2 = cx + d1(k), ny2 = ly + d2(k);
            if (nx2 < 0 || nx2 >= TS || ny2 < 0 || ny2 >= TS) rim |= rim_bit<TS>(nx2, ny2);
        }
    };
    auto load_nbrs = [&](int cl, Cell (&nb)[9]) {   // unconditional and together: one LDS latency for the whole neighbourhood
#pragma unroll
        for (int k = 1; k <= 8; k++) nb[k] = S.v[cl + d2(k) * LH + d1(k)];
    };
    auto pending_bits = [&](const Cell (&nb)[9]) {
        unsigned pb = 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) pb |= pending(Alg::head(nb[k])) ? 1u << (k - 1) : 0u;
        return pb;
    };
    // ---- bulk: lockstep sweeps over the lane's own cells (ready = no contributor pending), no atomics.  Forward sweeps (one or two
    // receivers per cell: long thin chains) run a few of them on a fresh tile (the policy's kBulkSweeps) and leave the stream cells to
    // the walks.  (The REVERSE sweeps - a finished cell releases up to eight senders - have a routine of their own: sweep_tile_rev.)
    if (full) {
        for (int sweep = 0; sweep < g.max_sweeps; sweep++) {   // (TileGeom::max_sweeps carries the policy's kBulkSweeps here; TDX_D8_BULK_SWEEPS overrides it - A/B hook)
            bool prog = false;
#pragma unroll
            for (int rr = 0; rr < RPL; rr++) {
                const int r = (sweep & 1) ? RPL - 1 - rr : rr;
                if (!((pendmask >> r) & 1u)) continue;
                const int ly = ry0 + r, c = ly * TS + lx, cl = (ly + 1) * LH + lx + 1;
                const unsigned inf = S.info[c];
                Cell nb[9];
                load_nbrs(cl, nb);
                if ((inf & 0xFFu & pending_bits(nb)) == 0u) {
                    alg.eval(S, c, cl, ly, inf, nb);
                    out_of_tile(inf, lx, ly);
                    pendmask &= ~(1u << r);
                    prog = true;
                }
            }
            // the vote through an LDS word and a barrier that waits for LDS only (__syncthreads_or: a reduction, two full barriers and a wait for every memory operation)
            const int slot = sweep % 3;
            if (prog) S.vote[slot] = 1;
            tdx_barrier_lds();
            const int any = S.vote[slot];
            if (tid == 0) S.vote[(slot + 2) % 3] = 0;
            if (!any) break;
        }
    }
#ifdef TDX_REV_CLOCKS
    fc2 = wall_clock64();
#endif
    // ---- pending contributors of the cells that are left
    unsigned readymask = 0;
    uint8_t* cnt8 = reinterpret_cast<uint8_t*>(S.cnt);
#pragma unroll
    for (int r = 0; r < RPL; r++) {
        const int ly = ry0 + r, c = ly * TS + lx, cl = (ly + 1) * LH + lx + 1;
        unsigned cn = 255u;
        const unsigned inf = S.info[c];
        Cell nb[9];
        load_nbrs(cl, nb);
        if ((pendmask >> r) & 1u) {
            cn = unsigned(__popc(inf & 0xFFu & pending_bits(nb)));
            if (cn == 0u) readymask |= 1u << r;
        }
        cnt8[c] = uint8_t(cn);
    }
    __syncthreads();
    // ---- walks: a lane follows a chain as long as it finishes the last pending contributor of a released cell; further cells
    // released by the same step go to the hand-over queue, which the workgroup drains in phases
#ifdef TDX_REV_CLOCKS
    fc3 = wall_clock64();
#endif
    auto walk_fwd = [&](int c, int phase) {   // the released cells (at most two) come from the tile's target words
        unsigned inf = S.info[c], tw = S.tw[c];
        uint32_t* const spare = &S.spare[tid & 63];
        for (;;) {
            const int ly = c / TS, cx = c % TS, cl = (ly + 1) * LH + cx + 1;
            Cell nb[9];
            load_nbrs(cl, nb);
            alg.eval(S, c, cl, ly, inf, nb);
            if (tw >> 26) out_of_tile(inf, cx, ly);   // releases a cell of a neighbouring tile (cells on the tile's rim only)
            const int t0 = int(tw & 0xFFFu), t1 = int((tw >> 12) & 0xFFFu);
            const bool ok0 = ((tw >> 24) & 1u) != 0u, ok1 = Alg::kMaxRelease > 1 && ((tw >> 25) & 1u) != 0u;
            const unsigned sh0 = 8u * unsigned(t0 & 3), sh1 = 8u * unsigned(t1 & 3);
            const unsigned old0 = __hip_atomic_fetch_sub(ok0 ? &S.cnt[t0 >> 2] : spare, 1u << sh0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            unsigned old1 = 0u, inf1 = 0u, tw1 = 0u;
            if (Alg::kMaxRelease > 1) old1 = __hip_atomic_fetch_sub(ok1 ? &S.cnt[t1 >> 2] : spare, 1u << sh1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned inf0 = S.info[t0], tw0 = S.tw[t0];
            if (Alg::kMaxRelease > 1) { inf1 = S.info[t1]; tw1 = S.tw[t1]; }
            const bool last0 = ok0 && ((old0 >> sh0) & 255u) == 1u, last1 = ok1 && ((old1 >> sh1) & 255u) == 1u;   // its last pending contributor
            if (last0 && last1) {   // a fork with both branches ready: the second one goes to the queue of the next phase
                const unsigned slot = atomicAdd(&S.nq[phase ^ 1], 1u);
                if (slot < unsigned(Lds<Alg, TSZ>::QLEN)) S.q[phase ^ 1][slot] = uint16_t(t1);
                else S.over = 1;
            }
            if (!(last0 || last1)) break;
            c = last0 ? t0 : t1;
            inf = last0 ? inf0 : inf1;
            tw = last0 ? tw0 : tw1;
        }
    };
    auto walk = [&](int c, int phase) { walk_fwd(c, phase); };
    for (unsigned m = readymask; m; m &= m - 1u) walk((ry0 + (__ffs(int(m)) - 1)) * TS + lx, 0);
    for (int phase = 1;; phase ^= 1) {
        __syncthreads();                        // every push into q[phase] has landed
        const unsigned n = S.nq[phase] < unsigned(Lds<Alg, TSZ>::QLEN) ? S.nq[phase] : unsigned(Lds<Alg, TSZ>::QLEN);
        if (n == 0u) break;
        for (unsigned i = tid; i < n; i += unsigned(NT)) walk(int(S.q[phase][i]), phase);
        __syncthreads();                        // q[phase] has been read by everybody
        if (tid == 0) S.nq[phase] = 0u;         // (nobody pushes into it before the next barrier)
    }
#ifdef TDX_REV_CLOCKS
    fc4 = wall_clock64();
#endif
    // ---- write back what this activation evaluated (one store per record)
    bool wrote = false;
    for (unsigned m = pend0; m; m &= m - 1u) {
        const int r = __ffs(int(m)) - 1, ly = ry0 + r, cl = (ly + 1) * LH + lx + 1;
        const Cell v = S.v[cl];
        if (!pending(Alg::head(v))) { A.v[size_t(y0 + ly) * size_t(g.nx) + size_t(x0 + lx)] = v; wrote = true; }
    }
    if (rim) atomicOr(&S.rim, rim);
    // "did anybody write" through LDS and a barrier that waits for LDS only: __syncthreads_or() sits out the acknowledgement of every global store above, and
    // nobody in this launch depends on them having landed (see sweep_tile_rev and dinf_sweep_tile.inc); the workgroup stages its next tile meanwhile
    if (wrote) S.wrote = 1;
    tdx_barrier_lds();
    const int res = (S.wrote ? (tilek::RES_CHANGED | S.rim) : 0) | (S.over ? tilek::RES_CAPPED : 0);
    tdx_barrier_lds();   // S is reused by the next tile
#ifdef TDX_REV_CLOCKS
    if (tid == 0 && clk_on) {
        const unsigned long long fc5 = wall_clock64();
        atomicAdd(&g_rev_clk[0], fc1 - fc0); atomicAdd(&g_rev_clk[1], fc2 - fc1); atomicAdd(&g_rev_clk[2], fc3 - fc2); atomicAdd(&g_rev_clk[3], fc4 - fc3);
        atomicAdd(&g_rev_clk[4], fc5 - fc4); atomicAdd(&g_rev_clk[5], 1ull); atomicAdd(&g_rev_clk[6], full ? 1ull : 0ull);
    }
#endif
    return res;
}

// ---- reverse policies (kMaxRelease > 2: DinfUpDependence, DinfRevAccum) -------------------------------------------------------
// A cell waits for its at most TWO receivers and a finished cell releases up to eight senders: a wide front, for which lockstep sweeps
// to the end beat the walks (docs/experiments_r02_r04.md).  Measured in round 6 (profiles/r06o_reverse_round_times.txt): a round took
// ~280 us whether 130 or 5 000 tiles were active - the time of the slowest tile, up to ~48 sweeps of >= 5 us each, because every sweep
// re-read the info word and all eight neighbour records of each pending cell and every evaluation recomputed the two fp64 proportions
// (two divisions + the sector table) inside a divergent branch.  Here everything a pending cell needs is made ONCE per activation and
// kept in registers - the window offsets of its two receivers, whether each counts, the two proportions, its own input record - so a
// sweep is two LDS loads per pending cell (all issued before the first use), two pattern compares and, for a ready cell, a multiply-add
// per receiver.  Only the work array goes through LDS.  The policy supplies
//   rev_row(inf, aux, a2, k, on, p)   receivers in ascending k (the reference's order), which of them count, their proportions
//   eval2(aux, on, p, n) -> Cell      the value from the receivers' records
// and keeps eval() for the verifier, which re-evaluates every cell from the final records with the original expression.
template <class Alg, int TSZ>
struct LdsRev {
    static constexpr int LH = Dim<TSZ>::LH;
    typename Alg::Cell v[LH * LH];
    int rim;
    int wrote;
    int vote[3];
};

template <class Alg, int TSZ>
__device__ __forceinline__ int sweep_tile_rev(const Alg& alg, const tilek::TileGeom& g, int tile, LdsRev<Alg, TSZ>& S, const Arrays<Alg>& A, bool clk_on = true) {
    using Cell = typename Alg::Cell;
    using Aux = typename Alg::Aux;
    constexpr int TS = Dim<TSZ>::TS, LH = Dim<TSZ>::LH, NT = Dim<TSZ>::NT, RPL = Dim<TSZ>::RPL, NSTAGE = Dim<TSZ>::NSTAGE;
    const int tid = threadIdx.x, lx = tid % TS, ry0 = (tid / TS) * RPL;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    const int x0 = tx * TS, y0 = ty * TS;
    if (tid == 0) { S.rim = 0; S.wrote = 0; S.vote[0] = 0; S.vote[1] = 0; S.vote[2] = 0; }
#ifdef TDX_REV_CLOCKS
    const unsigned long long tc0 = wall_clock64();
    unsigned long long tc1 = 0, tc2 = 0, tc3 = 0, nstretch = 0;
#endif
    uint32_t si[RPL];
    Aux sa[RPL];
    double srow[RPL];
    unsigned oki = 0;
    {   // every load before the first LDS store; addresses clamped, validity applied afterwards
        Cell s0[NSTAGE];
        unsigned ok = 0;
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) {
            const int e = tid + i * NT, ec = e < LH * LH ? e : LH * LH - 1;
            const int wy = ec / LH, wx = ec - wy * LH;
            const int gx = x0 - 1 + wx, gy = y0 - 1 + wy;
            if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny) ok |= 1u << i;
            const int gxc = gx < 0 ? 0 : (gx >= g.nx ? g.nx - 1 : gx), gyc = gy < 0 ? 0 : (gy >= g.ny ? g.ny - 1 : gy);
            s0[i] = A.v[size_t(gyc) * size_t(g.nx) + size_t(gxc)];
        }
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            const int gx = x0 + lx, gy = y0 + ry0 + r;
            if (gx < g.nx && gy < g.ny) oki |= 1u << r;
            const int gyc = gy >= g.ny ? g.ny - 1 : gy;
            const size_t idx = size_t(gyc) * size_t(g.nx) + size_t(gx >= g.nx ? g.nx - 1 : gx);
            si[r] = A.info[idx];
            sa[r] = A.aux[idx];
            srow[r] = A.rows[gyc];
        }
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) {
            const int e = tid + i * NT;
            if (e < LH * LH) S.v[e] = ((ok >> i) & 1u) ? s0[i] : Alg::outside();
        }
    }
    __syncthreads();
#ifdef TDX_REV_CLOCKS
    tc1 = wall_clock64();
#endif
    unsigned pendmask = 0, rimcell = 0;
    int off0[RPL], off1[RPL];
    bool on0[RPL], on1[RPL];
    double p0[RPL], p1[RPL];
#pragma unroll
    for (int r = 0; r < RPL; r++) {
        const int ly = ry0 + r, cl = (ly + 1) * LH + lx + 1;
        const int gx = x0 + lx, gy = y0 + ly;
        const unsigned inf = ((oki >> r) & 1u) ? si[r] : 0u;
        off0[r] = cl; off1[r] = cl; on0[r] = false; on1[r] = false; p0[r] = 0.; p1[r] = 0.;
        if (gx < g.nx && gy >= g.y_own0 && gy < g.y_own1 && pending(Alg::head(S.v[cl])) && !(inf & INFO_DEAD)) {
            pendmask |= 1u << r;
            int k[2];
            bool on[2];
            double p[2];
            Alg::rev_row(inf, sa[r], srow[r], k, on, p);
            on0[r] = on[0]; on1[r] = on[1]; p0[r] = p[0]; p1[r] = p[1];
            if (on[0]) off0[r] = cl + d2(k[0]) * LH + d1(k[0]);
            if (on[1]) off1[r] = cl + d2(k[1]) * LH + d1(k[1]);
            // the senders a finished cell releases in a neighbouring tile: that tile looks again
            const int rb = rim_bits_of_mask<TS>(Alg::rel_mask(inf), lx, ly);
            rimcell |= unsigned(rb) << (8 * r);
        }
    }
    static_assert(RPL <= 4, "rim bits of a lane's cells are packed one byte per row");
    const unsigned pend0 = pendmask;
    int rim = 0;
    // The waves sweep their own cells WITHOUT a barrier between sweeps: a receiver's record read meanwhile is the pending pattern or the
    // complete result (one LDS store), and pending -> result is the only transition, so a sweep can only be late, never wrong.  A vote
    // every REV_INNER sweeps ends the activation: nobody evaluated a cell in a whole stretch, and every wave made at least one full sweep
    // after the barrier that published the last result.  (A dependency chain inside one wave's rows advances at LDS latency; the vote is a
    // flag word per stretch - three, so that clearing one never meets its setters or its readers - and a barrier that waits for LDS only.)
    // REV_INNER: the last stretch of an activation finds nothing and is wasted, so it is short (measured at 16384^2, DinfUpDependence / DinfRevAccum:
    // 2: 93.8 / 85.4 ms, 4: 93.5 / 84.6, 8: 98.0 / 86.9, 16: 111 / 99.5, 32: 137 / 112; a vote per sweep, the first form: 115 / 101).
    constexpr int REV_INNER = 4;
#ifdef TDX_REV_CLOCKS
    tc2 = wall_clock64();
#endif
    for (int stretch = 0;; stretch++) {
#ifdef TDX_REV_CLOCKS
        nstretch++;
#endif
        bool prog = false;
#pragma unroll 1
        for (int it = 0; it < REV_INNER; it++) {
            if (!__any(pendmask != 0u)) break;
            asm volatile("" ::: "memory");   // the records are read again in every sweep
            Cell n0[RPL], n1[RPL];
#pragma unroll
            for (int r = 0; r < RPL; r++) { n0[r] = S.v[off0[r]]; n1[r] = S.v[off1[r]]; }
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                if (!((pendmask >> r) & 1u)) continue;
                if ((on0[r] && pending(Alg::head(n0[r]))) || (on1[r] && pending(Alg::head(n1[r])))) continue;
                const bool on[2] = {on0[r], on1[r]};
                const double p[2] = {p0[r], p1[r]};
                const Cell n[2] = {n0[r], n1[r]};
                S.v[(ry0 + r + 1) * LH + lx + 1] = alg.eval2(sa[r], on, p, n);
                rim |= int((rimcell >> (8 * r)) & 255u);
                pendmask &= ~(1u << r);
                prog = true;
            }
        }
        const int slot = stretch % 3;
        if (prog) S.vote[slot] = 1;
        tdx_barrier_lds();
        const int any = S.vote[slot];
        if (tid == 0) S.vote[(slot + 2) % 3] = 0;
        if (!any) break;
    }
#ifdef TDX_REV_CLOCKS
    tc3 = wall_clock64();
#endif
    // write back what this activation evaluated (one store per record).  "Did anybody write" goes through LDS and a barrier that waits for LDS only:
    // __syncthreads_or() would sit out the acknowledgement of every store above, and nobody in this launch depends on them having landed (a neighbour that
    // stages its ring meanwhile sees the old or the new record; the next round is another launch; the solo hand-over of round_driver fences for itself).
    bool wrote = false;
    for (unsigned m = pend0 & ~pendmask; m; m &= m - 1u) {
        const int r = __ffs(int(m)) - 1, ly = ry0 + r;
        A.v[size_t(y0 + ly) * size_t(g.nx) + size_t(x0 + lx)] = S.v[(ly + 1) * LH + lx + 1];
        wrote = true;
    }
    if (rim) atomicOr(&S.rim, rim);
    if (wrote) S.wrote = 1;
    tdx_barrier_lds();
    const int res = S.wrote ? (tilek::RES_CHANGED | S.rim) : 0;
    tdx_barrier_lds();   // S is reused by the next tile
#ifdef TDX_REV_CLOCKS
    if (tid == 0 && clk_on) {
        const unsigned long long tc4 = wall_clock64();
        atomicAdd(&g_rev_clk[0], tc1 - tc0); atomicAdd(&g_rev_clk[1], tc2 - tc1); atomicAdd(&g_rev_clk[2], tc3 - tc2); atomicAdd(&g_rev_clk[3], tc4 - tc3);
        atomicAdd(&g_rev_clk[4], nstretch); atomicAdd(&g_rev_clk[5], 1ull); atomicMax(&g_rev_clk[6], tc3 - tc2); atomicMax(&g_rev_clk[7], nstretch);
        { const unsigned long long us = (tc3 - tc2) / 100; atomicAdd(&g_rev_clk[8 + (us < 15 ? us : 15)], 1ull); atomicAdd(&g_rev_clk[24 + (nstretch < 15 ? nstretch : 15)], 1ull); }
    }
#endif
    return res;
}

// MINW: waves per SIMD the register allocation is held to (= 256-thread tiles per CU of the 32 x 32 geometry; policy constant kMinWaves32):
// 4 is what the kernels take by themselves (107-117 VGPRs); 5 (<= 102 VGPRs) puts a fifth tile on a CU where the LDS allows it - measured
// per policy at 16384^2 (profiles/r03k_*): GridNet 64.0 -> 60.0 ms (3 spilled registers; with the target words of round 4 eleven, and 4 is as fast: 59 ms), DinfUpDependence 381 -> 376 ms (2), DinfRevAccum
// slower (10 spills), the limited accumulations are LDS-bound at 4 tiles, weighted AreaD8 / ExtremeUp need 88 VGPRs anyway
template <class Alg, int TSZ, int MINW = 4>
__global__ __launch_bounds__(Dim<TSZ>::NT, MINW) void sweep_kernel(Alg alg, tilek::TileGeom g, const uint32_t* __restrict__ list, unsigned long long* __restrict__ count,
                                                   uint32_t* __restrict__ flags_cur, uint32_t* __restrict__ flags_next, uint32_t* __restrict__ list_next,
                                                   unsigned pull_max, Arrays<Alg> A) {
    __shared__ tilek::TileLds L;
    if constexpr (Alg::kMaxRelease > 2) {
        __shared__ LdsRev<Alg, TSZ> S;
#ifdef TDX_REV_CLOCKS
        const bool clk_on = unsigned(count[0]) <= 1200u;
        tilek::round_driver(list, count, flags_cur, flags_next, list_next, pull_max, g, L, [&](int tile, bool) { return sweep_tile_rev<Alg, TSZ>(alg, g, tile, S, A, clk_on); });
#else
        tilek::round_driver(list, count, flags_cur, flags_next, list_next, pull_max, g, L, [&](int tile, bool) { return sweep_tile_rev<Alg, TSZ>(alg, g, tile, S, A); });
#endif
    } else {
        __shared__ Lds<Alg, TSZ> S;
#ifdef TDX_REV_CLOCKS
        const bool clk_on = unsigned(count[0]) >= 50000u;   // the bulk rounds
        tilek::round_driver(list, count, flags_cur, flags_next, list_next, pull_max, g, L, [&](int tile, bool full) { return sweep_tile<Alg, TSZ>(alg, g, tile, full, S, A, clk_on); });
#else
        tilek::round_driver(list, count, flags_cur, flags_next, list_next, pull_max, g, L, [&](int tile, bool full) { return sweep_tile<Alg, TSZ>(alg, g, tile, full, S, A); });
#endif
    }
}

// the pending pattern that is left (cells on or below a cycle, cells fed by the p == 0 quirk) becomes `value`
static __global__ __launch_bounds__(256) void finish_kernel(float* __restrict__ v0, size_t first, size_t n, float value) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < first + n && pending(v0[i])) v0[i] = value;
}

// initial state of the primary result array on the owned rows: pending where the cell participates, `other` elsewhere
static __global__ __launch_bounds__(256) void init_kernel(const uint32_t* __restrict__ info, float* __restrict__ v0, size_t first, size_t n, float other) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < first + n) v0[i] = (info[i] & INFO_PART) ? __uint_as_float(PENDING_BITS) : other;
}

// ---- upstream closure of outlet cells (outlets mode) through the tile relaxation engine (flats.hpp: reach_closure) ----
// mask of the reachability relaxation: the neighbour a cell drains to; a p == 0 cell counts as draining to its south-east
// neighbour, because the in-degree counts it there (src/commonLib.cpp:257-266, src/gridnet.cpp:285-300)
static __global__ __launch_bounds__(256) void reach_mask_kernel(const int16_t* __restrict__ P, size_t n, int16_t nodata, uint8_t* __restrict__ mask) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const int16_t p = P[i];
    unsigned m = 0;
    if (!is_nodata_s(p, nodata)) {
        if (p >= 1 && p <= 8) m = 1u << (p - 1);
        else if (p == 0) m = 1u << 7;
    }
    mask[i] = uint8_t(m);
}
// outlet cells (array coordinates; only those in the owned rows, and with skip_nodata only those on a cell with a direction
// value): reach = 1 and their tile is activated
static __global__ __launch_bounds__(256) void reach_seed_kernel(const int32_t* __restrict__ ox, const int32_t* __restrict__ oy, int nout, int nx, int ny, int y_own0,
                                                                int y_own1, int tiles_x, const int16_t* __restrict__ P, int16_t nodata, int skip_nodata,
                                                                int32_t* __restrict__ reach, uint32_t* __restrict__ tile_flags) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    const int x = ox[o], y = oy[o];
    if (x < 0 || x >= nx || y < y_own0 || y >= y_own1) return;   // globalToLocal + isInPartition (src/commonLib.cpp:289-291)
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    if (skip_nodata && is_nodata_s(P[idx], nodata)) return;
    reach[idx] = 1;
    tilek::activate_tiles_around(x, y, nx, ny, tiles_x, tile_flags);
}

template <int BYTES> struct BitsOf;
template <> struct BitsOf<4> { using type = uint32_t; };
template <> struct BitsOf<8> { using type = uint2; };
template <> struct BitsOf<16> { using type = uint4; };

// activation flags between the two tile geometries (a 64 x 64 tile = 2 x 2 tiles of 32 x 32)
static __global__ __launch_bounds__(256) void flags_down_kernel(const uint32_t* __restrict__ f64, int tiles_x64, uint32_t* __restrict__ f32, int tiles_x32,
                                                                int tiles_y32) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles_x32 * tiles_y32) return;
    const int tx = t % tiles_x32, ty = t / tiles_x32;
    f32[t] = f64[(ty >> 1) * tiles_x64 + (tx >> 1)] ? tilek::FLAG_FULL : 0u;
}
static __global__ __launch_bounds__(256) void flags_up_kernel(uint32_t* __restrict__ f32, int tiles_x32, int tiles_y32, uint32_t* __restrict__ f64, int tiles_x64) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles_x32 * tiles_y32) return;
    if (f32[t]) {
        f32[t] = 0u;
        const int tx = t % tiles_x32, ty = t / tiles_x32;
        f64[(ty >> 1) * tiles_x64 + (tx >> 1)] = tilek::FLAG_HALO;   // walks only: the bulk sweeps have run
    }
}

// ---- verifier (TDX_SWEEP_VERIFY=1) ------------------------------------------------------------------------------------------
// The sweeps rely on a record being read and written by ONE instruction, so that a tile staging its ring while the neighbouring tile
// writes back in the same round sees the old or the new record, never a mixture.  That holds for aligned 4 / 8 / 16-byte accesses on
// gfx950, but it is not a guarantee of the HIP memory model.  With TDX_SWEEP_VERIFY=1 every sweep is followed by one more pass over the
// QUIESCENT work array (nothing else is running): each tile is staged again and every owned, participating cell is checked -
//   * a cell that is still pending must have a pending contributor (or be one of the cells that can never become ready);
//   * an evaluated cell must have no pending contributor, and re-evaluating it from its contributors' FINAL records with the policy's
//     own expression must reproduce its record bit for bit.
// A record torn or read stale at any point of the sweep leaves a cell whose value does not follow from its contributors, which this
// finds; the call then fails with TDX_ERR_VERIFY and names the first cell.  A policy whose evaluation consumes part of its own
// record (TransLimAlg: the input concentration becomes the deposition) restores it through unevaluate().
template <class Alg>
__device__ __forceinline__ auto unevaluate_record(const Alg& alg, typename Alg::Cell& me, size_t idx, int) -> decltype(alg.unevaluate(me, idx), void()) { alg.unevaluate(me, idx); }
template <class Alg>
__device__ __forceinline__ void unevaluate_record(const Alg&, typename Alg::Cell&, size_t, long) {}

template <class Cell>
__device__ __forceinline__ bool same_record(const Cell& a, const Cell& b) {
    const uint32_t* pa = reinterpret_cast<const uint32_t*>(&a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
    bool eq = true;
#pragma unroll
    for (unsigned i = 0; i < sizeof(Cell) / 4; i++) eq = eq && pa[i] == pb[i];
    return eq;
}

// out[0] = cells checked, out[1] = mismatches, out[2] = smallest linear index of a mismatch, out[3] = kind of that... (bit 62/63 of out[2])
template <class Alg>
__global__ __launch_bounds__(Dim<32>::NT) void verify_kernel(Alg alg, tilek::TileGeom g, Arrays<Alg> A, unsigned long long* __restrict__ out) {
    using Cell = typename Alg::Cell;
    constexpr int TS = Dim<32>::TS, LH = Dim<32>::LH, RPL = Dim<32>::RPL;
    __shared__ Lds<Alg, 32> S;
    const int tile = blockIdx.x;
    const int tid = threadIdx.x, lx = tid % TS, ry0 = (tid / TS) * RPL;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    const int x0 = tx * TS, y0 = ty * TS;
    stage_tile<Alg, 32>(g, tile, S, A);
    unsigned long long checked = 0, bad = 0, first = ~0ull;
    for (int r = 0; r < RPL; r++) {
        __syncthreads();   // staging / the previous row's restore has landed
        const int ly = ry0 + r, c = ly * TS + lx, cl = (ly + 1) * LH + lx + 1;
        const int gx = x0 + lx, gy = y0 + ly;
        const unsigned inf = S.info[c];
        const bool mine = gx < g.nx && gy >= g.y_own0 && gy < g.y_own1 && (inf & INFO_PART) != 0u;
        const Cell me = S.v[cl];
        Cell nb[9];
#pragma unroll
        for (int k = 1; k <= 8; k++) nb[k] = S.v[cl + d2(k) * LH + d1(k)];
        unsigned pb = 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) pb |= pending(Alg::head(nb[k])) ? 1u << (k - 1) : 0u;
        pb &= inf & 0xFFu;
        const size_t idx = size_t(gy) * size_t(g.nx) + size_t(gx);
        __syncthreads();   // every lane holds its neighbourhood: the records may be rewritten
        bool wrong = false;
        if (mine) {
            checked++;
            if (pending(Alg::head(me))) wrong = pb == 0u && !(inf & INFO_DEAD);      // ready, never evaluated
            else if (pb != 0u) wrong = true;                                         // evaluated ahead of a contributor
            else {
                Cell in = me;
                unevaluate_record(alg, in, idx, 0);
                S.v[cl] = in;
                alg.eval(S, c, cl, ly, inf, nb);
                const Cell again = S.v[cl];
                S.v[cl] = me;
                wrong = !same_record(again, me);
            }
        }
        if (wrong) { bad++; first = first < idx ? first : idx; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        checked += __shfl_xor(checked, off, 64);
        bad += __shfl_xor(bad, off, 64);
        const unsigned long long f = __shfl_xor(first, off, 64);
        first = f < first ? f : first;
    }
    if ((tid & 63) == 0) {
        atomicAdd(out + 0, checked);
        if (bad) { atomicAdd(out + 1, bad); atomicMin(out + 2, first); }
    }
}

static inline bool verify_enabled() {
    const char* e = getenv("TDX_SWEEP_VERIFY");   // read per call: the tests switch it on for single cases
    return e != nullptr && atoi(e) != 0;
}
// reads back a verifier's three counters (device words out[0..2]) and turns mismatches into an error
static inline int verify_report(tdx_context* ctx, const char* what, unsigned long long* d_out, int nx) {
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail + TDX_MAIL_VERIFY, d_out, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long checked = ctx->h_mail[TDX_MAIL_VERIFY], bad = ctx->h_mail[TDX_MAIL_VERIFY + 1], first = ctx->h_mail[TDX_MAIL_VERIFY + 2];
    const bool talk = getenv("TDX_SWEEP_VERIFY") != nullptr && atoi(getenv("TDX_SWEEP_VERIFY")) > 1;
    if (talk) fprintf(stderr, "taudem_amd: sweep verifier (%s): %llu cells checked, %llu do not follow from their contributors\n", what, checked, bad);
    if (bad == 0) return TDX_OK;
    char msg[256];
    snprintf(msg, sizeof msg, "sweep verifier (%s): %llu of %llu cells do not follow from their contributors' final records; first at row %llu column %llu of the strip array",
             what, bad, checked, first / (unsigned long long)nx, first % (unsigned long long)nx);
    return tdx_fail(ctx, TDX_ERR_VERIFY, msg);
}
template <class Alg>
static int verify(tdx_context* ctx, const Strip& st, Alg alg, Arrays<Alg> A) {
    tilek::TileGeom g32 = tilek::make_geom(st.nx, st.ny_arr, st.y0, st.y1);
    g32.tiles_x = (st.nx + 31) / 32; g32.tiles_y = (st.ny_arr + 31) / 32;
    unsigned long long* d_out = reinterpret_cast<unsigned long long*>(ctx->d_mail) + TDX_MAIL_VERIFY;
    const unsigned long long init[3] = {0ull, 0ull, ~0ull};
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_out, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // `init` is a local
    hipLaunchKernelGGL((verify_kernel<Alg>), dim3(unsigned(g32.tiles_x) * unsigned(g32.tiles_y)), dim3(Dim<32>::NT), 0, ctx->stream, alg, g32, A, d_out);
    return verify_report(ctx, "tile dependency sweep", d_out, st.nx);
}

// Runs the sweep to the global fixed point: bulk rounds on 32 x 32 tiles until a round has at most TDX_D8_BULK_UNTIL (default 6000)
// active tiles, the rest on 64 x 64 tiles.  The work array must be initialised (pending pattern on participating owned cells) and
// its halo rows exchanged; cells still pending on return (on or below a cycle, fed by the p == 0 quirk) are the caller's to finish.
template <class Alg>
static int run(tdx_context* ctx, const Strip& st, Alg alg, Arrays<Alg> A, uint32_t* /*flags_unused*/, unsigned long long* counts, int64_t* rounds_out,
               int64_t* launches_out, int64_t* outer_out) {
    using Bits = typename BitsOf<sizeof(typename Alg::Cell)>::type;
    hipStream_t s = ctx->stream;
    tilek::TileGeom geom = tilek::make_geom(st.nx, st.ny_arr, st.y0, st.y1);
    static const int bulk_sweeps_env = getenv("TDX_D8_BULK_SWEEPS") ? std::max(0, atoi(getenv("TDX_D8_BULK_SWEEPS"))) : -1;
    geom.max_sweeps = bulk_sweeps_env >= 0 ? bulk_sweeps_env : Alg::kBulkSweeps;
    tilek::TileGeom geom32 = geom;
    geom32.tiles_x = (st.nx + 31) / 32; geom32.tiles_y = (st.ny_arr + 31) / 32;
    const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y), ntiles32 = size_t(geom32.tiles_x) * size_t(geom32.tiles_y);
    uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));
    uint32_t* flags32 = static_cast<uint32_t*>(ctx->scratch(TDX_S_G, ntiles32 * 4 * (1 + tilek::SCHED_LIST_WORDS)));
    if (!flags || !flags32) return TDX_ERR_NOMEM;
    const tilek::Sched sched{flags, flags + ntiles, counts}, sched32{flags32, flags32 + ntiles32, counts};
    // rounds run on 32 x 32 tiles while more than this many are active (policy constant; TDX_D8_BULK_UNTIL overrides it for every policy)
    static const unsigned long long bulk_until = getenv("TDX_D8_BULK_UNTIL") ? strtoull(getenv("TDX_D8_BULK_UNTIL"), nullptr, 10) : (unsigned long long)Alg::kBulkUntil;
    // (max_rounds > 0: stop after that many rounds with the active tiles' flags left in the schedule's flag half `*parity_out` - the multi-strip tail)
    auto run_rounds = [&](bool small, const tilek::TileGeom& gg, const tilek::Sched& sc, unsigned long long stop_at, bool* active_left, int* parity_out,
                          int max_rounds = 0) -> int {
        RoundRunner<flatk::LevelOp> runner(ctx, s, flatk::LevelOp{nullptr, nullptr}, gg, sc, ctx->h_mail + TDX_MAIL_RUN_A, nullptr);
        if (max_rounds > 0) { runner.batch = max_rounds; runner.batch_max = max_rounds; }
        if (small) { runner.grid_full = unsigned(std::min(runner.ntiles, 16 * ctx->num_cus)); runner.grid_small = unsigned(std::min(runner.ntiles, 4 * ctx->num_cus)); }
        static const bool print_rounds = getenv("TDX_DEBUG_ROUNDS") != nullptr;   // active tiles per round on stderr
        runner.print_counts = print_rounds;
        if (print_rounds) fprintf(stderr, "\nd8 sweep rounds(%d tiles of %d):", runner.ntiles, small ? 32 : 64);
        runner.custom_launch = [&](const tilek::TileGeom& rg, unsigned grid, hipStream_t ls, const uint32_t* list, unsigned long long* count, uint32_t* fcur, uint32_t* fnext,
                                   uint32_t* lnext, unsigned pull_max) {
            if (small) hipLaunchKernelGGL((sweep_kernel<Alg, 32, Alg::kMinWaves32>), dim3(grid), dim3(Dim<32>::NT), 0, ls, alg, rg, list, count, fcur, fnext, lnext, pull_max, A);
            else hipLaunchKernelGGL((sweep_kernel<Alg, 64>), dim3(grid), dim3(Dim<64>::NT), 0, ls, alg, rg, list, count, fcur, fnext, lnext, pull_max, A);
        };
        int rcl = runner.start();
        if (rcl != TDX_OK) return rcl;
        *active_left = false;
        while (!runner.done) {
            rcl = runner.enqueue();
            if (rcl != TDX_OK) return rcl;
            TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
            runner.collect();
            if (!runner.done && stop_at > 0 && runner.last_count <= stop_at) { *active_left = true; *parity_out = runner.parity; break; }
            if (!runner.done && max_rounds > 0 && runner.rounds >= max_rounds) { *active_left = true; *parity_out = runner.parity; break; }
        }
        if (rounds_out) *rounds_out += runner.rounds;
        if (launches_out) *launches_out += runner.launches;
        return TDX_OK;
    };
    hipLaunchKernelGGL(tilek::fill_u32_kernel, dim3(tdx_blocks_for(ntiles, 256)), dim3(256), 0, s, flags, tilek::FLAG_FULL, ntiles);   // round 0: every tile
    bool bulk = bulk_until > 0 && ntiles32 > bulk_until;
    int rc;
    for (;;) {
        bool left = false;
        int par = 0;
        if (bulk) {
            hipLaunchKernelGGL(flags_down_kernel, dim3(tdx_blocks_for(ntiles32, 256)), dim3(256), 0, s, flags, geom.tiles_x, flags32, geom32.tiles_x, geom32.tiles_y);
            TDX_HIP_CHECK(ctx, hipMemsetAsync(flags, 0, ntiles * 4, s));
            rc = run_rounds(true, geom32, sched32, bulk_until, &left, &par);
            if (rc != TDX_OK) return rc;
            if (left) {   // what is still active goes on in 64 x 64 tiles
                uint32_t* f32next = par ? sched32.list + 2 * ntiles32 : sched32.flags;
                hipLaunchKernelGGL(flags_up_kernel, dim3(tdx_blocks_for(ntiles32, 256)), dim3(256), 0, s, f32next, geom32.tiles_x, geom32.tiles_y, flags, geom.tiles_x);
            }
            bulk = false;   // (strip re-activations are few tiles: 64 x 64)
        } else left = true;
        if (left) {
            // Multi-strip tail: at most `eager` rounds between two exchanges (areadinf.hip has the reasoning and the measurement): the flow paths that
            // cross strip boundaries advance side by side instead of each waiting for the longest chain of the strip it enters.
            const int eager_env = getenv("TDX_SWEEP_EAGER_ROUNDS") ? std::max(0, atoi(getenv("TDX_SWEEP_EAGER_ROUNDS"))) : 8;   // (0: local fixed points)
            rc = run_rounds(false, geom, sched, 0, &left, &par, st.multi() ? eager_env : 0);
            if (rc != TDX_OK) return rc;
            if (left && par)   // stopped with tiles still active: the next schedule starts from the first flag half
                hipLaunchKernelGGL(tilek::flags_fold_kernel, dim3(tdx_blocks_for(ntiles, 256)), dim3(256), 0, s, sched.flags, sched.list + 2 * ntiles, int(ntiles));
        }
        if (!st.multi()) break;
        // the neighbours' boundary rows (as bit patterns: a pending record must compare equal to itself): cells finished there
        // release the owned cells they drain into (addBorders() + queue refill of src/aread8.cpp:282-303); tiles that see a changed
        // halo cell run again
        int64_t changed = 0;
        Bits outside_bits;
        const typename Alg::Cell oc = Alg::outside();
        memcpy(&outside_bits, &oc, sizeof(Bits));
        rc = strip_exchange<Bits>(ctx, st, reinterpret_cast<Bits*>(A.v), outside_bits, flags, geom.tiles_x, &changed, true, left ? 1 : 0);
        if (rc != TDX_OK) return rc;
        if (changed == 0) break;
        if (outer_out) (*outer_out)++;
    }
#ifdef TDX_REV_CLOCKS
    if constexpr (Alg::kMaxRelease <= 2) {
        unsigned long long h[40];
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rev_clk), sizeof h);
        if (h[5]) fprintf(stderr, "sweep clocks, rounds of >= 50000 tiles (us per activation): activations %llu (full %llu)  stage %.2f  lockstep %.2f  count %.2f  walks %.2f  writeback %.2f\n", h[5], h[6],
                h[0] / 100.0 / h[5], h[1] / 100.0 / h[5], h[2] / 100.0 / h[5], h[3] / 100.0 / h[5], h[4] / 100.0 / h[5]);
        memset(h, 0, sizeof h);
        hipMemcpyToSymbol(HIP_SYMBOL(g_rev_clk), h, sizeof h);
    }
    if constexpr (Alg::kMaxRelease > 2) {
        unsigned long long h[40];
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rev_clk), sizeof h);
        fprintf(stderr, "rev clocks (100 MHz ticks -> us): activations %llu  stage %.2f  setup %.2f  sweeps %.2f (max %.2f)  writeback %.2f us per activation; stretches %.2f (max %llu)\n", h[5],
                h[0] / 100.0 / h[5], h[1] / 100.0 / h[5], h[2] / 100.0 / h[5], h[6] / 100.0, h[3] / 100.0 / h[5], double(h[4]) / h[5], h[7]);
        fprintf(stderr, "  sweep phase us histogram (0..14, 15+):");
        for (int i = 0; i < 16; i++) fprintf(stderr, " %llu", h[8 + i]);
        fprintf(stderr, "\n  stretches histogram (0..14, 15+):");
        for (int i = 0; i < 16; i++) fprintf(stderr, " %llu", h[24 + i]);
        fprintf(stderr, "\n");
        memset(h, 0, sizeof h);
        hipMemcpyToSymbol(HIP_SYMBOL(g_rev_clk), h, sizeof h);
    }
#endif
    if (verify_enabled()) return verify(ctx, st, alg, A);
    return TDX_OK;
}

}  // namespace d8sweep
