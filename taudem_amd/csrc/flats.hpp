// Flat resolution shared by D8FlowDir and DinfFlowDir: the incfall / incrise relaxations of
// resolveflats() (src/d8.cpp:509-646, src/dinf.cpp:650-787).
//
// Both relaxations of the reference are breadth-first LEVEL fields:
//   incfall: elev2(c) = sweep in which c stops incrementing
//            level 1  = a non-crossing neighbour is <= and has a direction
//            level 2 <= an equal, non-crossing neighbour that is NOT in the flat queue and has no
//                       direction (its elev2 stays 1 < st from sweep 2 on)
//            level t  = 1 + min level over equal, non-crossing neighbours that are in the queue
//   incrise: q(c) = 1 where a neighbour is strictly higher, else 1 + min q over the 8 queue neighbours
// i.e. fixed points of  v(c) <- min(v(c), 1 + min over an eligible-neighbour mask of v(n)).  The
// reference re-sweeps every flat cell once per level (O(N^1.5)); here the level-1/2 seeds and the two
// 8-bit eligibility masks are produced by one classification pass over the flat queue and the two
// fields are then relaxed by the tile engine of tile_relax.hpp (in-LDS convergence per 64x64 tile,
// device-chained rounds) - rounds ~ flat diameter / 64 instead of one launch per level.
//
// Marker arrays (int32 per cell):
//   lvl: -1 not in the flat queue Q; 0 in Q, not (yet) stopped; t>0 stopped incrementing in sweep t
//        => elev2 = t for stopped cells, 1+T for cells that never stop (pits), 1 outside Q
//   rq : -1 not in Q; 0 in Q, unmarked (dn == 0); q>0 first marked in incrise sweep q (dn == 1)
//        => s = Tr - q + 1 for marked cells, 0 otherwise
// T / Tr are the numbers of sweeps the reference's while-loops execute:
//   incfall loop ends when a sweep increments as many cells as the previous one (src/d8.cpp:523):
//       T = 1 if every cell stops in sweep 1, else max(L,1)+1 with L = last sweep that stopped a cell
//   incrise loop ends when the marked count stops growing (src/d8.cpp:631): Tr = Qmax + 1
#pragma once
#include <algorithm>
#include <functional>
#include <vector>

#include "context.hpp"
#include "device_common.hpp"
#include "strips.hpp"
#include "tile_relax.hpp"

struct FlatLevels { int T; int Tr; int has_pits; };
// phases of a flow-direction call in the segment trace (context.hpp: static strings, the same on every rank): the slope pass, then per call of
// resolveflats() (src/d8.cpp:307-316) the classification, the two level fields, the directions (setFlow2 / SET2) and the next iteration's elevations
struct FlatPhases { const char *classify, *levels, *stats, *directions, *next; };
static inline const FlatPhases& flat_phases(int iteration) {   // iteration = 1, 2, 3 ... (4 and later share one name)
    static const FlatPhases P[4] = {{"flats 1: classify", "flats 1: level fields", "flats 1: statistics", "flats 1: directions", "flats 1: next elev"},
                                    {"flats 2: classify", "flats 2: level fields", "flats 2: statistics", "flats 2: directions", "flats 2: next elev"},
                                    {"flats 3: classify", "flats 3: level fields", "flats 3: statistics", "flats 3: directions", "flats 3: next elev"},
                                    {"flats 4+: classify", "flats 4+: level fields", "flats 4+: statistics", "flats 4+: directions", "flats 4+: next elev"}};
    return P[iteration < 1 ? 0 : (iteration > 4 ? 3 : iteration - 1)];
}
// Level markers are stored as int16, like the reference's elev2 / dn / s partitions (SHORT_TYPE, src/d8.cpp:483,486,595): half the
// bytes of every tile image, classification pass and list gather.  Levels saturate at LVL_SAT, which flats_bfs reports as an error
// (the reference's int16 counters overflow there too).
using lvl_t = int16_t;
constexpr int LVL_SAT = 32767;
constexpr int TDX_FLATS_TOO_DEEP = -4242;   // internal: flats_bfs on int16 fields met a level beyond them; the caller re-runs on int32 fields
template <class LV>
struct FlatBuffersT { LV* lvl; LV* rq; };
using FlatBuffers = FlatBuffersT<lvl_t>;

// elev2 + s as the reference's int16 arithmetic leaves it (src/d8.cpp:545,640-645)
// (LV = int16: wraps like the reference's short; LV = int32: the wider fields of a flat deeper than 32 766 levels, see flats_bfs)
template <class LV>
__host__ __device__ __forceinline__ LV flat_elev2(int lvl, int rq, FlatLevels fl) {
    const int e = (lvl < 0) ? 1 : (lvl > 0 ? lvl : 1 + fl.T);
    const int s = (rq > 0) ? (fl.Tr - rq + 1) : 0;
    return (LV)(e + s);
}

namespace flatk {
using namespace tdxk;

// Per flat cell: seeds (lvl = 1 "low", 2 quirk source; rq = 1 "higher neighbour"), the eligibility
// masks of both relaxations (bit k-1 = neighbour k may hand its level to this cell) and the
// activation flag of the cell's tile.  Flat cells are interior cells, so all 8 neighbours exist.
constexpr int CLASSIFY_ITEMS = 4;
template <class Traits, class LV>
__global__ __launch_bounds__(256) void classify_kernel(Traits tr, const float* __restrict__ Z, int nx, int tiles_x,
                                                       const uint32_t* __restrict__ list, unsigned long long nq, LV* __restrict__ lvl,
                                                       LV* __restrict__ rq, uint8_t* __restrict__ fmask, uint8_t* __restrict__ rmask,
                                                       uint32_t* __restrict__ tile_flags, uint8_t* __restrict__ tile_masked) {
    const unsigned long long base = (unsigned long long)blockIdx.x * (256 * CLASSIFY_ITEMS) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < CLASSIFY_ITEMS; i++) {
        const unsigned long long q = base + (unsigned long long)i * 256;
        if (q < nq) {
            const size_t c = list[q];
            const float z0 = Z[c];
            bool low = false, quirk = false, higher = false;
            unsigned fm = 0, rm = 0;
#pragma unroll
            for (int k = 1; k <= 8; k++) {
                const size_t n = size_t(ptrdiff_t(c) + ptrdiff_t(d2(k)) * nx + d1(k));
                const float zd = z0 - Z[n];
                const bool inq = lvl[n] >= 0;   // sign only: concurrent seed writes keep the sign
                if (zd < 0) higher = true;
                if (inq) rm |= 1u << (k - 1);
                if (!tr.dont_cross(c, nx, k)) {
                    if (zd >= 0 && tr.has_direction(n)) low = true;
                    else if (zd == 0) {
                        if (inq) fm |= 1u << (k - 1);
                        else quirk = true;
                    }
                }
            }
            if (low) lvl[c] = 1;
            else if (quirk) lvl[c] = 2;
            if (higher) rq[c] = 1;
            fmask[c] = uint8_t(low ? 0u : fm);     // a level-1 cell can never improve
            rmask[c] = uint8_t(higher ? 0u : rm);
            const unsigned y = unsigned(c / size_t(nx)), x = unsigned(c - size_t(y) * size_t(nx));
            const unsigned tile = (y / tilek::TS) * unsigned(tiles_x) + x / tilek::TS;
            tile_flags[tile] = tilek::FLAG_FULL;
            if (!low && fm != rm) tile_masked[tile] = 1;   // an in-queue neighbour that incfall must not read: no plain tile (see LevelPlainT)
        }
    }
}

// Classification for flat resolution as ONE streaming pass (replaces marker reset + list gathers): the flat queue is
// exactly {p == 0} (D-infinity: {angle == -1}) in every outer iteration (resolved cells leave it, nothing enters it), so queue membership, the
// level-1/2 seeds of incfall, the seeds of incrise and both eligibility masks are a function of the 3x3 windows of
// Z and the direction raster.  Every owned cell gets its markers (lvl / rq: -1 outside the queue) and masks; lanes walk 16-row column
// segments with both windows in registers (6 row loads per output row).  Semantics: flatk::classify_kernel.
// (62-column window like d8_slope_kernel: a lane loads ONE value per row and array, the west / east neighbours are lane shifts; the 2 x 18 row
// loads of a lane are issued back to back.  The first version read three overlapping cells per row and array inside the row loop: 1.36 ms.)
constexpr int CLS_COLS = 62, CLS_ROWS = 16;
template <class LV, class Src>
__global__ __launch_bounds__(256) void classify_stream_kernel(const float* __restrict__ Z, Src src, int nx, int ny,
                                                                 int y_own0, int y_own1, int tiles_x, LV* __restrict__ lvl,
                                                                 LV* __restrict__ rq, uint8_t* __restrict__ fmask, uint8_t* __restrict__ rmask,
                                                                 uint32_t* __restrict__ tile_flags, uint8_t* __restrict__ tile_masked, int nbx, int xmap,
                                                                 uint8_t* __restrict__ notfull) {
    using tilek::lane_left;
    using tilek::lane_right;
    const int bx = tdxk::xcd_block_x(nbx, xmap);
    if (bx < 0) return;
    const int lx = threadIdx.x & 63;
    const int x = bx * CLS_COLS - 1 + lx;
    const int ybase = __builtin_amdgcn_readfirstlane(y_own0 + blockIdx.y * (4 * CLS_ROWS) + (threadIdx.x >> 6) * CLS_ROWS);
    const bool mine = lx >= 1 && lx <= CLS_COLS && x < nx;
    const bool inx = x >= 0 && x < nx;
    const int xc = x < 0 ? 0 : (x >= nx ? nx - 1 : x);
    float z[CLS_ROWS + 2];
    typename Src::Raw pw[CLS_ROWS + 2];
#pragma unroll
    for (int j = 0; j < CLS_ROWS + 2; j++) {
        const int y = ybase - 1 + j, yc = y < 0 ? 0 : (y >= ny ? ny - 1 : y);
        const size_t o = size_t(yc) * size_t(nx) + size_t(xc);
        z[j] = Z[o];
        pw[j] = src.load(o);
    }
    // The direction codes in ONE-HOT form (bit c for a code c in 0 .. 8, nothing for nodata / outside): "in the queue" is bit 0, "has a direction" bits 1 .. 8,
    // dontCross two bit tests - and the per-neighbour case analysis below integer and / or on 0 / 1 values.  (As nested if / else on comparisons it compiled to
    // ~145 vector and ~200 scalar instructions per cell row - lane-mask algebra and branches - and the pass waited for its instructions, not for memory:
    // profiles/r04zzzz_pmc_sq_summary.json.)
    unsigned ow[CLS_ROWS + 2];
#pragma unroll
    for (int j = 0; j < CLS_ROWS + 2; j++) {
        const int y = ybase - 1 + j;
        const bool in = inx && y >= 0 && y < ny;   // (never looked at from a flat cell: flat cells are interior cells)
        if (!in) z[j] = 0.f;
        ow[j] = in ? Src::onehot(pw[j]) : 0u;
    }
    int flag_row0 = -1, flag_row1 = -1;   // tile rows (of the relaxation's tile grid) in which this lane saw a flat cell
    int masked_row = -1;                  // ... a flat cell whose incfall mask shuts out an in-queue neighbour (flatk::LevelPlainT)
    unsigned fall_all = 0xFFu, rise_all = 0xFFu;   // AND of this lane's masks: 0xFF = every cell of its column segment may move and look at all eight neighbours (flats.hpp: OPEN WATER)
#pragma unroll
    for (int r = 0; r < CLS_ROWS; r++) {
        const int y = ybase + r;
        const float zn1 = z[r], zc1 = z[r + 1], zs1 = z[r + 2];
        const unsigned on1 = ow[r], oc1 = ow[r + 1], os1 = ow[r + 2];
        const float zn0 = lane_left(zn1, 0.f), zn2 = lane_right(zn1, 0.f), zc0 = lane_left(zc1, 0.f), zc2 = lane_right(zc1, 0.f);
        const float zs0 = lane_left(zs1, 0.f), zs2 = lane_right(zs1, 0.f);
        const unsigned on0 = lane_left(on1, 0u), on2 = lane_right(on1, 0u), oc0 = lane_left(oc1, 0u), oc2 = lane_right(oc1, 0u);
        const unsigned os0 = lane_left(os1, 0u), os2 = lane_right(os1, 0u);
        if (mine && y < y_own1) {
            const size_t idx = size_t(y) * size_t(nx) + size_t(x);
            LV l = -1, q = -1;
            unsigned fm = 0, rm = 0;
            if (oc1 & 1u) {   // a flat cell: interior, all eight neighbours valid
                const float z0 = zc1;
                bool higher = false;
                unsigned lowi = 0, fmq = 0;   // fmq: bits 0-7 the incfall mask, bit 8 the quirk (an equal neighbour that is neither in the queue nor has a direction)
                // neighbour k: elevation, one-hot code, "does not cross" (0 / 1; dontCross(k), src/d8.cpp:54-100, from the cardinal neighbours' codes).  Per neighbour,
                // as in flatk::classify_kernel: higher |= zd < 0; rm bit if in the queue; and unless the step crosses a flow path:
                // zd >= 0 towards a cell with a direction -> low; else zd == 0 -> fm bit (in the queue) or the quirk.  zd == 0 implies zd >= 0, so the
                // second case only sees cells without a direction.
#define TDX_CLS(K, ZN, ON, NC)                                                                                   \
    {                                                                                                             \
        const float zd = z0 - (ZN);                                                                               \
        const unsigned inq = (ON) & 1u, isdir = min((ON) & 0x1FEu, 1u), oth = ((ON) & 0x1FFu) ? 0u : 1u;          \
        higher |= zd < 0;                                                                                         \
        rm |= inq << ((K) - 1);                                                                                   \
        lowi |= zd >= 0 ? ((NC) & isdir) : 0u;                                                                    \
        fmq |= zd == 0 ? ((0u - (NC)) & ((inq << ((K) - 1)) | (oth << 8))) : 0u;                                  \
    }
                TDX_CLS(1, zc2, oc2, 1u)
                TDX_CLS(2, zn2, on2, (((oc2 >> 4) | (on1 >> 8)) & 1u) ^ 1u)
                TDX_CLS(3, zn1, on1, 1u)
                TDX_CLS(4, zn0, on0, (((on1 >> 6) | (oc0 >> 2)) & 1u) ^ 1u)
                TDX_CLS(5, zc0, oc0, 1u)
                TDX_CLS(6, zs0, os0, (((os1 >> 4) | (oc0 >> 8)) & 1u) ^ 1u)
                TDX_CLS(7, zs1, os1, 1u)
                TDX_CLS(8, zs2, os2, (((oc2 >> 6) | (os1 >> 2)) & 1u) ^ 1u)
#undef TDX_CLS
                const bool low = lowi != 0u, quirk = (fmq >> 8) != 0u;
                fm = fmq & 0xFFu;
                l = low ? 1 : (quirk ? 2 : 0);
                q = higher ? 1 : 0;
                const int tr = y / tilek::TS;
                if (!low && fm != rm) masked_row = tr;
                if (low) fm = 0;       // a level-1 cell can never improve
                if (higher) rm = 0;
                if (flag_row0 < 0) flag_row0 = tr; else if (tr != flag_row0) flag_row1 = tr;
            }
            lvl[idx] = l;
            rq[idx] = q;
            fmask[idx] = uint8_t(fm);
            rmask[idx] = uint8_t(rm);
            fall_all &= fm;
            rise_all &= rm;
        }
    }
    if (notfull != nullptr && mine) {   // the tile rows this lane's segment lies in are not full for a field unless every cell of the segment is (rows beyond the owned ones never are)
        const int ylast = ybase + CLS_ROWS - 1;
        const int tr0 = ybase / tilek::TS, tr1 = (ylast < ny ? ylast : ny - 1) / tilek::TS;
        const bool short_seg = ylast >= y_own1;
        if (fall_all != 0xFFu || short_seg) { notfull[2 * (size_t(tr0) * tiles_x + x / tilek::TS)] = 1; if (tr1 != tr0) notfull[2 * (size_t(tr1) * tiles_x + x / tilek::TS)] = 1; }
        if (rise_all != 0xFFu || short_seg) { notfull[2 * (size_t(tr0) * tiles_x + x / tilek::TS) + 1] = 1; if (tr1 != tr0) notfull[2 * (size_t(tr1) * tiles_x + x / tilek::TS) + 1] = 1; }
    }
    if (flag_row0 >= 0) tile_flags[flag_row0 * tiles_x + x / tilek::TS] = tilek::FLAG_FULL;
    if (flag_row1 >= 0) tile_flags[flag_row1 * tiles_x + x / tilek::TS] = tilek::FLAG_FULL;
    if (masked_row >= 0) {   // rare (dontCross at a lake shore): all the tile rows this lane touched, a superset is harmless
        tile_masked[flag_row0 * tiles_x + x / tilek::TS] = 1;
        if (flag_row1 >= 0) tile_masked[flag_row1 * tiles_x + x / tilek::TS] = 1;
    }
}

// level field in the marker convention above: values <= 0 read as +inf; only cells with a mask move.  S = storage type of the field in
// HBM (the values in registers are 32-bit either way): int16 for the two level fields, where a candidate level saturates at LVL_SAT -
// the fixed point of the saturating operator is min(level, LVL_SAT) per cell, i.e. the exact field whenever every level fits, and
// flats_bfs turns a saturated maximum into the error the reference's overflowing int16 counters stand for; int32 for the plain marks.
//
// PLAIN tiles.  A neighbour that a level-field mask excludes is, almost always, a cell outside the queue - which holds +inf for good - so the masked
// minimum equals the minimum over all eight neighbours and the mask only says WHETHER the cell may move.  (incrise: always - its mask is
// "neighbour in the queue".  incfall: unless an in-queue neighbour is shut out by dontCross or by a different elevation; the classification
// marks the tiles that own such a cell in TM.)  Unmarked tiles run LevelPlainT: the nine-way minimum as min3 + six DPP-fused v_min_i32, the lock
// as one v_med3_i32 - 12 VALU instructions per 64-cell row instead of 38 - and the int16 saturation moves to the store (the fixed point of
// "saturate when stored" is the same min(level, LVL_SAT): a stored LVL_SAT never improves a neighbour below LVL_SAT).
template <class S>
struct LevelPlainT {
    using T = int;
    static constexpr int kUniform = 8;
    S* G;
    const uint8_t* M;
    static __device__ __forceinline__ int inf() { return 0x3fffffff; }
    using Raw = S;
    using CellRaw = uint8_t;
    __device__ __forceinline__ S load_raw(size_t idx) const { return G[idx]; }
    static __device__ __forceinline__ int decode(S g) { return g > 0 ? int(g) : inf(); }
    static __device__ __forceinline__ bool raw_can_move(S g) { return g >= 0; }   // a negative marker = outside the queue: its mask byte may be stale (flats_bfs)
    __device__ __forceinline__ void store(size_t idx, int v) const { G[idx] = S(sizeof(S) == 2 && v > LVL_SAT ? LVL_SAT : v); }
    __device__ __forceinline__ uint8_t cell_raw(size_t idx) const { return M[idx]; }
    static __device__ __forceinline__ void cell_decode(uint8_t m, int& lock, unsigned& mask) { lock = m ? 0 : inf(); mask = m; }
    // The constant of a cell is 0 where it may move and THE VALUE IT WAS LOADED WITH where it may not (mask 0: seeds, cells outside the queue,
    // cells of other strips): median(m + 1, 0, own) = min(m + 1, own) (both positive), median(m + 1, own, own) = own.
    static __device__ __forceinline__ int cell_floor(int lock, int v) { return lock == 0 ? 0 : v; }
    static __device__ __forceinline__ int apply(int lock, int own, int m) { return tilek::med3_raw(m + 1, lock, own); }
    static __device__ __forceinline__ bool settled(int, int v) { return v <= 1; }
    // activation filter (tile_relax.hpp): a halo cell that may move (mask != 0) improves from a neighbour value below its own level - 1
    static __device__ __forceinline__ int act_never() { return int(0x80000000u); }
    static __device__ __forceinline__ int act_threshold(uint8_t m, int v) { return m ? v - 1 : act_never(); }
    // ... and the value alone says as much: -1 = outside the queue, a seed's level (1, 2) is below every new value
    static __device__ __forceinline__ int act_threshold_raw(S g) { return g >= 0 ? decode(g) - 1 : act_never(); }
};

// ---- OPEN WATER: aligned blocks of FULL tiles in closed form ---------------------------------------------------------------------------------------------
// A tile is FULL for a level field when every one of its 4096 cells may move and may look at all eight neighbours (mask 0xFF: the cell and its neighbours
// are in the queue, level with each other, no flow path in between - the inside of a lake).  There the operator is the plain chessboard distance: inside a
// square block R of full tiles the fixed point against the ring of cells around R is
//       v(c) = min over ring cells b of  v(b) + max(|dx|, |dy|)                                  (every king's path from b to c inside R is allowed)
// and the rest of the raster sees R only through R's RIM (its outermost cells).  So a block is ONE node of the round schedule (tile_relax.hpp:
// TileGeom::remap / blk_k): an activation recomputes the rim from the ring in closed form - per ring edge a prefix / suffix minimum of h, h - j and h + j
// answer every rim cell in O(1) - and raises the flags of the tiles around what moved; the inside is filled ONCE, after the relaxation has ended, from the
// final rim (rim values along an edge differ by at most one from cell to cell, so a cell at distance d from an edge takes d + the minimum over the 2d + 1
// rim cells across: sparse tables).  A front crosses a block of K x K tiles in one round instead of K: the lake that carries the tail of incfall at
// BASELINE.json configs[1] (8 001 levels) is 3 998 full tiles wide open, and the CPU model of the schedule (scripts/sim/level_vcycle.c, SIM_MACRO=8)
// needs 73 rounds instead of 168.  The values are those of the tile relaxation: the same fixed point of the same operator.
constexpr int MACRO_KMAX = 8;                         // largest block edge in tiles (ring edges of 64 K + 2 cells in LDS)
constexpr int MACRO_LMAX = MACRO_KMAX * 64 + 2;
constexpr int MACRO_INF = 0x3fffffff;
constexpr int MACRO_LDS_WORDS = 5 * MACRO_LMAX + 4 * (MACRO_LMAX - 2) + 8;   // (+ moved[8])   // ring edge: h, pre, suf, pm, sp; rim accumulators t, b, l, r; a few words

// Per aligned 8 x 8 region of tiles and field: the largest aligned blocks (8, 4, 2 tiles on edge) of tiles that are full for the field (notfull[2 t + field] == 0
// and wholly inside the owned rows / the raster).  remap[t] = first tile of t's block + log2 K in bits 27-28 (tilek::entry_tile / entry_blk_k), t itself outside blocks; blk_k[t] = K for a block's first tile, else 0.
// One thread per region and field (blockIdx.y); a region's marks are eight 16-byte rows.
static __global__ __launch_bounds__(64) void find_blocks_kernel(const uint8_t* __restrict__ notfull, int tiles_x, int tiles_y, int nx, int y_lo, int y_hi, int kmax,
                                                                uint32_t* __restrict__ remapF, uint8_t* __restrict__ blkF, uint32_t* __restrict__ remapR, uint8_t* __restrict__ blkR) {
    const int field = int(blockIdx.y);
    uint32_t* const remap = field ? remapR : remapF;
    uint8_t* const blk_k = field ? blkR : blkF;
    const int rx = (tiles_x + 7) / 8;
    const int reg = blockIdx.x * 64 + threadIdx.x;
    if (reg >= rx * ((tiles_y + 7) / 8)) return;
    const int bx0 = (reg % rx) * 8, by0 = (reg / rx) * 8;
    unsigned long long fm = 0;   // bit j * 8 + i: tile (bx0 + i, by0 + j) is full
    const bool vec = (tiles_x & 7) == 0;   // rows of 8 tiles = 16 aligned bytes
    for (int j = 0; j < 8; j++) {
        const int ty = by0 + j;
        if (ty >= tiles_y || ty * tilek::TS < y_lo || (ty + 1) * tilek::TS > y_hi) continue;
        unsigned rowbits = 0;
        if (vec) {
            const uint4 w = *reinterpret_cast<const uint4*>(notfull + 2 * (size_t(ty) * tiles_x + bx0));
            const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 8; i++) if (((ww[i >> 1] >> (16 * (i & 1) + 8 * field)) & 0xFFu) == 0u) rowbits |= 1u << i;
        } else {
            for (int i = 0; i < 8 && bx0 + i < tiles_x; i++) if (notfull[2 * (size_t(ty) * tiles_x + bx0 + i) + field] == 0) rowbits |= 1u << i;
        }
        for (int i = 0; i < 8; i++) if (bx0 + i >= tiles_x || (bx0 + i + 1) * tilek::TS > nx) rowbits &= ~(1u << i);
        fm |= (unsigned long long)rowbits << (8 * j);
    }
    // block edge per tile of the region: 8 if the whole region is full, else 4 per full aligned quadrant, else 2 per full aligned pair of pairs
    auto full_block = [&](int i0, int j0, int kk) {
        unsigned long long m = 0;
        for (int b = 0; b < kk; b++) m |= ((kk == 8 ? 0xFFull : (kk == 4 ? 0xFull : 0x3ull)) << i0) << ((j0 + b) * 8);
        return (fm & m) == m;
    };
    for (int j = 0; j < 8; j++) {
        const int ty = by0 + j;
        if (ty >= tiles_y) break;
        for (int i = 0; i < 8; i++) {
            const int tx = bx0 + i;
            if (tx >= tiles_x) break;
            const size_t t = size_t(ty) * tiles_x + tx;
            int k = 0;
            for (int kk = kmax; kk >= 2 && !k; kk >>= 1) if (full_block(i & ~(kk - 1), j & ~(kk - 1), kk)) k = kk;
            if (k) {
                const int i0 = i & ~(k - 1), j0 = j & ~(k - 1);
                remap[t] = uint32_t(size_t(by0 + j0) * tiles_x + bx0 + i0) | (uint32_t(k == 8 ? 3 : (k == 4 ? 2 : 1)) << tilek::TILE_BLK_SHIFT);   // (+ log2 K: a list entry)
                blk_k[t] = uint8_t((i == i0 && j == j0) ? k : 0);
            } else { remap[t] = uint32_t(t); blk_k[t] = 0; }
        }
    }
}

// only a block's first tile is a node of the schedule: the flags the classification raised on its other tiles are dropped
static __global__ __launch_bounds__(256) void block_flags_kernel(uint32_t* __restrict__ flags, const uint32_t* __restrict__ remap, int ntiles) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < ntiles && tilek::entry_tile(remap[t]) != t) flags[t] = 0u;
}
// inclusive minimum scans of one ring edge h[0 .. L) by ONE wave each: wave 0 prefix of h, 1 suffix of h, 2 prefix of h - j, 3 suffix of h + j
__device__ __forceinline__ void macro_edge_scans(const int* __restrict__ h, int L, int* __restrict__ pre, int* __restrict__ suf, int* __restrict__ pm, int* __restrict__ sp) {
    const int wv = int(threadIdx.x >> 6), lane = int(threadIdx.x & 63);
    const bool fwd = (wv & 1) == 0;
    int* const out = wv == 0 ? pre : (wv == 1 ? suf : (wv == 2 ? pm : sp));
    int carry = MACRO_INF;
    for (int base = 0; base < L; base += 64) {
        const int j = fwd ? base + lane : L - 1 - base - lane;          // forward scans walk up, suffix scans walk down
        int v = MACRO_INF;
        if (j >= 0 && j < L) { v = h[j]; if (wv == 2) v -= j; else if (wv == 3) v += j; }
        // inclusive minimum scan over the wave's 64 lanes in seven DPP steps (rows of 16: row_shr 1, 2, 4, 8; then lane 15 of a row into the next row, lane 31 into
        // the upper half; a lane without a source keeps +inf) - as ds_bpermute shuffles the six steps were 40 % of a block update
        {
#define TDX_MACRO_SCAN_STEP(CTRL, ROWS) { const int t = __builtin_amdgcn_update_dpp(MACRO_INF, v, CTRL, ROWS, 0xf, false); v = t < v ? t : v; }
            TDX_MACRO_SCAN_STEP(0x111, 0xf) TDX_MACRO_SCAN_STEP(0x112, 0xf) TDX_MACRO_SCAN_STEP(0x114, 0xf) TDX_MACRO_SCAN_STEP(0x118, 0xf)   // row_shr:1, 2, 4, 8
            TDX_MACRO_SCAN_STEP(0x142, 0xa)                                                                                                   // row_bcast:15 into rows 1 and 3
            TDX_MACRO_SCAN_STEP(0x143, 0xc)                                                                                                   // row_bcast:31 into rows 2 and 3
#undef TDX_MACRO_SCAN_STEP
        }
        v = carry < v ? carry : v;
        if (j >= 0 && j < L) out[j] = v;
        carry = __shfl(v, 63, 64);
    }
}
// best level a ring edge (arrays over ring index 0 .. L) offers a cell at ring index a, d >= 1 cells away from the edge's line: min over j of h[j] + max(|a - j|, d)
__device__ __forceinline__ int macro_offer(const int* __restrict__ h, const int* __restrict__ pre, const int* __restrict__ suf, const int* __restrict__ pm,
                                           const int* __restrict__ sp, int L, int a, int d) {
    const int lo = a - d, hi = a + d;
    int near;   // minimum of h over [lo, hi] (clipped): the window touches an end of the edge unless d == 1
    if (lo <= 0) near = pre[hi < L - 1 ? hi : L - 1];
    else if (hi >= L - 1) near = suf[lo];
    else { near = h[lo]; for (int j = lo + 1; j <= hi; j++) near = h[j] < near ? h[j] : near; }   // (d == 1: three cells)
    int best = near + d;
    if (lo - 1 >= 0) { const int f = pm[lo - 1] + a; best = f < best ? f : best; }
    if (hi + 1 <= L - 1) { const int f = sp[hi + 1] - a; best = f < best ? f : best; }
    return best;
}

template <int INC, class S>   // INC 1: breadth-first level field; 0: plain reachability ("some selected neighbour is marked")
struct LevelOpT {
    using T = int;
    static constexpr int kMacroLdsWords = INC == 1 ? MACRO_LDS_WORDS : 0;   // (what tilek::has_macro looks for: the level fields only, not the closure)
    static constexpr int kUniform = 0;
    S* G;
    const uint8_t* M;
    const uint8_t* TM = nullptr;   // INC 1 only: per tile, 1 = a cell of the tile needs its mask (null: no tile does)
    static constexpr bool kHasPlain = INC == 1;
    using Plain = LevelPlainT<S>;
    __device__ __forceinline__ Plain plain() const { return Plain{G, M}; }
    __device__ __forceinline__ bool tile_masked(int tile) const { return TM != nullptr && TM[tile] != 0; }
    static __device__ __forceinline__ int inf() { return 0x3fffffff; }
    using Raw = S;
    using CellRaw = uint8_t;
    __device__ __forceinline__ S load_raw(size_t idx) const { return G[idx]; }
    static __device__ __forceinline__ int decode(S g) { return g > 0 ? int(g) : inf(); }
    static __device__ __forceinline__ bool raw_can_move(S g) { return g >= 0; }   // (the closure's marks are 0 / 1: always true there)
    __device__ __forceinline__ void store(size_t idx, int v) const { G[idx] = S(v); }
    __device__ __forceinline__ uint8_t cell_raw(size_t idx) const { return M[idx]; }
    static __device__ __forceinline__ void cell_decode(uint8_t m, int& cst, unsigned& mask) { cst = 0; mask = m; }
    static __device__ __forceinline__ int apply(int, int own, int m) {
        int t = m + INC;
        // a real level beyond the int16 range saturates; inf() + 1 (no neighbour reached yet) stays above every value
        if (sizeof(S) == 2 && INC) t = unsigned(t - LVL_SAT) < unsigned(inf() - LVL_SAT) ? LVL_SAT : t;
        return t < own ? t : own;
    }
    static __device__ __forceinline__ int cell_floor(int cst, int) { return cst; }
    static __device__ __forceinline__ bool settled(int, int v) { return v <= 1; }
    // activation filter (tile_relax.hpp): a halo cell with a mask improves from a neighbour value below its own value - INC (whichever neighbours the mask selects)
    static __device__ __forceinline__ int act_never() { return int(0x80000000u); }
    static __device__ __forceinline__ int act_threshold(uint8_t m, int v) { return m ? v - INC : act_never(); }
    // ... and the value alone says as much: a negative marker = outside the queue, a seed's value is below every new value
    static __device__ __forceinline__ int act_threshold_raw(S g) { return g >= 0 ? decode(g) - INC : act_never(); }

    // One activation of a macro block (see OPEN WATER above; tilek::relax_kernel calls it for a block's first tile, all 256 threads): the rim of the k x k tile
    // block from the ring around it, in closed form; rim cells that moved are stored and the tiles outside the block that touch them are activated.
    __device__ __forceinline__ int macro_update(const tilek::TileGeom& g, int tile, int k, int* __restrict__ lds, tilek::TileLds& TL, uint32_t* __restrict__ flags_next,
                                                unsigned long long* __restrict__ dbg = nullptr) const {
        const int tid = int(threadIdx.x);
        const unsigned long long tq0 = dbg ? __builtin_readcyclecounter() : 0ull;
        const int W = k * tilek::TS, L = W + 2;
        const int tx0 = tile % g.tiles_x, ty0 = tile / g.tiles_x;
        const int X0 = tx0 * tilek::TS, Y0 = ty0 * tilek::TS;
        const size_t pitch = size_t(g.nx);
        int* const h = lds; int* const pre = h + MACRO_LMAX; int* const suf = pre + MACRO_LMAX; int* const pm = suf + MACRO_LMAX; int* const sp = pm + MACRO_LMAX;
        int* const acc = sp + MACRO_LMAX;                  // [4][MACRO_LMAX - 2]: rim edges top, bottom, left, right
        int* const moved = acc + 4 * (MACRO_LMAX - 2);     // [4]: per rim edge, bit s = a cell of its s-th 64-cell segment moved
        constexpr int AW = MACRO_LMAX - 2;
        // rim cell i of edge e (0 top, 1 bottom, 2 left, 3 right)
        auto rim_idx = [&](int e, int i) -> size_t {
            const int x = e == 2 ? 0 : (e == 3 ? W - 1 : i), y = e == 0 ? 0 : (e == 1 ? W - 1 : i);
            return size_t(Y0 + y) * pitch + size_t(X0 + x);
        };
        // ---- every global load of the activation is issued up front (addresses clamped, validity applied afterwards: a load behind a bounds test or
        // inside a loop with an LDS store is waited for before the next one is issued - the first version spent 12 memory latencies here)
        S raw_rim[4][2], raw_ring[4][3];
        unsigned ring_ok = 0;
        // (the first 4 (k + 2) threads also fetch, now, which node of the schedule stands for "their" tile outside the block: needed only at the very end)
        int out_target = -1;
        if (tid < 4 * (k + 2)) {
            const int e = tid / (k + 2), j = tid % (k + 2) - 1;
            const int ntx = e == 2 ? tx0 - 1 : (e == 3 ? tx0 + k : tx0 + j), nty = e == 0 ? ty0 - 1 : (e == 1 ? ty0 + k : ty0 + j);
            if (ntx >= 0 && ntx < g.tiles_x && nty >= 0 && nty < g.tiles_y) out_target = int(g.remap[size_t(nty) * g.tiles_x + ntx]);
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = tid + u * 256;
                raw_rim[e][u] = G[rim_idx(e, i < W ? i : W - 1)];
            }
#pragma unroll
        for (int E = 0; E < 4; E++)      // ring edges: 0 the row above, 1 the row below, 2 the column to the left, 3 the column to the right (corners included)
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int j = tid + u * 256;
                const int x = E == 2 ? X0 - 1 : (E == 3 ? X0 + W : X0 - 1 + j), y = E == 0 ? Y0 - 1 : (E == 1 ? Y0 + W : Y0 - 1 + j);
                if (j < L && x >= 0 && x < g.nx && y >= 0 && y < g.ny) ring_ok |= 1u << (E * 3 + u);
                const int xc = x < 0 ? 0 : (x >= g.nx ? g.nx - 1 : x), yc = y < 0 ? 0 : (y >= g.ny ? g.ny - 1 : y);
                raw_ring[E][u] = G[size_t(yc) * pitch + size_t(xc)];
            }
        int old[4][2], best[4][2];   // a thread's (up to) two cells of each rim edge: their levels as loaded, the best offer so far
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = tid + u * 256;
                old[e][u] = i < W ? decode(raw_rim[e][u]) : MACRO_INF;
                best[e][u] = old[e][u];
            }
        if (tid < 8) moved[tid] = 0;   // [0 .. 4): per ring edge, bit p + 1 = the tile outside at position p (-1 .. k) can be improved; [4 .. 8): the ring edge holds a level at all
#pragma unroll
        for (int E = 0; E < 4; E++) {
            bool fin = false;
#pragma unroll
            for (int u = 0; u < 3; u++) fin = fin || (((ring_ok >> (E * 3 + u)) & 1u) && raw_ring[E][u] > 0);
            if (__ballot(fin) != 0ull && (tid & 63) == 0) moved[4 + E] = 1;
        }
#pragma unroll
        for (int E = 0; E < 4; E++) {
            __syncthreads();            // (the previous edge's tables are no longer read; the first time: the flags above are in place)
            if (moved[4 + E] == 0) continue;   // nothing on this side of the block has a level yet: no offers (uniform)
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int j = tid + u * 256;
                if (j < L) h[j] = ((ring_ok >> (E * 3 + u)) & 1u) ? decode(raw_ring[E][u]) : MACRO_INF;
            }
            __syncthreads();
            macro_edge_scans(h, L, pre, suf, pm, sp);
            __syncthreads();
            // Offers of ring edge E to the four rim edges.  With E and e known at compile time (both loops are unrolled) the general form
            // min_j h[j] + max(|a - j|, d) = min(d + min of h over [a - d, a + d], a + pm[a - d - 1], sp[a + d + 1] - a)       (macro_offer)
            // - a = position along the ring edge, d = distance from its line - collapses to one to five LDS reads without a branch:
            //   the edge beside the rim (d = 1): three cells and the two far terms; the edge across (d = W): W + the edge's minimum, one number for all;
            //   an edge at right angles (a = 1 or a = W, d = 1 .. W): a prefix or suffix minimum and one far term.
            const int all_min = pre[L - 1];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = tid + u * 256;
                if (i < W) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        int off;
                        const bool horiz_E = E < 2, horiz_e = e < 2;
                        if (horiz_E == horiz_e) {
                            if (E == e) {            // beside: a = i + 1, d = 1
                                const int a = i + 1;
                                int m3 = h[a - 1] < h[a] ? h[a - 1] : h[a]; m3 = h[a + 1] < m3 ? h[a + 1] : m3;
                                off = m3 + 1;
                                if (a - 2 >= 0) { const int f = pm[a - 2] + a; off = f < off ? f : off; }
                                if (a + 2 <= L - 1) { const int f = sp[a + 2] - a; off = f < off ? f : off; }
                            } else off = all_min + W;   // across
                        } else {
                            // at right angles: the rim cell's coordinate ALONG edge e is i; its distance from ring edge E's line is i + 1 (E on the low side: 0, 2) or W - i
                            const int d = (E & 1) == 0 ? i + 1 : W - i;
                            if ((e & 1) == 0) {      // rim edge on the low side (top / left): the cell sits at a = 1 of ring edge E
                                off = d + pre[1 + d < L - 1 ? 1 + d : L - 1];
                                if (d + 2 <= L - 1) { const int f = sp[d + 2] - 1; off = f < off ? f : off; }
                            } else {                 // on the high side (bottom / right): a = W
                                off = d + suf[W - d > 0 ? W - d : 0];
                                if (W - d - 1 >= 0) { const int f = pm[W - d - 1] + W; off = f < off ? f : off; }
                            }
                        }
                        best[e][u] = off < best[e][u] ? off : best[e][u];
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = tid + u * 256;
                if (i < W) acc[e * AW + i] = best[e][u];
            }
        __syncthreads();
        const unsigned long long tq1 = dbg ? __builtin_readcyclecounter() : 0ull;
        // (a corner cell is on two rim edges: both accumulators hold offers for it, the smaller one is its level)
        if (tid < 4) {
            const int ea = tid < 2 ? 0 : 1, ia = (tid & 1) ? W - 1 : 0;     // corner tid: top-left, top-right, bottom-left, bottom-right
            const int eb = (tid & 1) ? 3 : 2, ib = tid < 2 ? 0 : W - 1;
            const int m = acc[ea * AW + ia] < acc[eb * AW + ib] ? acc[ea * AW + ia] : acc[eb * AW + ib];
            acc[ea * AW + ia] = m; acc[eb * AW + ib] = m;
        }
        __syncthreads();
        // rim cells that moved are stored; chg (the ring tables' space) keeps the new level of a rim cell that moved, +inf for one that did not
        int* const chg = h;   // [4][AW]
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = tid + u * 256;
                if (i < W) {
                    const int v = acc[e * AW + i];
                    const bool mv = v < old[e][u] && v < MACRO_INF;
                    if (mv) G[rim_idx(e, i)] = S(sizeof(S) == 2 && v > LVL_SAT ? LVL_SAT : v);
                    chg[e * AW + i] = mv ? v : MACRO_INF;
                }
            }
        __syncthreads();
        // Which tiles outside the block have something to gain (the engine's activation filter, tile_relax.hpp): ring cell j of edge E touches the rim cells
        // j - 2 .. j of rim edge E; it is in the queue (raw >= 0) and one of them moved to a level more than one below its own.  Necessary for an improvement,
        // so a flag that is withheld could not have led to a change; without the test neighbouring blocks woke each other up after every update.
#pragma unroll
        for (int E = 0; E < 4; E++)
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int j = tid + u * 256;
                bool gain = false;
                if (j < L && ((ring_ok >> (E * 3 + u)) & 1u) && raw_ring[E][u] >= 0) {
                    int best = MACRO_INF;
                    for (int i = j - 2; i <= j; i++) if (i >= 0 && i < W && chg[E * AW + i] < best) best = chg[E * AW + i];
                    gain = best + INC < decode(raw_ring[E][u]);
                }
                // (the ring cells j = 1 + 64 p .. 64 p + 64 lie in the tile at position p along the edge, j = 0 at position -1, j = L - 1 at position k: a wave's 64 cells
                // of one u span at most two positions)
                const int p = j == 0 ? -1 : (j - 1) / tilek::TS;
                const int p0 = __builtin_amdgcn_readfirstlane(p);
                const unsigned long long b_lo = __ballot(gain && p == p0), b_hi = __ballot(gain && p != p0);   // (one LDS atomic per wave and position, not one per cell)
                if ((tid & 63) == 0) {
                    if (b_lo != 0ull) atomicOr(&moved[E], 1 << (p0 + 1));
                    if (b_hi != 0ull) atomicOr(&moved[E], 1 << (p0 + 2));
                }
            }
        __syncthreads();
        const unsigned long long tq2 = dbg ? __builtin_readcyclecounter() : 0ull;
        // the tiles outside the block along each edge (positions -1 .. k: the corners' diagonal neighbours too) that can be improved
        if (tid < 4 * (k + 2)) {
            const int e = tid / (k + 2), j = tid % (k + 2) - 1;
            if (((moved[e] >> (j + 1)) & 1) && out_target >= 0) {
                if (atomicMax(&flags_next[tilek::entry_tile(uint32_t(out_target))], tilek::FLAG_HALO) == 0u) TL.pend[atomicAdd(&TL.npend, 1u)] = uint32_t(out_target);
            }
        }
        __syncthreads();
        if (dbg && tid == 0) {   // TDX_DEBUG_ROUNDS=1: cycles of the offers (loads + four ring edges), of the stores + gain test, of the activations
            const unsigned long long tq3 = __builtin_readcyclecounter();
            atomicAdd(dbg + 9, tq1 - tq0); atomicAdd(dbg + 10, tq2 - tq1); atomicAdd(dbg + 11, tq3 - tq2);
        }
        return 0;
    }
};
using LevelOp = LevelOpT<1, lvl_t>;
using ReachOp = LevelOpT<0, int32_t>;

// out[0] = max level, out[1] = #cells never reached by incfall, out[2] = max incrise level
template <class LV>
static __global__ __launch_bounds__(256) void flat_stats_kernel(const uint32_t* __restrict__ list, unsigned long long nq,
                                                                const LV* __restrict__ lvl, const LV* __restrict__ rq,
                                                                unsigned long long* __restrict__ out) {
    const unsigned long long base = (unsigned long long)blockIdx.x * (256 * 8) + threadIdx.x;
    int ml = 0, mr = 0;
    unsigned unv = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned long long q = base + (unsigned long long)i * 256;
        if (q < nq) {
            const size_t c = list[q];
            const int l = lvl[c], r = rq[c];
            ml = l > ml ? l : ml;
            mr = r > mr ? r : mr;
            unv += (l == 0);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int a = __shfl_xor(ml, off, 64), b = __shfl_xor(mr, off, 64);
        const unsigned u = __shfl_xor(unv, off, 64);
        ml = a > ml ? a : ml;
        mr = b > mr ? b : mr;
        unv += u;
    }
    // one atomic per block and counter, and none when the running maximum already covers this block
    __shared__ int s_ml[4], s_mr[4];
    __shared__ unsigned s_unv[4];
    const int w = int(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) { s_ml[w] = ml; s_mr[w] = mr; s_unv[w] = unv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { ml = s_ml[i] > ml ? s_ml[i] : ml; mr = s_mr[i] > mr ? s_mr[i] : mr; unv += s_unv[i]; }
        if ((unsigned long long)ml > __hip_atomic_load(out + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + 0, (unsigned long long)ml);
        if (unv) atomicAdd(out + 1, (unsigned long long)unv);
        if ((unsigned long long)mr > __hip_atomic_load(out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + 2, (unsigned long long)mr);
    }
}

// the same three numbers from a pass over the owned rows (cells outside the queue hold -1 in both fields): for a dense queue, where
// reading two int16 per cell costs less than building and gathering through a list of a third of the raster
template <class LV>
static __global__ __launch_bounds__(256) void flat_stats_stream_kernel(const LV* __restrict__ lvl, const LV* __restrict__ rq, size_t first, size_t count,
                                                                       unsigned long long* __restrict__ out) {
    const size_t base = first + size_t(blockIdx.x) * (256 * 16) + size_t(threadIdx.x);
    int ml = 0, mr = 0;
    unsigned unv = 0;
    LV l[16], r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {   // all loads first (clamped), a wave reads 128 contiguous bytes of each field per step
        const size_t c = base + size_t(i) * 256, cc = c < first + count ? c : first + count - 1;
        l[i] = lvl[cc]; r[i] = rq[cc];
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (base + size_t(i) * 256 < first + count) {
            ml = l[i] > ml ? int(l[i]) : ml;
            mr = r[i] > mr ? int(r[i]) : mr;
            unv += unsigned(l[i] == 0);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int a = __shfl_xor(ml, off, 64), b = __shfl_xor(mr, off, 64);
        const unsigned u = __shfl_xor(unv, off, 64);
        ml = a > ml ? a : ml;
        mr = b > mr ? b : mr;
        unv += u;
    }
    __shared__ int s_ml[4], s_mr[4];
    __shared__ unsigned s_unv[4];
    const int w = int(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) { s_ml[w] = ml; s_mr[w] = mr; s_unv[w] = unv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { ml = s_ml[i] > ml ? s_ml[i] : ml; mr = s_mr[i] > mr ? s_mr[i] : mr; unv += s_unv[i]; }
        if ((unsigned long long)ml > __hip_atomic_load(out + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + 0, (unsigned long long)ml);
        if (unv) atomicAdd(out + 1, (unsigned long long)unv);
        if ((unsigned long long)mr > __hip_atomic_load(out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + 2, (unsigned long long)mr);
    }
}

// the same, 8 cells per 16-byte load (first and count multiples of 8: whole rows of a raster whose width is one): the one-cell form spends its
// time on 2-byte load instructions, not on bytes
static __global__ __launch_bounds__(256) void flat_stats_stream8_kernel(const int16_t* __restrict__ lvl, const int16_t* __restrict__ rq, size_t first, size_t count,
                                                                        unsigned long long* __restrict__ out) {
    const size_t nvec = count / 8;
    const uint4* L = reinterpret_cast<const uint4*>(lvl + first);
    const uint4* R = reinterpret_cast<const uint4*>(rq + first);
    int ml = 0, mr = 0;
    unsigned unv = 0;
    // a few thousand blocks stride over the raster: the block's three numbers end in atomics on three words, and one word takes ~90 M atomics/s
    // (a block per 8 K cells was 32 768 additions to the pit counter: 0.36 ms of the pass's 0.38)
    for (size_t base = size_t(blockIdx.x) * (256 * 4) + size_t(threadIdx.x); base < nvec; base += size_t(gridDim.x) * (256 * 4)) {
        uint4 l[4], r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const size_t v = base + size_t(i) * 256, vc = v < nvec ? v : nvec - 1;
            l[i] = L[vc]; r[i] = R[vc];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (base + size_t(i) * 256 < nvec) {
                const unsigned lw[4] = {l[i].x, l[i].y, l[i].z, l[i].w}, rw[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int a = int(int16_t(lw[k] & 0xffffu)), b = int(int16_t(lw[k] >> 16));
                    const int c = int(int16_t(rw[k] & 0xffffu)), d = int(int16_t(rw[k] >> 16));
                    ml = a > ml ? a : ml; ml = b > ml ? b : ml;
                    mr = c > mr ? c : mr; mr = d > mr ? d : mr;
                    unv += unsigned(a == 0) + unsigned(b == 0);
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int a = __shfl_xor(ml, off, 64), b = __shfl_xor(mr, off, 64);
        const unsigned u = __shfl_xor(unv, off, 64);
        ml = a > ml ? a : ml;
        mr = b > mr ? b : mr;
        unv += u;
    }
    __shared__ int s_ml[4], s_mr[4];
    __shared__ unsigned s_unv[4];
    const int w = int(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) { s_ml[w] = ml; s_mr[w] = mr; s_unv[w] = unv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { ml = s_ml[i] > ml ? s_ml[i] : ml; mr = s_mr[i] > mr ? s_mr[i] : mr; unv += s_unv[i]; }
        if ((unsigned long long)ml > __hip_atomic_load(out + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + 0, (unsigned long long)ml);
        if (unv) atomicAdd(out + 1, (unsigned long long)unv);
        if ((unsigned long long)mr > __hip_atomic_load(out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out + 2, (unsigned long long)mr);
    }
}

// The inside of every macro block from its FINAL rim (see OPEN WATER above), one workgroup per ROW OF TILES of a block (fill_list[0 .. *fill_count): the first
// tile of each row, written by fill_list_kernel): level = min over the four rim edges of (distance to the edge + minimum of the rim over the 2 d + 1 cells
// across), by sparse tables built once per row of tiles.  int16 fields only (tables of uint16: a level fits, 0xFFFF = not reached).  Both fields in one launch
// (blockIdx.y).
struct FillArgs { int16_t* G; const uint32_t* remap; const uint8_t* blk_k; uint32_t* list; unsigned long long* count; };
static __global__ __launch_bounds__(256) void fill_list_kernel(FillArgs a0, FillArgs a1, int ntiles, int tiles_x) {
    const FillArgs a = blockIdx.y ? a1 : a0;
    const int t = blockIdx.x * 256 + threadIdx.x;
    bool on = false;
    if (t < ntiles) { const uint32_t rep = uint32_t(tilek::entry_tile(a.remap[t])); on = a.blk_k[rep] != 0 && (uint32_t(t) % uint32_t(tiles_x)) == (rep % uint32_t(tiles_x)); }
    const unsigned long long pos = tdxk::block_reserve(on ? 1u : 0u, a.count);
    if (on) a.list[pos] = uint32_t(t);
}
constexpr int FILL_LEVELS = 10;   // windows of up to 2^9 = 512 cells = the longest rim edge
static __global__ __launch_bounds__(256) void macro_fill_kernel(FillArgs a0, FillArgs a1, int nx, int tiles_x) {
    __shared__ uint16_t tab[4][FILL_LEVELS][MACRO_KMAX * 64];
    const FillArgs a = blockIdx.y ? a1 : a0;
    int16_t* __restrict__ G = a.G;
    const unsigned long long n = *a.count;
    const int tid = int(threadIdx.x), lx = tid & 63, wv = tid >> 6;
    for (unsigned long long it = blockIdx.x; it < n; it += gridDim.x) {
        const int t = int(a.list[it]), rep = tilek::entry_tile(a.remap[t]), k = int(a.blk_k[rep]);
        const int W = k * tilek::TS;
        const int X0 = (rep % tiles_x) * tilek::TS, Y0 = (rep / tiles_x) * tilek::TS;
        const int yt = (t / tiles_x) * tilek::TS - Y0;   // this row of tiles inside the block
        __syncthreads();   // (the previous row's tables are no longer read)
        for (int i = tid; i < W; i += 256) {
            const int16_t p = G[size_t(Y0) * nx + X0 + i], q = G[size_t(Y0 + W - 1) * nx + X0 + i], c = G[size_t(Y0 + i) * nx + X0], d = G[size_t(Y0 + i) * nx + X0 + W - 1];
            tab[0][0][i] = p > 0 ? uint16_t(p) : 0xFFFFu; tab[1][0][i] = q > 0 ? uint16_t(q) : 0xFFFFu;
            tab[2][0][i] = c > 0 ? uint16_t(c) : 0xFFFFu; tab[3][0][i] = d > 0 ? uint16_t(d) : 0xFFFFu;
        }
        for (int lev = 1; (1 << lev) <= W; lev++) {
            __syncthreads();
            const int half = 1 << (lev - 1), cnt = W - (1 << lev) + 1;
            for (int i = tid; i < 4 * cnt; i += 256) {
                const int e = i / cnt, j = i - e * cnt;
                const uint16_t p = tab[e][lev - 1][j], q = tab[e][lev - 1][j + half];
                tab[e][lev][j] = p < q ? p : q;
            }
        }
        __syncthreads();
        auto rmq = [&](int e, int lo, int hi) -> unsigned {
            lo = lo < 0 ? 0 : lo; hi = hi > W - 1 ? W - 1 : hi;
            const int lev = 31 - __builtin_clz(unsigned(hi - lo + 1));
            const unsigned p = tab[e][lev][lo], q = tab[e][lev][hi - (1 << lev) + 1];
            return p < q ? p : q;
        };
        for (int xt = 0; xt < W; xt += tilek::TS) {   // the tiles of the row
            const int x = xt + lx;
#pragma unroll 4
            for (int r = 0; r < 16; r++) {
                const int y = yt + wv * 16 + r;
                const int dt = y, db = W - 1 - y, dl = x, dr = W - 1 - x;
                unsigned v = rmq(0, x - dt, x + dt) + unsigned(dt), w2 = rmq(1, x - db, x + db) + unsigned(db);
                v = w2 < v ? w2 : v;
                w2 = rmq(2, y - dl, y + dl) + unsigned(dl); v = w2 < v ? w2 : v;
                w2 = rmq(3, y - dr, y + dr) + unsigned(dr); v = w2 < v ? w2 : v;
                G[size_t(Y0 + y) * nx + X0 + x] = v >= 0xFFFFu ? int16_t(0) : int16_t(v > unsigned(LVL_SAT) ? LVL_SAT : int(v));
            }
        }
    }
}

// What a flat iteration needs cleared before its classification, in ONE launch (they were five runtime fills): the stage counters, the activation flags, the
// per-tile "needs its masks" marks (tm_mode 0: none, 1: every tile, 2: the first half - the TDX_FLATS_MASKED hooks) and the count rings of both round schedules.
static __global__ __launch_bounds__(256) void prepare_kernel(unsigned long long* __restrict__ d_cnt, uint32_t* __restrict__ flags0, uint8_t* __restrict__ tmask, int ntiles,
                                                             int tm_mode, unsigned long long* __restrict__ countsA, unsigned long long* __restrict__ countsB,
                                                             uint8_t* __restrict__ notfull) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 8) d_cnt[t] = 0ull;
    if (t < ntiles) {
        flags0[t] = 0u; tmask[t] = uint8_t(tm_mode == 1 || (tm_mode == 2 && t < (ntiles + 1) / 2));
        if (notfull) { notfull[2 * t] = 0; notfull[2 * t + 1] = 0; }
    }
    if (t < 2 * tilek::COUNT_RING) { countsA[t] = 0ull; if (countsB) countsB[t] = 0ull; }
}

template <class LV>
static __global__ __launch_bounds__(256) void reset_q_kernel(const uint32_t* __restrict__ list, unsigned long long nq, LV* __restrict__ lvl,
                                                      LV* __restrict__ rq) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    lvl[list[q]] = 0;
    rq[list[q]] = 0;
}

template <class LV>
static __global__ __launch_bounds__(256) void overwrite_elev_kernel(size_t n, const LV* __restrict__ lvl, const LV* __restrict__ rq,
                                                             FlatLevels fl, float* __restrict__ Zout) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) Zout[i] = (float)flat_elev2<LV>(lvl[i], rq[i], fl);
}

// The next iteration's elevation only where it will be read: on the cells of the new flat queue and their 8 neighbours (flat
// cells are interior cells).  Several threads may store the same value to one cell.
template <class LV>
static __global__ __launch_bounds__(256) void overwrite_elev_sparse_kernel(const uint32_t* __restrict__ list, unsigned long long nq, int nx,
                                                                           const LV* __restrict__ lvl, const LV* __restrict__ rq, FlatLevels fl,
                                                                           float* __restrict__ Zout) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const size_t c = list[q];
    int l[9], r[9];
#pragma unroll
    for (int k = 0; k <= 8; k++) {
        const size_t n = k ? size_t(ptrdiff_t(c) + ptrdiff_t(d2(k)) * nx + d1(k)) : c;
        l[k] = lvl[n]; r[k] = rq[n];
    }
#pragma unroll
    for (int k = 0; k <= 8; k++) {
        const size_t n = k ? size_t(ptrdiff_t(c) + ptrdiff_t(d2(k)) * nx + d1(k)) : c;
        Zout[n] = (float)flat_elev2<LV>(l[k], r[k], fl);
    }
}

// markers of the cells of the PREVIOUS queue back to "not in Q"
template <class LV>
static __global__ __launch_bounds__(256) void unmark_q_kernel(const uint32_t* __restrict__ list, unsigned long long nq, LV* __restrict__ lvl,
                                                              LV* __restrict__ rq) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    lvl[list[q]] = -1;
    rq[list[q]] = -1;
}

}  // namespace flatk

static inline int flats_read_counters(tdx_context* ctx, int nwords) {
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail, size_t(nwords) * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}

template <class LV>
static inline int flats_reset_markers(tdx_context* ctx, const Strip& st, const uint32_t* qlist, unsigned long long nq, LV* lvl, LV* rq) {
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    TDX_HIP_CHECK(ctx, hipMemsetAsync(lvl, 0xFF, n * sizeof(LV), ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemsetAsync(rq, 0xFF, n * sizeof(LV), ctx->stream));
    if (nq) hipLaunchKernelGGL((flatk::reset_q_kernel<LV>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, ctx->stream, qlist, nq, lvl, rq);
    int rc = strip_exchange<LV>(ctx, st, lvl, LV(-1));   // queue membership of the neighbours' boundary rows
    if (rc != TDX_OK) return rc;
    return strip_exchange<LV>(ctx, st, rq, LV(-1));
}

// Same markers as flats_reset_markers when lvl / rq are "not in Q" everywhere except on the cells of the previous queue `qold`
// (which is what a flat iteration leaves behind): a later iteration of a few thousand cells does not rewrite two rasters.
template <class LV>
static inline int flats_reset_markers_after(tdx_context* ctx, const Strip& st, const uint32_t* qold, unsigned long long nq_old, const uint32_t* qlist,
                                            unsigned long long nq, LV* lvl, LV* rq) {
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    if (nq_old > n / 16) return flats_reset_markers(ctx, st, qlist, nq, lvl, rq);
    if (nq_old) hipLaunchKernelGGL((flatk::unmark_q_kernel<LV>), dim3(tdx_blocks_for(nq_old, 256)), dim3(256), 0, ctx->stream, qold, nq_old, lvl, rq);
    if (nq) hipLaunchKernelGGL((flatk::reset_q_kernel<LV>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, ctx->stream, qlist, nq, lvl, rq);
    int rc = strip_exchange<LV>(ctx, st, lvl, LV(-1));
    if (rc != TDX_OK) return rc;
    return strip_exchange<LV>(ctx, st, rq, LV(-1));
}

// elevDEM := (float)elev2 where the next iteration (queue `qlist`) reads it: the queue cells and their neighbours
template <class LV>
static inline int flats_overwrite_elevation_sparse(tdx_context* ctx, int nx, const uint32_t* qlist, unsigned long long nq, const LV* lvl, const LV* rq,
                                                   FlatLevels fl, float* zout) {
    TdxSpan sp(ctx, TDX_K_MISC);
    if (nq) hipLaunchKernelGGL((flatk::overwrite_elev_sparse_kernel<LV>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, ctx->stream, qlist, nq, nx, lvl, rq, fl, zout);
    return TDX_OK;
}

template <class LV>
static inline int flats_overwrite_elevation(tdx_context* ctx, size_t n, const LV* lvl, const LV* rq, FlatLevels fl, float* zout) {
    TdxSpan sp(ctx, TDX_K_MISC);
    hipLaunchKernelGGL((flatk::overwrite_elev_kernel<LV>), dim3(tdx_blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, lvl, rq, fl, zout);
    return TDX_OK;
}

// One level field to its GLOBAL fixed point: relax the strip with frozen halo rows, exchange boundary
// rows, re-activate the tiles that see a changed halo cell, until no halo cell changed on any rank
// (the per-level share() + MPI_Allreduce of src/d8.cpp:549-550,620-630, once per strip crossing instead
// of once per level).
template <class Op = flatk::LevelOp>
// eager_rounds > 0 (multi-strip only): at most that many rounds between two exchanges instead of strip-local fixed points - for fields whose fronts are many
// and independent (reach_closure); a single front (a lake's level field) gains nothing and would only pay for the extra exchanges.
static inline int flats_relax_field(tdx_context* ctx, const Strip& st, tilek::TileGeom geom, typename Op::Raw* field, const uint8_t* mask, tilek::Sched sc,
                                    int64_t* rounds, int64_t* launches, const uint8_t* tile_masked = nullptr, int eager_rounds = 0) {
    Op op{field, mask};
    if constexpr (Op::kHasPlain) op.TM = tile_masked;
    for (;;) {
        bool left = false;
        int rc = (st.multi() && eager_rounds > 0) ? tile_relax_run_bounded(ctx, op, geom, sc, eager_rounds, &left, rounds, launches) : tile_relax_run(ctx, op, geom, sc, rounds, launches);
        if (rc != TDX_OK) return rc;
        if (!st.multi()) return TDX_OK;
        int64_t changed = 0;
        rc = strip_exchange<typename Op::Raw>(ctx, st, field, typename Op::Raw(-1), sc.flags, geom.tiles_x, &changed, true, left ? 1 : 0);   // halo exchange + the termination vote in one step
        if (rc != TDX_OK) return rc;
        if (changed == 0) return TDX_OK;
    }
}

// Runs classification + both level relaxations for the flat queue `qlist` (cells of this strip); on return
// lvl/rq hold the levels (halo rows included) and *out the sweep counts of the reference's loops.
// `stream_classify` (optional): replaces the marker reset + list-based classification by one streaming pass over the
// whole strip that writes lvl / rq / both masks of EVERY owned cell and raises the tile flags; qlist may then be null (no list was
// built for a dense queue): the level statistics come from a pass over the owned rows instead.
// (notfull: two bytes per tile, set to 1 where the tile is NOT full for incfall [2 t] / incrise [2 t + 1] - see OPEN WATER; nullptr: not wanted)
using StreamClassifyFn = std::function<void(const tilek::TileGeom&, uint8_t* fmask, uint8_t* rmask, uint32_t* tile_flags, uint8_t* tile_masked, uint8_t* notfull)>;
template <class Traits, class LV>
static int flats_bfs(tdx_context* ctx, Traits tr, const float* Z, const Strip& st, const uint32_t* qlist, unsigned long long nq,
                     FlatBuffersT<LV> b, FlatLevels* out, tdx_stats* stats, const StreamClassifyFn* stream_classify = nullptr, int iteration = 1) {
    using LOp = flatk::LevelOpT<1, LV>;
    const FlatPhases& ph = flat_phases(iteration);
    ctx->phase = ph.classify;
    hipStream_t s = ctx->stream;
    const int nx = st.nx;
    const size_t n = size_t(nx) * size_t(st.ny_arr);
    const tilek::TileGeom geom = tilek::make_geom(nx, st.ny_arr, st.y0, st.y1);
    const int ntiles = geom.tiles_x * geom.tiles_y;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    uint8_t* fmask = static_cast<uint8_t*>(ctx->scratch(TDX_S_E, n));
    uint8_t* rmask = static_cast<uint8_t*>(ctx->scratch(TDX_S_F, n));
    uint32_t* flags0 = static_cast<uint32_t*>(ctx->scratch(TDX_S_G, size_t(ntiles) * 4 * 2 + size_t(ntiles)));
    uint32_t* list = static_cast<uint32_t*>(ctx->scratch(TDX_S_H, size_t(ntiles) * 4 * tilek::SCHED_LIST_WORDS));
    unsigned long long* counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_K, size_t(tilek::COUNT_RING) * 16));
    if (!fmask || !rmask || !flags0 || !list || !counts) return TDX_ERR_NOMEM;
    uint32_t* flags = flags0 + ntiles;
    uint8_t* tmask = reinterpret_cast<uint8_t*>(flags0 + 2 * size_t(ntiles));   // incfall: tiles that are not plain
    const char* e_masked = getenv("TDX_FLATS_MASKED");   // A/B and test hook (read per call): 1 = every tile on the masked form, 2 = the first half of the tiles
    const bool no_plain = e_masked && e_masked[0] == '1';
    const bool half_plain = e_masked && e_masked[0] == '2';
    TdxSpan sp(ctx, TDX_K_BFS);
    static const bool no_pair = getenv("TDX_FLATS_SEQUENTIAL") != nullptr;
    static const bool strips_sequential = getenv("TDX_FLATS_STRIPS_SEQUENTIAL") != nullptr;   // (A/B hook: one field after the other in a multi-strip run)
    const bool pair = !(st.multi() && strips_sequential) && !ctx->kernel_timing && !no_pair;
    // second schedule of the pair (own flags / list / counts)
    uint32_t* flagsB = pair ? static_cast<uint32_t*>(ctx->scratch(TDX_S_L, size_t(ntiles) * 4 * (1 + tilek::SCHED_LIST_WORDS))) : nullptr;
    unsigned long long* countsB = pair ? static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16)) : nullptr;
    if (pair && (!flagsB || !countsB)) return TDX_ERR_NOMEM;
    // OPEN WATER (macro blocks of full tiles, see above): with the streaming classification (a dense queue: the first iteration of D8FlowDir and of DinfFlowDir), int16
    // fields, the register tile kernel.  (From the LIST classification - later iterations - per-tile "full" marks were built and measured in round 6: a wave's 64
    // list entries span three or four tiles, the marks cost more than the rounds they save - DinfFlowDir 24.2 -> 32.6 ms at 16384^2, profiles/r06l_macro_dinf_ab.txt;
    // with the streaming classification DinfFlowDir gains instead: 24.3 -> 22.3 ms, profiles/r06ae_*.)  TDX_FLATS_MACRO=0 switches it off, =2 / 4 / 8 sets the largest
    // block edge (read per call: A/B and test hook).
    int macro_k = 0;
    if constexpr (sizeof(LV) == 2) {
        const char* e_macro = getenv("TDX_FLATS_MACRO");
        macro_k = e_macro ? atoi(e_macro) : flatk::MACRO_KMAX;
        macro_k = macro_k >= 8 ? 8 : (macro_k >= 4 ? 4 : (macro_k >= 2 ? 2 : 0));
        if (!stream_classify || getenv("TDX_FLATS_FUSED") != nullptr || getenv("TDX_RELAX_LDS") != nullptr || no_plain || half_plain) macro_k = 0;
    }
    uint8_t* notfull = nullptr; uint32_t *remapF = nullptr, *remapR = nullptr, *fill_list = nullptr; uint8_t *blkF = nullptr, *blkR = nullptr;
    if (macro_k) {
        // [notfull 2 nt][blkF nt][blkR nt][remapF 4 nt][remapR 4 nt][fill_list 2 x 4 nt]
        uint8_t* m = static_cast<uint8_t*>(ctx->scratch(TDX_S_MACRO, size_t(ntiles) * 20 + 64));
        if (!m) return TDX_ERR_NOMEM;
        notfull = m; blkF = m + 2 * size_t(ntiles); blkR = blkF + ntiles;
        remapF = reinterpret_cast<uint32_t*>(m + 4 * size_t(ntiles)); remapR = remapF + ntiles; fill_list = remapR + ntiles;
    }
    hipLaunchKernelGGL(flatk::prepare_kernel, dim3(tdx_blocks_for(size_t(std::max(ntiles, 2 * tilek::COUNT_RING)), 256)), dim3(256), 0, s, d_cnt, flags0, tmask, ntiles,
                       no_plain ? 1 : (half_plain ? 2 : 0), counts, countsB, notfull);
    if (stream_classify) (*stream_classify)(geom, fmask, rmask, flags0, tmask, notfull);
    else {
        // The list classification writes the masks of the QUEUE's cells only.  The register tile kernel reads a mask byte only where the level marker says
        // "in the queue" (LevelOpT::raw_can_move), so what other cells hold - a mask of an earlier iteration, or nothing yet - is never looked at; only the
        // LDS-resident tile kernel and the worklist schedule (test hooks), which decode masks without the marker, need the two rasters cleared (read per call: the hooks are).
        const bool clear_masks = getenv("TDX_RELAX_LDS") != nullptr || getenv("TDX_RELAX_ASYNC") != nullptr || getenv("TDX_FLATS_CLEAR_MASKS") != nullptr;
        if (clear_masks) {
            TDX_HIP_CHECK(ctx, hipMemsetAsync(fmask, 0, n, s));
            TDX_HIP_CHECK(ctx, hipMemsetAsync(rmask, 0, n, s));
        }
        if (nq)
            hipLaunchKernelGGL((flatk::classify_kernel<Traits, LV>), dim3(tdx_blocks_for(nq, 256 * flatk::CLASSIFY_ITEMS)), dim3(256), 0, s, tr, Z, nx,
                               geom.tiles_x, qlist, nq, b.lvl, b.rq, fmask, rmask, flags0, tmask);
    }
    int rc = strip_exchange<LV>(ctx, st, b.lvl, LV(-1));   // the neighbours' seeds
    if (rc != TDX_OK) return rc;
    ctx->phase = ph.levels;
    rc = strip_exchange<LV>(ctx, st, b.rq, LV(-1));
    if (rc != TDX_OK) return rc;
    int64_t launches = 1, rounds_fall = 0, rounds_rise = 0;
    tilek::TileGeom geomF = geom, geomR = geom;
    if (macro_k) {
        // blocks of full tiles per field; in a multi-strip run the tile rows that hold the first / last owned row stay ordinary tiles (the halo exchange flags them directly)
        const int y_lo = st.multi() ? st.y0 + 1 : st.y0, y_hi = st.multi() ? st.y1 - 1 : st.y1;
        const unsigned nreg = unsigned(((geom.tiles_x + 7) / 8) * ((geom.tiles_y + 7) / 8));
        hipLaunchKernelGGL(flatk::find_blocks_kernel, dim3((nreg + 63) / 64, 2), dim3(64), 0, s, notfull, geom.tiles_x, geom.tiles_y, nx, y_lo, y_hi, macro_k, remapF, blkF, remapR, blkR);
        geomF.remap = remapF; geomF.blk_k = blkF;
        geomR.remap = remapR; geomR.blk_k = blkR;
    }
    if (pair) {
        // the two level fields are independent: relax them side by side on two streams (own flags / list / counts each)
        uint32_t* listB = flagsB + ntiles;
        // two round schedules on two HIP streams; TDX_FLATS_FUSED=1 (read per call): one launch per round for both fields on one stream - 0.5 ms slower at 16384^2
        // (a round ends when the slower field's does), but independent of how the runtime schedules two streams
        const bool two_streams = getenv("TDX_FLATS_FUSED") == nullptr || getenv("TDX_RELAX_LDS") != nullptr;
        // round 0 of both schedules from the classification's flags in one launch (two streams); the fused form starts its runners itself
        const uint32_t* start_flags = two_streams ? flags0 : nullptr;
        if (!two_streams) {
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(flags, flags0, size_t(ntiles) * 4, hipMemcpyDeviceToDevice, s));
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(flagsB, flags0, size_t(ntiles) * 4, hipMemcpyDeviceToDevice, s));
        }
        // Multi-strip: both fields to their strip-local fixed points side by side, then BOTH boundary rows are exchanged and one vote decides - the outer
        // rounds are those of the deeper field instead of the sum of both, and the shallower field's relaxation hides behind the deeper one's
        // (profiles/r05b_*: incrise's 6.7 + 1.5 ms of the critical path at BASELINE.json configs[3] ran after incfall's 27 ms).
        for (;;) {
            rc = two_streams ? tile_relax_run_pair(ctx, LOp{b.lvl, fmask, tmask}, tilek::Sched{flags, list, counts},
                                                   LOp{b.rq, rmask, no_plain ? tmask : nullptr}, tilek::Sched{flagsB, listB, countsB}, geomF, &rounds_fall, &launches, start_flags, &geomR)
                             : tile_relax_run_fused(ctx, LOp{b.lvl, fmask, tmask}, tilek::Sched{flags, list, counts},
                                                    LOp{b.rq, rmask, no_plain ? tmask : nullptr}, tilek::Sched{flagsB, listB, countsB}, geom, &rounds_fall, &launches);
            if (rc != TDX_OK) return rc;
            start_flags = nullptr;   // (later outer rounds start from what the exchanges flagged)
            if (!st.multi()) break;
            int64_t ch_fall = 0, ch_rise = 0;
            rc = strip_exchange<LV>(ctx, st, b.lvl, LV(-1), flags, geom.tiles_x, &ch_fall);
            if (rc != TDX_OK) return rc;
            rc = strip_exchange<LV>(ctx, st, b.rq, LV(-1), flagsB, geom.tiles_x, &ch_rise);
            if (rc != TDX_OK) return rc;
            int64_t changed = ch_fall + ch_rise;
            rc = strip_allreduce(ctx, st, &changed, 1, TDX_OP_SUM);
            if (rc != TDX_OK) return rc;
            if (changed == 0) break;
        }
    } else {
        // ---- incfall ---- (with macro blocks: only a block's first tile starts active)
        const unsigned cg = tdx_blocks_for(size_t(ntiles), 256);
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(flags, flags0, size_t(ntiles) * 4, hipMemcpyDeviceToDevice, s));
        if (macro_k) hipLaunchKernelGGL(flatk::block_flags_kernel, dim3(cg), dim3(256), 0, s, flags, remapF, ntiles);
        rc = flats_relax_field<LOp>(ctx, st, geomF, b.lvl, fmask, tilek::Sched{flags, list, counts}, &rounds_fall, &launches, tmask);
        if (rc != TDX_OK) return rc;
        // ---- incrise ----
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(flags, flags0, size_t(ntiles) * 4, hipMemcpyDeviceToDevice, s));
        if (macro_k) hipLaunchKernelGGL(flatk::block_flags_kernel, dim3(cg), dim3(256), 0, s, flags, remapR, ntiles);
        rc = flats_relax_field<LOp>(ctx, st, geomR, b.rq, rmask, tilek::Sched{flags, list, counts}, &rounds_rise, &launches, no_plain ? tmask : nullptr);
        if (rc != TDX_OK) return rc;
    }
    if constexpr (sizeof(LV) == 2) {
        if (macro_k) {   // the inside of the macro blocks from their final rims (both fields; d_cnt[5], [6]: the numbers of tiles to fill, cleared by prepare_kernel)
            const unsigned cg = tdx_blocks_for(size_t(ntiles), 256);
            const flatk::FillArgs aF{reinterpret_cast<int16_t*>(b.lvl), remapF, blkF, fill_list, d_cnt + 5}, aR{reinterpret_cast<int16_t*>(b.rq), remapR, blkR, fill_list + ntiles, d_cnt + 6};
            hipLaunchKernelGGL(flatk::fill_list_kernel, dim3(cg, 2), dim3(256), 0, s, aF, aR, ntiles, geom.tiles_x);
            hipLaunchKernelGGL(flatk::macro_fill_kernel, dim3(unsigned(std::min(ntiles, 2 * ctx->num_cus)), 2), dim3(256), 0, s, aF, aR, nx, geom.tiles_x);
        }
    }
    ctx->phase = ph.stats;
    // (the stage counters were cleared by prepare_kernel and nothing of this function has touched them since)
    if (nq && qlist) hipLaunchKernelGGL((flatk::flat_stats_kernel<LV>), dim3(tdx_blocks_for(nq, 2048)), dim3(256), 0, s, qlist, nq, b.lvl, b.rq, d_cnt);
    else if (nq) {   // no list (dense first queue of D8FlowDir): one pass over the owned rows
        const size_t first = size_t(st.y0) * size_t(nx), count = size_t(st.y1 - st.y0) * size_t(nx);
        if constexpr (sizeof(LV) == 2) {
            if (first % 8 == 0 && count % 8 == 0)
                hipLaunchKernelGGL(flatk::flat_stats_stream8_kernel, dim3(std::min(tdx_blocks_for(count / 8, 256 * 4), 2048u)), dim3(256), 0, s, b.lvl, b.rq, first, count, d_cnt);
            else
                hipLaunchKernelGGL((flatk::flat_stats_stream_kernel<LV>), dim3(tdx_blocks_for(count, 256 * 16)), dim3(256), 0, s, b.lvl, b.rq, first, count, d_cnt);
        } else
            hipLaunchKernelGGL((flatk::flat_stats_stream_kernel<LV>), dim3(tdx_blocks_for(count, 256 * 16)), dim3(256), 0, s, b.lvl, b.rq, first, count, d_cnt);
    }
    rc = flats_read_counters(ctx, 3);
    if (rc != TDX_OK) return rc;
    int64_t mx[2] = {int64_t(ctx->h_mail[0]), int64_t(ctx->h_mail[2])}, unvisited = int64_t(ctx->h_mail[1]);
    rc = strip_allreduce(ctx, st, mx, 2, TDX_OP_MAX);
    if (rc != TDX_OK) return rc;
    rc = strip_allreduce(ctx, st, &unvisited, 1, TDX_OP_SUM);
    if (rc != TDX_OK) return rc;
    const int L = int(mx[0]), Qmax = int(mx[1]);
    // int16 fields (the reference's layout, src/d8.cpp:483,486) hold levels up to 32 766; a deeper flat - where the reference's short counters wrap -
    // makes the caller start over on int32 fields (TDX_FLATS_TOO_DEEP; every rank sees the same maxima, so every rank does).  TDX_LEVELS_LIMIT: test hook.
    if (sizeof(LV) == 2) {
        static const int limit = getenv("TDX_LEVELS_LIMIT") ? std::max(2, std::min(atoi(getenv("TDX_LEVELS_LIMIT")), 32767)) : 32767;
        if (L >= limit || Qmax >= limit) { ctx->abort_call(); return TDX_FLATS_TOO_DEEP; }
    } else if (L >= 0x3ffffff0 || Qmax >= 0x3ffffff0)
        return tdx_fail(ctx, TDX_ERR_ARG, "flat resolution deeper than 2^30 levels");
    out->T = (L == 1 && unvisited == 0) ? 1 : ((L > 1 ? L : 1) + 1);
    out->has_pits = unvisited > 0 ? 1 : 0;
    out->Tr = Qmax + 1;
    if (stats) {
        stats->levels_fall += L;
        stats->levels_rise += Qmax;
        stats->levels_fall_max = std::max<int64_t>(stats->levels_fall_max, L);
        stats->levels_rise_max = std::max<int64_t>(stats->levels_rise_max, Qmax);
        stats->launches[TDX_K_BFS] += launches + 1;
        stats->rounds += rounds_fall + rounds_rise;
    }
    return TDX_OK;
}

// ---- upstream closure (outlets) ---------------------------------------------------------------------------------
// reach[c] = 1 for the seed cells on entry (0 elsewhere); mask[c] selects the neighbours c drains to.  On return
// reach == 1 exactly on the cells from which a seed is reachable downstream (src/commonLib.cpp:285-385 computes the
// same set by a level-synchronous upstream search, one MPI round per level).  `flags` must hold FLAG_FULL for the
// tiles that contain seeds.  Works across strips (boundary rows of reach are exchanged between outer rounds).
static inline int reach_closure(tdx_context* ctx, const Strip& st, int32_t* reach, const uint8_t* mask, uint32_t* flags, uint32_t* list,
                                unsigned long long* counts, int64_t* rounds, int64_t* launches) {
    // halo rows start at 0: the first exchange after the local pass brings in the neighbours' marks and flags the tiles
    const tilek::TileGeom geom = tilek::make_geom(st.nx, st.ny_arr, st.y0, st.y1);
    // (the closure of many outlets spreads upstream along every tributary at once: bounded rounds between the exchanges; TDX_REACH_EAGER_ROUNDS=0: fixed points)
    const int eager = getenv("TDX_REACH_EAGER_ROUNDS") ? std::max(0, atoi(getenv("TDX_REACH_EAGER_ROUNDS"))) : 32;
    return flats_relax_field<flatk::ReachOp>(ctx, st, geom, reach, mask, tilek::Sched{flags, list, counts}, rounds, launches, nullptr, eager);
}
