// Tile relaxation engine shared by PitRemove (minimax-path surface) and flat resolution (the two
// breadth-first level fields of resolveflats).  All three are fixed points of a monotone, order-free
// neighbourhood operator
//       v(c) <- apply(cst(c), v(c), min over the neighbours selected by mask(c) of v(n))
// that only ever DEcreases v from +inf, so any schedule that runs to convergence yields the same
// values.  The schedule used on gfx950:
//
//   * the raster is cut into 64x64 tiles; a 256-thread workgroup loads a tile of v plus a one-cell halo, every
//     lane owns a 16-row column segment and relaxes it in place ("chaotic" relaxation) in alternating
//     downward / upward sweeps, looking only at rows in which something can have moved, until a sweep
//     moves nothing.  Two tile kernels implement this: relax_tile_reg (default: the segment lives in
//     VGPRs, neighbour columns come from neighbour LANES through DPP wave shifts, LDS only carries the
//     rows that cross waves) and relax_tile (the tile lives in LDS, 66x67 x 4 B = 17.7 KB; TDX_RELAX_LDS=1);
//   * changed cells are written back; a tile whose rim changed raises the activation flag of the
//     neighbouring tiles that see that rim in their halo;
//   * rounds (ONE launch each: relax the tiles of this round's list; the workgroup that first raises a
//     tile's flag appends it to the next round's list) are enqueued in batches with NO host round trip in
//     between: the list length lives in device memory and the resident workgroups pull tiles from a
//     per-round cursor; the host reads back one batch of per-round counts at a time and stops at the
//     first empty round.
//
// HBM traffic per activation of a tile: one tile image of v (+ the per-cell constant) in, changed
// cells out.  Critical path: (longest dependency path measured in tiles) rounds x one launch.
#pragma once
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <type_traits>

#include "context.hpp"
#include "device_common.hpp"

namespace tilek {
using namespace tdxk;

constexpr int TS = 64;          // tile edge
constexpr int LH = TS + 2;      // tile + halo edge
constexpr int LP = LH + 1;      // LDS row pitch: odd, so that column-strided accesses are conflict-free
constexpr int RPW = 16;         // rows per lane
constexpr int NWAVE = TS / RPW; // waves per tile
constexpr int NTHR = NWAVE * 64;
constexpr int COUNT_RING = 1024;
constexpr int MAX_SWEEPS = 8;
constexpr int PULL_MAX = 64;    // tiles a workgroup takes from the round's list per cursor atomic (large rounds)

struct TileGeom {
    int nx, ny;             // raster (strip incl. halo rows) size
    int tiles_x, tiles_y;
    int y_own0, y_own1;     // rows [y_own0, y_own1) may be updated; others are read-only halo rows
    int max_sweeps;         // sweeps per activation before a tile yields (it re-activates itself)
    int chain_max;          // solo rounds: tile hand-overs inside one launch (0 = one tile per launch; TDX_SOLO_CHAIN)
    // Where a round reports its size to the host (set per batch by RoundRunner::enqueue; null = nobody listens): the first workgroup of the round
    // launched with the counter `count` stores count[0] to cnt_host[count - cnt_dev] - pinned host memory - so that the host needs no copy
    // kernel between two batches of rounds (61 of them per 16384^2 pipeline step, ~6 us each: profiles/r04m_timeline_*).
    // The report carries the batch's sequence number (cnt_tag, upper 32 bits; the count is below 2^32): a batch of empty rounds that is still in flight
    // when its runner ends writes into the same host slots as the next runner's first batch, and the host must not take those late zeros for its own.
    unsigned long long* cnt_host;
    const unsigned long long* cnt_dev;
    unsigned long long cnt_tag;
    int act_filter;         // 1: a moved rim cell raises a neighbour's flag only if it can improve a cell of it (relax_tile_reg; TDX_ACT_FILTER_OFF=1: every moved rim cell does)
    // MACRO BLOCKS (optional; flats.hpp: open water of a level field).  An aligned K x K block of tiles that an operator can solve in closed form is ONE node
    // of the schedule: every activation of one of its tiles is redirected to its first tile - remap[t] = that tile's index + log2 K in bits 27-28 (a list entry:
    // entry_tile / entry_blk_k) for EVERY tile of a block, t itself elsewhere - and the entries of blocks are served by Op::macro_update (macro_role) instead of
    // the tile kernel.  blk_k[t] = K for a block's first tile, zero elsewhere.  nullptr: no blocks.
    const uint32_t* remap;
    const uint8_t* blk_k;
};

static inline TileGeom make_geom(int nx, int ny, int y_own0, int y_own1) {
    TileGeom g;
    g.nx = nx; g.ny = ny;
    g.tiles_x = (nx + TS - 1) / TS; g.tiles_y = (ny + TS - 1) / TS;
    g.y_own0 = y_own0; g.y_own1 = y_own1;
    static const int ms = getenv("TDX_MAX_SWEEPS") ? atoi(getenv("TDX_MAX_SWEEPS")) : MAX_SWEEPS;
    g.max_sweeps = ms;
    static const int cm = getenv("TDX_SOLO_CHAIN") ? atoi(getenv("TDX_SOLO_CHAIN")) : 256;
    g.chain_max = cm;
    g.cnt_host = nullptr; g.cnt_dev = nullptr; g.cnt_tag = 0ull;
    static const int af = getenv("TDX_ACT_FILTER_OFF") ? 0 : 1;
    g.act_filter = af;
    g.remap = nullptr; g.blk_k = nullptr;
    return g;
}

// masked minimum of the 8 neighbours held in registers; bit k-1 of mask selects neighbour k
// (1 E, 2 NE, 3 N, 4 NW, 5 W, 6 SW, 7 S, 8 SE: src/commonLib.h:83-84)
template <class T>
__device__ __forceinline__ T masked_min(unsigned mask, T inf, T e, T ne, T n, T nw, T w, T sw, T s, T se) {
    T m = inf;
    if (mask & 1u) m = e < m ? e : m;
    if (mask & 2u) m = ne < m ? ne : m;
    if (mask & 4u) m = n < m ? n : m;
    if (mask & 8u) m = nw < m ? nw : m;
    if (mask & 16u) m = w < m ? w : m;
    if (mask & 32u) m = sw < m ? sw : m;
    if (mask & 64u) m = s < m ? s : m;
    if (mask & 128u) m = se < m ? se : m;
    return m;
}

// all 8 (or the 4 cardinal) neighbours, no per-cell mask: 4 / 2 three-input minima
template <class T>
__device__ __forceinline__ T uniform_min(bool fourway, T e, T ne, T n, T nw, T w, T sw, T s, T se) {
    T m = e < n ? e : n;
    const T m2 = w < s ? w : s;
    m = m2 < m ? m2 : m;
    if (!fourway) {
        T d = ne < nw ? ne : nw;
        const T d2 = sw < se ? sw : se;
        d = d2 < d ? d2 : d;
        m = d < m ? d : m;
    }
    return m;
}

// Op interface:
//   using T;  (4 bytes)  static T inf();
//   Raw load_raw(size_t idx) const; static T decode(Raw);   value of an in-grid cell (load and decode are split so
//                                        that all loads of a tile can be issued back to back)
//   void store(size_t idx, T v) const;   changed cell
//   CellRaw cell_raw(size_t idx) const; static void cell_decode(CellRaw, T& cst, unsigned& mask);
//                                        per-cell constant + neighbour mask (0 = never updated)
//   static T cell_floor(T cst, T v);     register tiles only: the constant apply() gets, given the value v the cell is loaded with
//                                        (lets an operator fold "a settled cell never moves" into the constant)
//   static T apply(T cst, T own, T m);   new value (must be <= own)
//   static bool settled(T cst, T v);     v can never decrease again
//   static constexpr int kUniform;       0: per-cell masks; 8 / 4: every updatable cell looks at all 8 / the 4 cardinal neighbours
//
// In-tile schedule: alternating downward / upward sweeps in which a lane owns a 16-row column segment
// (lanes of a wave = 64 consecutive columns, so LDS rows are read conflict-free).  The kernel is
// VALU-issue bound, so cells are only re-evaluated when they can move: every lane keeps a 16-bit DIRTY
// mask of its rows - a row is dirty when the cell itself or one of its 8 neighbours changed since it
// was last evaluated (own column: updated on the fly, so a value runs down/up a whole segment in one
// sweep; neighbour lanes: one cross-lane shift per sweep; neighbour waves: two ballots through LDS).
// Rows that no lane of the wave has marked dirty cost one scalar branch.  The tile is done after a
// sweep in which no cell moved; after MAX_SWEEPS it is written back and re-activates itself, so one
// slow tile cannot hold up a round.

__device__ __forceinline__ unsigned mask_at(const unsigned (&pk)[RPW / 4], int r) { return (pk[r >> 2] >> (8 * (r & 3))) & 0xFFu; }

struct TileLds {   // LDS of one workgroup
    int rim;
    unsigned long long top[2][NWAVE], bot[2][NWAVE];   // columns whose first / last segment row changed, per wave
    unsigned any[2][NWAVE];                            // per wave: did any cell move in this sweep
    unsigned rows[NWAVE];                              // per wave: OR of the lanes' dirty masks
    unsigned next;
    unsigned long long base;
    unsigned npend;                   // tiles activated by this workgroup and not yet appended to the next round's list
    int chain;                        // solo rounds: the one tile this activation hands over to (bit 31: with FLAG_FULL), -1 = none / several
    uint32_t pulled[PULL_MAX];        // the list entries of the current pull (fetched together: one memory latency per pull, not one per tile)
    uint32_t pend[PULL_MAX * 9];
};

// Activation flag of a tile.  FLAG_HALO: only cells of its halo ring moved since the tile last reached its local
// fixed point, so only its perimeter cells need a first look (interior cells see nothing but cells of the tile
// itself, which nobody else writes).  FLAG_FULL: look at every cell (first activation, capped activation, or a
// strip halo ROW inside the tile's area was rewritten by the exchange).
constexpr uint32_t FLAG_HALO = 1u, FLAG_FULL = 2u;
// A list entry / an activation target is a tile index (below 2^27: a strip holds at most 2^32 cells = 2^20 tiles of 64 x 64, 2^24 of 16 x 16) and, for the first
// tile of a MACRO BLOCK (TileGeom::remap), the block's edge as log2 K in bits 27-28: who looks at an entry knows without a further load whether it is a block's.
constexpr uint32_t TILE_IDX_MASK = 0x07ffffffu;
constexpr int TILE_BLK_SHIFT = 27;
__device__ __forceinline__ int entry_tile(uint32_t e) { return int(e & TILE_IDX_MASK); }
__device__ __forceinline__ int entry_blk_k(uint32_t e) { const unsigned c = (e >> TILE_BLK_SHIFT) & 3u; return c ? 1 << c : 0; }

// A cell that receives its value from OUTSIDE the relaxation (an outlet seed of the upstream closure) is news for every tile that holds one of its
// 8 neighbours: the relaxation itself only reports cells that MOVE on a rim, so a seed on the first / last row or column of its tile must activate
// the tiles beside it as well (found in round 4 by oracle/taudem_oracle.c: orc_dinfdecayaccum_check - the catchment above an outlet in the top row
// of a tile was lost whenever no other rim cell of that tile moved).
__device__ __forceinline__ void activate_tiles_around(int x, int y, int nx, int ny, int tiles_x, uint32_t* __restrict__ tile_flags) {
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int xn = x + dx, yn = y + dy;
            if (xn >= 0 && xn < nx && yn >= 0 && yn < ny) tile_flags[(yn / TS) * tiles_x + xn / TS] = FLAG_FULL;
        }
}

constexpr int RES_CHANGED = 1 << 8;   // some cell of the tile moved (bits 0-7: which rim parts moved: N S W E NW NE SW SE)
constexpr int RES_CAPPED = 1 << 9;    // stopped at max_sweeps before the tile-local fixed point

// Relaxes tile `tile` to its local fixed point (or max_sweeps) and writes changed cells back.  Returns the
// RES_* mask, identical in every thread.  All threads of the workgroup must call it.
template <class Op>
__device__ __forceinline__ int relax_tile(const Op& op, const TileGeom& g, int tile, bool full, typename Op::T* sV, TileLds& L, unsigned long long* __restrict__ dbg) {
    using T = typename Op::T;
    const int tid = threadIdx.x;
    const int lx = tid & 63;
    const int wv = tid >> 6;
    const int ry0 = wv * RPW;
        const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
        const int x0 = tx * TS, y0 = ty * TS;
        const unsigned long long tc0 = dbg ? __builtin_readcyclecounter() : 0ull;
        if (tid == 0) L.rim = 0;
        // Stage the value tile + halo ring.  All global loads are issued before the first LDS store (ONE memory
        // latency per tile) and addresses advance by one row per step (the load phase is VALU-issue bound:
        // 34 loads per lane, so every instruction of address arithmetic counts).  Lane (lx, wv) loads column
        // x0 + lx of the rows wv, wv + 4, ... of the 66-row window; lanes 0..131 also load one cell of the two
        // halo columns.
        // The loads are UNCONDITIONAL (addresses clamped into the raster, validity applied afterwards): a load inside a
        // divergent branch gets its own basic block and the compiler waits for it before the next one is issued.
        constexpr int NROWL = (LH + NWAVE - 1) / NWAVE;   // 17 row steps
        const int gx = x0 + lx;
        const bool col_ok = gx < g.nx;
        const int gxc = col_ok ? gx : g.nx - 1;
        const long long row_pitch = (long long)g.nx;
        typename Op::Raw stage[NROWL], stage_side;
        unsigned stage_ok = 0;   // bit i: stage[i] is a raster cell
        bool side_ok = false;
        {
            int hy = y0 - 1 + wv;
#pragma unroll
            for (int i = 0; i < NROWL; i++) {
                const bool ok = col_ok && hy >= 0 && hy < g.ny && (wv + i * NWAVE < LH);
                const int hyc = hy < 0 ? 0 : (hy >= g.ny ? g.ny - 1 : hy);
                stage[i] = op.load_raw(size_t((long long)hyc * row_pitch + gxc));
                if (ok) stage_ok |= 1u << i;
                hy += NWAVE;
            }
            {
                const int row = (tid >> 1) < LH ? (tid >> 1) : LH - 1, right = tid & 1;
                const int sx = right ? x0 + TS : x0 - 1, sy = y0 - 1 + row;
                side_ok = tid < 2 * LH && sx >= 0 && sx < g.nx && sy >= 0 && sy < g.ny;
                const int sxc = sx < 0 ? 0 : (sx >= g.nx ? g.nx - 1 : sx), syc = sy < 0 ? 0 : (sy >= g.ny ? g.ny - 1 : sy);
                stage_side = op.load_raw(size_t((long long)syc * row_pitch + sxc));
            }
        }
        const unsigned long long tcA = dbg ? __builtin_readcyclecounter() : 0ull;
        typename Op::CellRaw craw[RPW];
        {
            int gy = y0 + ry0;
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const int gyc = gy >= g.ny ? g.ny - 1 : gy;
                craw[r] = op.cell_raw(size_t((long long)gyc * row_pitch + gxc));
                gy++;
            }
        }
        T cst[RPW];
        unsigned mk[RPW / 4] = {};
        unsigned live = 0;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int gy = y0 + ry0 + r;
            T c = Op::inf();
            unsigned m = 0;
            if (col_ok && gy >= g.y_own0 && gy < g.y_own1) Op::cell_decode(craw[r], c, m);
            cst[r] = c;
            mk[r >> 2] |= (m & 0xFFu) << (8 * (r & 3));
            if (m) live |= 1u << r;
        }
        const unsigned long long tcB = dbg ? __builtin_readcyclecounter() : 0ull;
#pragma unroll
        for (int i = 0; i < NROWL; i++)
            if (wv + i * NWAVE < LH) sV[(wv + i * NWAVE) * LP + lx + 1] = ((stage_ok >> i) & 1u) ? Op::decode(stage[i]) : Op::inf();
        if (tid < 2 * LH) sV[(tid >> 1) * LP + ((tid & 1) ? LH - 1 : 0)] = side_ok ? Op::decode(stage_side) : Op::inf();
        const unsigned long long tcC = dbg ? __builtin_readcyclecounter() : 0ull;
        __syncthreads();
        const unsigned long long tcD = dbg ? __builtin_readcyclecounter() : 0ull;
        if (dbg && tid == 0) { atomicAdd(dbg + 9, tcA - tc0); atomicAdd(dbg + 10, tcB - tcA); atomicAdd(dbg + 11, tcC - tcB); atomicAdd(dbg + 12, tcD - tcC); }
#pragma unroll
        for (int r = 0; r < RPW; r++)
            if (Op::settled(cst[r], sV[(ry0 + r + 1) * LP + lx + 1])) live &= ~(1u << r);
        unsigned moved = 0;   // rows of this lane that changed during this activation

        const unsigned long long tc1 = dbg ? __builtin_readcyclecounter() : 0ull;
        bool any_change = false, capped = false;
        // first look: everything, or (FLAG_HALO) only the cells that touch the halo ring
        unsigned perim = (lx == 0 || lx == TS - 1) ? ((1u << RPW) - 1u) : 0u;
        if (wv == 0) perim |= 1u;
        if (wv == NWAVE - 1) perim |= 1u << (RPW - 1);
        unsigned dirty = full ? live : (live & perim);
        for (int iter = 0;; iter++) {
            unsigned chg = 0;
            const int cur = iter & 1;
            // Within a sweep the rows of a segment are visited in lockstep by the 64 lanes, so a cell that
            // moves in one row can hand its value to the three cells below (above) it in the SAME sweep:
            // `prev` = columns that moved in the row just visited.
            // `rows` (wave-uniform, lives in an SGPR): rows in which at least one lane has something to look at; the other
            // rows of the unrolled loop cost two scalar instructions.  A row in which a cell moved pulls in the next one.
            unsigned long long prev = 0ull;
            if (lx == 0) L.rows[wv] = 0u;
            if (dirty) atomicOr(&L.rows[wv], dirty);   // LDS operations of one wave execute in order: no barrier needed
            unsigned rows = __builtin_amdgcn_readfirstlane(L.rows[wv]);
            if ((iter & 1) == 0) {   // downward
#pragma unroll
                for (int r = 0; r < RPW; r++) {
                    if (!((rows >> r) & 1u)) { prev = 0ull; continue; }
                    const unsigned long long dil = prev | (prev << 1) | (prev >> 1);
                    const bool look = ((((dirty >> r) | unsigned(dil >> lx)) & (live >> r)) & 1u) != 0u;
                    bool ch = false;
                    if (look) {
                        const int c = (ry0 + r + 1) * LP + lx + 1;
                        const T own = sV[c];
                        T m;
                        if (Op::kUniform == 4) m = uniform_min<T>(true, sV[c + 1], Op::inf(), sV[c - LP], Op::inf(), sV[c - 1], Op::inf(), sV[c + LP], Op::inf());
                        else if (Op::kUniform == 8) m = uniform_min<T>(false, sV[c + 1], sV[c - LP + 1], sV[c - LP], sV[c - LP - 1], sV[c - 1], sV[c + LP - 1], sV[c + LP], sV[c + LP + 1]);
                        else m = masked_min<T>(mask_at(mk, r), Op::inf(), sV[c + 1], sV[c - LP + 1], sV[c - LP], sV[c - LP - 1], sV[c - 1],
                                               sV[c + LP - 1], sV[c + LP], sV[c + LP + 1]);
                        const T wn = Op::apply(cst[r], own, m);
                        if (wn != own) {
                            sV[c] = wn;
                            ch = true;
                            chg |= 1u << r;
                            if (Op::settled(cst[r], wn)) live &= ~(1u << r);
                        }
                    }
                    prev = __ballot(ch);
                    if (prev) rows |= 2u << r;
                }
            } else {                 // upward
#pragma unroll
                for (int r = RPW - 1; r >= 0; r--) {
                    if (!((rows >> r) & 1u)) { prev = 0ull; continue; }
                    const unsigned long long dil = prev | (prev << 1) | (prev >> 1);
                    const bool look = ((((dirty >> r) | unsigned(dil >> lx)) & (live >> r)) & 1u) != 0u;
                    bool ch = false;
                    if (look) {
                        const int c = (ry0 + r + 1) * LP + lx + 1;
                        const T own = sV[c];
                        T m;
                        if (Op::kUniform == 4) m = uniform_min<T>(true, sV[c + 1], Op::inf(), sV[c - LP], Op::inf(), sV[c - 1], Op::inf(), sV[c + LP], Op::inf());
                        else if (Op::kUniform == 8) m = uniform_min<T>(false, sV[c + 1], sV[c - LP + 1], sV[c - LP], sV[c - LP - 1], sV[c - 1], sV[c + LP - 1], sV[c + LP], sV[c + LP + 1]);
                        else m = masked_min<T>(mask_at(mk, r), Op::inf(), sV[c + 1], sV[c - LP + 1], sV[c - LP], sV[c - LP - 1], sV[c - 1],
                                               sV[c + LP - 1], sV[c + LP], sV[c + LP + 1]);
                        const T wn = Op::apply(cst[r], own, m);
                        if (wn != own) {
                            sV[c] = wn;
                            ch = true;
                            chg |= 1u << r;
                            if (Op::settled(cst[r], wn)) live &= ~(1u << r);
                        }
                    }
                    prev = __ballot(ch);
                    if (prev) rows |= (1u << r) >> 1;
                }
            }
            // rows to look at next: 3x3 dilation of everything that moved in this sweep
            moved |= chg;
            const unsigned cl = __shfl_up(chg, 1, 64), cr = __shfl_down(chg, 1, 64);
            const unsigned h = chg | (lx > 0 ? cl : 0u) | (lx < 63 ? cr : 0u);
            unsigned nd = (h | (h << 1) | (h >> 1)) & ((1u << RPW) - 1u);
            const unsigned long long btop = __ballot((chg & 1u) != 0u), bbot = __ballot((chg & (1u << (RPW - 1))) != 0u);
            const unsigned long long bany = __ballot(chg != 0u);
            if (lx == 0) { L.top[cur][wv] = btop; L.bot[cur][wv] = bbot; L.any[cur][wv] = bany != 0ull; }
            __syncthreads();   // the only barrier of a sweep: everything a wave needs from the others is read after it
            unsigned any = 0;
#pragma unroll
            for (int w = 0; w < NWAVE; w++) any |= L.any[cur][w];
            if (any) any_change = true;
            if (!any || iter + 1 >= g.max_sweeps) {
                capped = any != 0;
                if (dbg && tid == 0) { atomicAdd(dbg, (unsigned long long)(iter + 1)); atomicAdd(dbg + 1, 1ull); }
                break;
            }
            const unsigned long long above = wv > 0 ? L.bot[cur][wv - 1] : 0ull, below = wv < NWAVE - 1 ? L.top[cur][wv + 1] : 0ull;
            if ((lx ? (above >> (lx - 1)) : (above << 1)) & 7ull) nd |= 1u;
            if ((lx ? (below >> (lx - 1)) : (below << 1)) & 7ull) nd |= 1u << (RPW - 1);
            dirty = nd & live;
        }
        const unsigned long long tc2 = dbg ? __builtin_readcyclecounter() : 0ull;
        if (any_change) {   // uniform
            int rim = 0;
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const int ly = ry0 + r;
                if ((moved >> r) & 1u) {
                    const T w = sV[(ly + 1) * LP + lx + 1];
                    op.store(size_t(y0 + ly) * size_t(g.nx) + size_t(gx), w);   // changed cells are in-grid and owned
                    const bool top = (ly == 0), bot = (ly == TS - 1), lef = (lx == 0), rig = (lx == TS - 1);
                    if (top) rim |= 1;
                    if (bot) rim |= 2;
                    if (lef) rim |= 4;
                    if (rig) rim |= 8;
                    if (top && lef) rim |= 16;
                    if (top && rig) rim |= 32;
                    if (bot && lef) rim |= 64;
                    if (bot && rig) rim |= 128;
                }
            }
            if (rim) atomicOr(&L.rim, rim);
        }
        __syncthreads();
        if (dbg && tid == 0) {   // phase cycles: load, sweeps, write-back
            const unsigned long long tc3 = __builtin_readcyclecounter();
            atomicAdd(dbg + 6, tc1 - tc0); atomicAdd(dbg + 7, tc2 - tc1); atomicAdd(dbg + 8, tc3 - tc2);
        }
        const int res = (any_change ? (L.rim | RES_CHANGED) : 0) | (capped ? RES_CAPPED : 0);
        __syncthreads();   // L.rim and the value tile are reused by the next tile
        return res;
}

// ---- register-resident variant ------------------------------------------------------------------------------------
// The LDS variant above spends its time waiting for LDS round trips (every row of a sweep is a read-compute-write chain
// through LDS).  Here a lane keeps its 16-row column segment in VGPRs; the values of the two neighbouring columns come
// from the neighbouring LANES through DPP wavefront shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1: one VALU op, no
// LDS), the tile's left / right halo column enters as the `old` operand of the shift in lane 0 / lane 63, and a value
// that moves is seen by the next row of the same sweep simply because it sits in a register.  LDS only carries what
// crosses waves: the first and last row of every 16-row band (double-buffered by sweep parity) and three flags per
// band and sweep.  Dirty tracking is per BAND and wave-uniform: a band in which nothing can have changed costs one scalar
// branch, a dirty band is swept whole as straight-line code (see the sweep loop of relax_tile_reg for why not per row).
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;
template <class T>
__device__ __forceinline__ T lane_left(T x, T edge) {    // value held by lane - 1; lane 0 gets `edge`
    static_assert(sizeof(T) == 4, "32-bit values");
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x), DPP_WAVE_SHR1, 0xf, 0xf, false));
}
template <class T>
__device__ __forceinline__ T lane_right(T x, T edge) {   // value held by lane + 1; lane 63 gets `edge`
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x), DPP_WAVE_SHL1, 0xf, 0xf, false));
}

constexpr int REG_LDS_WORDS = 2 * LH + 2 * NWAVE * 2 * TS + 2 * TS + 2 * LH;   // halo columns + double-buffered band boundary rows + activation thresholds of the halo rows and columns

// three-input minimum as ONE instruction (the C++ pattern a < b ? a : b compiles to compare + select per pair for floats)
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float min_raw(float a, float b) {   // (fminf / fmaxf add a canonicalising v_max x, x per operand)
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float max_raw(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int min3_raw(int a, int b, int c) {
    int d;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// bit k of m8 ? x : inf, as bit-field extract + bit-field insert (no compare, nothing for the compiler to hoist into SGPR pairs)
template <class T>
__device__ __forceinline__ T keep_if_bit(unsigned m8, int k, T x, T inf) {
    const int t = __builtin_amdgcn_sbfe(int(m8), k, 1);   // 0 or -1
    const int xi = __builtin_bit_cast(int, x), fi = __builtin_bit_cast(int, inf);
    return __builtin_bit_cast(T, (xi & t) | (fi & ~t));
}

// Six scratch registers for the lane shifts.  A wave_shr:1 never writes lane 0 and a wave_shl:1 never writes lane 63,
// so those lanes keep the +inf they are initialised with: "no neighbour lane" reads as +inf without a copy of the `old`
// operand in front of every DPP move.  What lanes 0 / 63 really have on that side - the tile's halo column - is constant
// during an activation and enters through a per-row precomputed minimum (halo_min).
template <class T>
struct ShiftRegs { T la, lc, lb, ra, rc, rb; };

// acc = min(acc, value of the neighbouring lanes) fused into ONE VOP2-DPP instruction per neighbour: a lane without a
// source lane (lane 0 of a wave_shr, lane 63 of a wave_shl) is simply not written, i.e. its missing neighbour counts as
// +inf.  (The s_nop covers the two wait states a DPP read needs after a VALU write of its source - hipcc does not see
// into the asm block.)
__device__ __forceinline__ float min_with_side_lanes(float acc, float x) {
    asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(x));
    return acc;
}
// Eight neighbours, separably: min over (a, b, hm) and over the side lanes' three-row minimum t = min3(a, c, b) - the side lanes hold the same
// rows as this one (a = the row above as this sweep has left it).  One block, so that the order is fixed: t, then the accumulator (the one
// wait state between t's write and its first DPP read that the s_nop 0 completes to two), then the two DPP-fused minima: 4 VALU instructions.
__device__ __forceinline__ float min8_separable(float a, float c, float b, float hm) {
    float acc, t;
    asm("v_min3_f32 %1, %2, %3, %4\n\tv_min3_f32 %0, %2, %4, %5\n\ts_nop 0\n\t"
        "v_min_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "=&v"(acc), "=&v"(t)
        : "v"(a), "v"(c), "v"(b), "v"(hm));
    return acc;
}
__device__ __forceinline__ int min8_separable(int a, int c, int b, int hm) {
    int acc, t;
    asm("v_min3_i32 %1, %2, %3, %4\n\tv_min3_i32 %0, %2, %4, %5\n\ts_nop 0\n\t"
        "v_min_i32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_i32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "=&v"(acc), "=&v"(t)
        : "v"(a), "v"(c), "v"(b), "v"(hm));
    return acc;
}
__device__ __forceinline__ float min_with_side_lanes3(float acc, float a, float c, float b) {
    asm("s_nop 1\n\t"
        "v_min_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_f32_dpp %0, %2, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_f32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(a), "v"(c), "v"(b));
    return acc;
}

// the same for 32-bit integers (v_min_i32 is a VOP2 instruction, so it takes the DPP modifier as well)
__device__ __forceinline__ int min_with_side_lanes(int acc, int x) {
    asm("s_nop 1\n\tv_min_i32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_i32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(x));
    return acc;
}
__device__ __forceinline__ int min_with_side_lanes3(int acc, int a, int c, int b) {
    asm("s_nop 1\n\t"
        "v_min_i32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_i32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_i32_dpp %0, %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_i32_dpp %0, %2, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_i32_dpp %0, %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_min_i32_dpp %0, %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(a), "v"(c), "v"(b));
    return acc;
}
// median of three as ONE instruction: with lock in {0, inf} and cand, own > 0 this is "own if locked, else min(cand, own)"
__device__ __forceinline__ float med3_raw(float a, float b, float c) {
    float d;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int med3_raw(int a, int b, int c) {
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Minimum over the 8 (4) neighbours of one row of cells.  a / c / b: this column in the row above / this row / the row
// below; hm: precomputed minimum over this row's neighbours in the halo column (lanes 0 and 63; +inf elsewhere).
template <class Op>
__device__ __forceinline__ typename Op::T reg_row_min(unsigned m8, typename Op::T a, typename Op::T c, typename Op::T b, typename Op::T hm,
                                                      ShiftRegs<typename Op::T>& s) {
    using T = typename Op::T;
    if constexpr (Op::kUniform != 0) {
#ifdef TDX_RELAX_NINEWAY
        const T acc = min3_raw(a, b, hm);
        return Op::kUniform == 8 ? min_with_side_lanes3(acc, a, c, b) : min_with_side_lanes(acc, c);   // 7 (3) instructions
#else
        if constexpr (Op::kUniform == 8) return min8_separable(a, c, b, hm);   // 4 instructions
        return min_with_side_lanes(min3_raw(a, b, hm), c);
#endif
    }
    s.lc = lane_left(c, s.lc);
    s.rc = lane_right(c, s.rc);
    if (Op::kUniform == 4) return min3_raw(min3_raw(a, b, s.lc), s.rc, hm);
    s.la = lane_left(a, s.la);
    s.ra = lane_right(a, s.ra);
    s.lb = lane_left(b, s.lb);
    s.rb = lane_right(b, s.rb);
    if (Op::kUniform == 8) return min3_raw(min3_raw(a, s.la, s.ra), min3_raw(s.lc, s.rc, b), min3_raw(s.lb, s.rb, hm));
    const T inf = Op::inf();   // bit k-1 of m8 selects neighbour k (1 E, 2 NE, 3 N, 4 NW, 5 W, 6 SW, 7 S, 8 SE)
    return min3_raw(min3_raw(keep_if_bit(m8, 0, s.rc, inf), keep_if_bit(m8, 1, s.ra, inf), keep_if_bit(m8, 2, a, inf)),
                    min3_raw(keep_if_bit(m8, 3, s.la, inf), keep_if_bit(m8, 4, s.lc, inf), keep_if_bit(m8, 5, s.lb, inf)),
                    min3_raw(keep_if_bit(m8, 6, b, inf), keep_if_bit(m8, 7, s.rb, inf), hm));
}
// The halo-column part of a row's neighbourhood: ha / hc / hb = halo-column cell beside the row above / this row / the
// row below (left column in lane 0, right column in lane 63).
template <class Op>
__device__ __forceinline__ typename Op::T halo_min(unsigned m8, bool left, bool right, typename Op::T ha, typename Op::T hc, typename Op::T hb) {
    using T = typename Op::T;
    const T inf = Op::inf();
    if (!left && !right) return inf;
    if (Op::kUniform == 4) return hc;
    if (Op::kUniform == 8) return min3_raw(ha, hc, hb);
    return left ? min3_raw(keep_if_bit(m8, 3, ha, inf), keep_if_bit(m8, 4, hc, inf), keep_if_bit(m8, 5, hb, inf))
                : min3_raw(keep_if_bit(m8, 1, ha, inf), keep_if_bit(m8, 0, hc, inf), keep_if_bit(m8, 7, hb, inf));
}

// x = 2 * x + (a != b): compare + add-with-carry, two VALU instructions per row for "which rows of this lane moved"
template <class T>
__device__ __forceinline__ unsigned shift_in_ne(unsigned x, T a, T b) {
    static_assert(sizeof(T) == 4, "32-bit values");
    if constexpr (std::is_same<T, float>::value)
        asm("v_cmp_neq_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
    else
        asm("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
    return x;
}

// bitwise OR over the 64 lanes of a wave (uniform result): four row_shr steps leave a row's OR in its lane 15, then four lane reads
__device__ __forceinline__ unsigned wave_or(unsigned x) {
    int y = int(x);
    y |= __builtin_amdgcn_update_dpp(0, y, 0x111, 0xf, 0xf, true);   // row_shr:1 (lanes without a source read 0)
    y |= __builtin_amdgcn_update_dpp(0, y, 0x112, 0xf, 0xf, true);   // row_shr:2
    y |= __builtin_amdgcn_update_dpp(0, y, 0x114, 0xf, 0xf, true);   // row_shr:4
    y |= __builtin_amdgcn_update_dpp(0, y, 0x118, 0xf, 0xf, true);   // row_shr:8
    return unsigned(__builtin_amdgcn_readlane(y, 15) | __builtin_amdgcn_readlane(y, 31) | __builtin_amdgcn_readlane(y, 47) | __builtin_amdgcn_readlane(y, 63));
}

// ---- activation filter.  A rim cell that moved matters to the tile beside it only if one of the (up to three) cells of that tile it touches can still be improved by
// it.  Op::act_threshold(cell_raw, value) (optional) gives, for a HALO cell as loaded, the bound below which a new neighbour value improves it - "never" (Op::act_never())
// for a cell that cannot move.  The halo as loaded is an upper bound of what the cell holds now (values only fall), so "new value < threshold as loaded" is necessary for an
// improvement: a flag that the test withholds could not have led to a change, now or later (the scripts/sim_*_rounds.py models: 17-19 % fewer activations, same fixed point).
template <class Op, class = void>
struct has_act_threshold : std::false_type {};
template <class Op>
struct has_act_threshold<Op, std::void_t<decltype(Op::act_never())>> : std::true_type {};
// Op::act_threshold_raw(raw) (optional): the same bound from the halo VALUE alone - then the halo columns can be filtered as well at no extra load (the level fields:
// a cell outside the queue holds -1, a seed a level no neighbour can undercut)
template <class Op, class = void>
struct has_act_threshold_raw : std::false_type {};
template <class Op>
struct has_act_threshold_raw<Op, std::void_t<decltype(Op::act_threshold_raw(typename Op::Raw()))>> : std::true_type {};
template <class T>
__device__ __forceinline__ T max_t(T a, T b) { return a > b ? a : b; }
// Op::raw_can_move(raw) (optional): false = the cell's VALUE says it never moves, whatever its mask byte holds (the level fields: a negative marker = outside the
// flat queue).  Lets a caller leave the mask bytes of such cells stale instead of clearing two whole rasters per flat iteration (flats.hpp).
template <class Op, class = void>
struct has_raw_can_move : std::false_type {};
template <class Op>
struct has_raw_can_move<Op, std::void_t<decltype(Op::raw_can_move(typename Op::Raw()))>> : std::true_type {};

// Same contract as relax_tile (sV must hold REG_LDS_WORDS words).
template <class Op>
__device__ __forceinline__ int relax_tile_reg(const Op& op, const TileGeom& g, int tile, typename Op::T* sV, TileLds& L, unsigned long long* __restrict__ dbg) {
    using T = typename Op::T;
    const int tid = threadIdx.x;
    const int lx = tid & 63;
    const int wv = tid >> 6;
    const int ry0 = wv * RPW;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    const int x0 = tx * TS, y0 = ty * TS;
    T* sSide = sV;            // [2][LH]: left / right halo column, window rows 0..65 (window row j = raster row y0 - 1 + j)
    T* sRow = sV + 2 * LH;    // [parity][band][top, bottom][TS]
    const unsigned long long tc0 = dbg ? __builtin_readcyclecounter() : 0ull;
    if (tid == 0) L.rim = 0;
    if (lx == 0) L.rows[wv] = 0u;
    // ---- load: 16 + 16 + 2 global loads per lane, issued back to back (addresses clamped, validity applied afterwards)
    const int gx = x0 + lx;
    const bool col_ok = gx < g.nx;
    const int gxc = col_ok ? gx : g.nx - 1;
    const long long row_pitch = (long long)g.nx;
    typename Op::Raw raw[RPW], raw_edge, raw_side;
    typename Op::CellRaw craw[RPW], craw_edge = {};
    {
        int gy = y0 + ry0;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int gyc = gy >= g.ny ? g.ny - 1 : gy;
            const size_t idx = size_t((long long)gyc * row_pitch + gxc);
            raw[r] = op.load_raw(idx);
            craw[r] = op.cell_raw(idx);
            gy++;
        }
    }
    const int ey = (wv == 0) ? y0 - 1 : y0 + TS;   // tile halo row above (first band) / below (last band); unused by the inner bands
    const bool edge_ok = col_ok && ey >= 0 && ey < g.ny;
    bool side_ok;
    {
        const int eyc = ey < 0 ? 0 : (ey >= g.ny ? g.ny - 1 : ey);
        raw_edge = op.load_raw(size_t((long long)eyc * row_pitch + gxc));
        if constexpr (has_act_threshold<Op>::value && !has_act_threshold_raw<Op>::value) craw_edge = op.cell_raw(size_t((long long)eyc * row_pitch + gxc));
        const int row = (tid >> 1) < LH ? (tid >> 1) : LH - 1, right = tid & 1;
        const int sx = right ? x0 + TS : x0 - 1, sy = y0 - 1 + row;
        side_ok = tid < 2 * LH && sx >= 0 && sx < g.nx && sy >= 0 && sy < g.ny;
        const int sxc = sx < 0 ? 0 : (sx >= g.nx ? g.nx - 1 : sx), syc = sy < 0 ? 0 : (sy >= g.ny ? g.ny - 1 : sy);
        raw_side = op.load_raw(size_t((long long)syc * row_pitch + sxc));   // (132 cache lines for 132 cells - measured not to be what bounds a round: docs/experiments_r06.md section 2)
    }
    T v[RPW], cst[RPW];
    unsigned mk[RPW / 4] = {};
    unsigned live = 0;
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const int gy = y0 + ry0 + r;
        v[r] = (col_ok && gy < g.ny) ? Op::decode(raw[r]) : Op::inf();
        T c = Op::inf();
        unsigned m = 0;
        bool movable = col_ok && gy >= g.y_own0 && gy < g.y_own1;
        if constexpr (has_raw_can_move<Op>::value) movable = movable && Op::raw_can_move(raw[r]);
        if (movable) Op::cell_decode(craw[r], c, m);
        if (m && !Op::settled(c, v[r])) live |= 1u << r;
        cst[r] = Op::cell_floor(c, v[r]);
        mk[r >> 2] |= (m & 0xFFu) << (8 * (r & 3));
    }
    const T edge_row = edge_ok ? Op::decode(raw_edge) : Op::inf();
    if (tid < 2 * LH) sSide[(tid & 1) * LH + (tid >> 1)] = side_ok ? Op::decode(raw_side) : Op::inf();
    // activation thresholds of the halo (see has_act_threshold): the row above / below per lane as the maximum over the three cells a rim cell touches, the side columns in LDS
    T* sEdgeT = sRow + 2 * NWAVE * 2 * TS;   // [above, below][TS] (kept in LDS, not in a register across the sweeps: the kernels sit at their register limit)
    T* sSideT = sEdgeT + 2 * TS;             // [2][LH] (operators with act_threshold_raw only)
    if constexpr (has_act_threshold<Op>::value) {
        T te;
        if constexpr (has_act_threshold_raw<Op>::value) te = edge_ok ? Op::act_threshold_raw(raw_edge) : Op::act_never();
        else te = edge_ok ? Op::act_threshold(craw_edge, Op::decode(raw_edge)) : Op::act_never();
        const T te3 = max_t(te, max_t(lane_left(te, Op::act_never()), lane_right(te, Op::act_never())));
        if (wv == 0) sEdgeT[lx] = te3;
        if (wv == NWAVE - 1) sEdgeT[TS + lx] = te3;
        if constexpr (has_act_threshold_raw<Op>::value)
            if (tid < 2 * LH) sSideT[(tid & 1) * LH + (tid >> 1)] = side_ok ? Op::act_threshold_raw(raw_side) : Op::act_never();
    }
    sRow[((0 * NWAVE + wv) * 2 + 0) * TS + lx] = v[0];
    sRow[((0 * NWAVE + wv) * 2 + 1) * TS + lx] = v[RPW - 1];
    if (live) atomicOr(&L.rows[wv], live);
    __syncthreads();
    // per row: minimum over the neighbours in the tile's halo column (window rows ry0 + r .. ry0 + r + 2), lanes 0 and 63 only
    T hm[RPW];
    {
        const bool left = lx == 0, right = lx == TS - 1;
        T sd[RPW + 2];
#pragma unroll
        for (int j = 0; j < RPW + 2; j++) sd[j] = sSide[(right ? LH : 0) + ry0 + j];
#pragma unroll
        for (int r = 0; r < RPW; r++) hm[r] = halo_min<Op>(mask_at(mk, r), left, right, sd[r], sd[r + 1], sd[r + 2]);
    }
    ShiftRegs<T> sh = {Op::inf(), Op::inf(), Op::inf(), Op::inf(), Op::inf(), Op::inf()};
    const unsigned rows_live = __builtin_amdgcn_readfirstlane(L.rows[wv]);   // rows of this band with a cell that can still move
    unsigned moved = 0;     // rows of this lane that changed during this activation
    const unsigned long long tc1 = dbg ? __builtin_readcyclecounter() : 0ull;
    bool any_change = false, capped = false;
    unsigned rows = rows_live;
    for (int iter = 0;; iter++) {
        const int cur = iter & 1;
        const T up = (wv == 0) ? edge_row : sRow[((cur * NWAVE + wv - 1) * 2 + 1) * TS + lx];
        const T dn = (wv == NWAVE - 1) ? edge_row : sRow[((cur * NWAVE + wv + 1) * 2 + 0) * TS + lx];
        // A band with a dirty row is swept whole, as straight-line code: no per-row "did anybody move" test.  (The per-row form - skip a
        // clean row, pull in the next row when this one moved - put a VALU compare -> VCC -> scalar compare -> branch chain between any two
        // rows: ~185 cycles per row for 12 VALU instructions; sweeping from the first dirty group of four rows was no faster than this.)
        // What moved is collected per lane - compare + add-with-carry shift the row's bit in - and reduced to the wave-uniform row mask
        // once per sweep.  10 VALU instructions per row for the uniform operators.
        unsigned chg_lane = 0;   // rows of this lane that moved in this sweep
        if (rows == 0u) {
        } else if ((iter & 1) == 0) {   // downward: a row that moved hands its value to the next one in the same sweep
#pragma unroll
            for (int r = 0; r < RPW; r++) {
                const T a = r ? v[r ? r - 1 : 0] : up, c = v[r], b = (r < RPW - 1) ? v[r < RPW - 1 ? r + 1 : r] : dn;
                unsigned m8 = mask_at(mk, r);
                if (Op::kUniform == 0) asm volatile("" : "+v"(m8));   // keeps the per-row mask arithmetic inside the sweep loop (128 hoisted conditions spill)
                const T m = reg_row_min<Op>(m8, a, c, b, hm[r], sh);
                const T wn = Op::apply(cst[r], c, m);
                chg_lane = shift_in_ne(chg_lane, wn, c);
                v[r] = wn;
            }
            chg_lane = __builtin_bitreverse32(chg_lane) >> (32 - RPW);   // row 0 was shifted in first
        } else {                        // upward (row RPW - 1 is shifted in first: bit r = row r)
#pragma unroll
            for (int r = RPW - 1; r >= 0; r--) {
                const T a = r ? v[r ? r - 1 : 0] : up, c = v[r], b = (r < RPW - 1) ? v[r < RPW - 1 ? r + 1 : r] : dn;
                unsigned m8 = mask_at(mk, r);
                if (Op::kUniform == 0) asm volatile("" : "+v"(m8));
                const T m = reg_row_min<Op>(m8, a, c, b, hm[r], sh);
                const T wn = Op::apply(cst[r], c, m);
                chg_lane = shift_in_ne(chg_lane, wn, c);
                v[r] = wn;
            }
        }
        moved |= chg_lane;
        const unsigned chg_rows = wave_or(chg_lane);   // wave-uniform: rows in which some cell moved in this sweep
        // publish this band's boundary rows for the next sweep (other parity: a slow wave may still be reading this one)
        sRow[(((cur ^ 1) * NWAVE + wv) * 2 + 0) * TS + lx] = v[0];
        sRow[(((cur ^ 1) * NWAVE + wv) * 2 + 1) * TS + lx] = v[RPW - 1];
        if (lx == 0) {
            L.any[cur][wv] = chg_rows != 0u;
            L.top[cur][wv] = chg_rows & 1u;
            L.bot[cur][wv] = (chg_rows >> (RPW - 1)) & 1u;
        }
        __syncthreads();   // the only barrier of a sweep
        unsigned any = 0;
#pragma unroll
        for (int w = 0; w < NWAVE; w++) any |= L.any[cur][w];
        if (any) any_change = true;
        if (!any || iter + 1 >= g.max_sweeps) {
            capped = any != 0;
            if (dbg && tid == 0) { atomicAdd(dbg, (unsigned long long)(iter + 1)); atomicAdd(dbg + 1, 1ull); }
            break;
        }
        unsigned nd = (chg_rows | (chg_rows << 1) | (chg_rows >> 1)) & ((1u << RPW) - 1u);
        const unsigned above = wv > 0 ? unsigned(L.bot[cur][wv > 0 ? wv - 1 : 0]) : 0u, below = wv < NWAVE - 1 ? unsigned(L.top[cur][wv < NWAVE - 1 ? wv + 1 : 0]) : 0u;
        if (above) nd |= 1u;
        if (below) nd |= 1u << (RPW - 1);
        rows = __builtin_amdgcn_readfirstlane(nd) & rows_live;
    }
    const unsigned long long tc2 = dbg ? __builtin_readcyclecounter() : 0ull;
    if (any_change) {   // uniform
        int rim = 0;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int ly = ry0 + r;
            if ((moved >> r) & 1u) {
                op.store(size_t(y0 + ly) * size_t(g.nx) + size_t(gx), v[r]);   // changed cells are in-grid and owned
                const bool top = (ly == 0), bot = (ly == TS - 1), lef = (lx == 0), rig = (lx == TS - 1);
                if constexpr (has_act_threshold<Op>::value) {
                    // a neighbour is told only if a cell of it that this cell touches can still be improved by the new value
                    const bool all = g.act_filter == 0;
                    if ((top || bot) && (all || v[r] < sEdgeT[(bot ? TS : 0) + lx])) rim |= top ? 1 : 2;
                    if constexpr (has_act_threshold_raw<Op>::value) {
                        if (lef || rig) {
                            const T* st = sSideT + (rig ? LH : 0);   // window rows ly .. ly + 2 = the cells beside rows ly - 1 .. ly + 1
                            if (all || v[r] < max_t(st[ly], max_t(st[ly + 1], st[ly + 2]))) rim |= lef ? 4 : 8;
                            if (top && (all || v[r] < st[0])) rim |= lef ? 16 : 32;
                            if (bot && (all || v[r] < st[LH - 1])) rim |= lef ? 64 : 128;
                        }
                    } else {
                        // (an operator whose threshold needs the cell's constant: the tiles beside and at the corners are told whenever their rim cell moves -
                        // their thresholds would be one more strided column load per activation, which costs what the withheld activations save: measured)
                        if (lef) rim |= 4;
                        if (rig) rim |= 8;
                        if (top && lef) rim |= 16;
                        if (top && rig) rim |= 32;
                        if (bot && lef) rim |= 64;
                        if (bot && rig) rim |= 128;
                    }
                } else {
                    if (top) rim |= 1;
                    if (bot) rim |= 2;
                    if (lef) rim |= 4;
                    if (rig) rim |= 8;
                    if (top && lef) rim |= 16;
                    if (top && rig) rim |= 32;
                    if (bot && lef) rim |= 64;
                    if (bot && rig) rim |= 128;
                }
            }
        }
        if (rim) atomicOr(&L.rim, rim);
    }
    __syncthreads();
    if (dbg && tid == 0) {   // phase cycles: load, sweeps, write-back
        const unsigned long long tc3 = __builtin_readcyclecounter();
        atomicAdd(dbg + 6, tc1 - tc0); atomicAdd(dbg + 7, tc2 - tc1); atomicAdd(dbg + 8, tc3 - tc2);
    }
    const int res = (any_change ? (L.rim | RES_CHANGED) : 0) | (capped ? RES_CAPPED : 0);
    __syncthreads();   // L.rim and the LDS rows are reused by the next tile
    return res;
}

// Activates tiles for the NEXT round.  The first workgroup that raises a tile's flag from 0 owns its list entry; the entries
// are collected in LDS and appended with ONE atomic on the next round's counter per pull of the cursor: a single hot
// address sustains only ~90 M atomics/s on MI355X, which is 0.7 ms for a round over all 65536 tiles of a 16384^2 raster.
// Only lanes of wave 0 push; relax_kernel flushes with the whole workgroup between two barriers.
// the tile that direction bit `tid` of `res` (bits 0-7: N S W E NW NE SW SE rim parts; lane 8: the tile itself when capped) activates, or -1
__device__ __forceinline__ int activation_target(int res, int tile, const TileGeom& g, uint32_t* flag) {
    const int tid = threadIdx.x;
    int target = -1;
    *flag = FLAG_HALO;
    if (tid < 8 && ((res >> tid) & 1)) {
        const int ddx[8] = {0, 0, -1, 1, -1, 1, -1, 1};
        const int ddy[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
        const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
        const int ntx = tx + ddx[tid], nty = ty + ddy[tid];
        if (ntx >= 0 && ntx < g.tiles_x && nty >= 0 && nty < g.tiles_y) target = nty * g.tiles_x + ntx;
        if (g.remap != nullptr && target >= 0) target = int(g.remap[target]);   // a tile of a macro block: the block's first tile stands for it (+ the block bits)
    }
    if (tid == 8 && (res & RES_CAPPED)) { target = tile; *flag = FLAG_FULL; }   // not yet at its fixed point: run again, everything dirty
    return target;
}
__device__ __forceinline__ void flag_neighbours(int res, int tile, const TileGeom& g, uint32_t* __restrict__ flags_next, TileLds& L) {
    uint32_t flag;
    const int target = activation_target(res, tile, g, &flag);
    if (target >= 0 && atomicMax(&flags_next[entry_tile(uint32_t(target))], flag) == 0u) L.pend[atomicAdd(&L.npend, 1u)] = uint32_t(target);
}
// ---- schedule 1: rounds, ONE launch per round.  Workgroups pull the tiles of the current round's list from a device
// cursor (1 ... PULL_MAX list entries per atomic, depending on the size of the round), consume (read + clear) their
// activation flags in this round's flag half and activate tiles for the next round in the other half / the other list
// (count[1] = the next round's size).  Flags raised in round r are only read in round r + 1, so everything a tile loads
// was written before its launch started.
// The work distribution of one round, shared by every tile kernel that uses this schedule (the relaxation below, the
// dependency sweeps of tile_dep.hpp): body(tile, full) processes one tile with the whole workgroup and returns its RES_* mask
// (uniform); all threads must call it, and it must end with a barrier.
template <class Body>
__device__ __forceinline__ void round_driver(const uint32_t* __restrict__ list, unsigned long long* __restrict__ count, uint32_t* __restrict__ flags_cur,
                                             uint32_t* __restrict__ flags_next, uint32_t* __restrict__ list_next, unsigned pull_max, const TileGeom& g,
                                             TileLds& L, Body body, unsigned bid, unsigned nblocks) {
    // bid / nblocks: this workgroup's index among the workgroups of THIS schedule (a launch can carry two: relax_pair_kernel)
    const int chain_max = g.chain_max;
    const unsigned nact = unsigned(count[0]);
    if (g.cnt_host != nullptr && bid == 0u && threadIdx.x == 0)   // this round's size, for the host (final: the previous round's launch has ended)
        __hip_atomic_store(g.cnt_host + (count - g.cnt_dev), g.cnt_tag | (unsigned long long)count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t entry0 = list[bid];                 // for a small round (below); fetched together with the count: nblocks <= number of tiles
    unsigned long long* cursor = count + COUNT_RING;   // per-round work cursor: blocks pull tiles, so the load balances itself
    unsigned pull = nact / (2u * nblocks);
    pull = pull < 1u ? 1u : (pull > pull_max ? pull_max : pull);
    // A small round (the long tail of a relaxation is one dependency front crossing one tile per round) is pure latency:
    // with no more tiles than workgroups every workgroup takes the list entry of its index and the cursor is not used.
    const bool fixed = nact <= nblocks;
    // at most ceil(nact / pull) pulls find work: the other workgroups of the grid (sized before the round's length is
    // known) leave without touching the cursor - 2048 atomics on one address are ~20 us, as much as a whole small round
    if (bid * pull >= nact) return;
    if (threadIdx.x == 0) L.npend = 0u;
    for (unsigned turn = 0;; turn++) {
        __syncthreads();   // the activations of the previous pull are all in L.pend
        const unsigned npend = L.npend;
        if (threadIdx.x == 0) {
            L.base = npend ? atomicAdd(count + 1, (unsigned long long)npend) : 0ull;
            L.next = fixed ? (turn == 0u ? bid : nact) : unsigned(atomicAdd(cursor, (unsigned long long)pull));
        }
        __syncthreads();
        const unsigned first = L.next;
        const unsigned long long base = L.base;
        for (unsigned i = threadIdx.x; i < npend; i += blockDim.x) list_next[base + i] = L.pend[i];
        if (threadIdx.x == 0) L.npend = 0u;   // the next push comes after the first barrier of the next tile
        if (first >= nact) break;
        const unsigned last = first + pull < nact ? first + pull : nact;
        if (!fixed) {
            if (threadIdx.x < last - first) L.pulled[threadIdx.x] = list[first + threadIdx.x];
            __syncthreads();
        }
        for (unsigned it = first; it < last; it++) {
            const uint32_t ent = fixed ? entry0 : L.pulled[it - first];
            int tile = entry_tile(ent);
            bool full = flags_cur[tile] >= FLAG_FULL;
            int res;
            if (entry_blk_k(ent)) {   // a macro block: served by the block workgroups of this launch (macro_role); here only its flag is taken down
                __syncthreads();      // (every lane has read the flag)
                if (threadIdx.x == 0) flags_cur[tile] = 0u;
                continue;
            }
            // REQUIREMENT of the hand-over below: while a schedule is in a solo round NOTHING else reads or writes the field it relaxes - true for every
            // caller (one field per schedule; the pair / fused runs of flats.hpp drive two DIFFERENT fields, the dependency sweeps own their work array).
            // A caller that shares a field between two concurrent schedules must switch it off (TileGeom::chain_max = 0).
            // A SOLO round (one active tile in the whole raster: the tail of every dependency sweep is one chain of such rounds, the
            // longest flow path crossing one tile per round) hands over inside the launch: while an activation activates exactly one
            // tile, this workgroup - the only one running - goes on into it, instead of a list append, the end of the kernel, the next
            // launch and its count -> list -> flag -> data chain of dependent loads.  Nobody else reads or writes during a solo round, so
            // the only ordering needed is this workgroup's own stores before its next loads.  (ONE call site of body(): a second inlined
            // copy of a tile kernel changes its register allocation - the D-infinity decay sweep fell from 4 to 3 waves per SIMD.)
            for (int hop = 0;; hop++) {
                res = body(tile, full);   // (ends with a barrier: every lane has read the flag)
                if (hop == 0 && threadIdx.x == 0) flags_cur[tile] = 0u;
                if (nact != 1u || hop >= chain_max || !(res & (RES_CHANGED | RES_CAPPED))) break;
                uint32_t flag;
                const int target = activation_target(res, tile, g, &flag);
                if (threadIdx.x < 64) {
                    const unsigned long long bal = __ballot(target >= 0);
                    if (target >= 0 && (bal & (bal - 1ull)) == 0ull) L.chain = target | (flag >= FLAG_FULL ? int(0x80000000u) : 0);
                    if (threadIdx.x == 0 && (bal == 0ull || (bal & (bal - 1ull)) != 0ull)) L.chain = -1;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // this lane's write-back has landed, its cached lines are dropped
                __syncthreads();
                const int ch = L.chain;
                if (ch == -1) break;
                if (entry_blk_k(uint32_t(ch))) break;   // a macro block is never entered by hand-over: it is flagged for its own workgroups (below)
                tile = entry_tile(uint32_t(ch));   // (the next body()'s first barrier comes after every lane has read L.chain)
                full = ch < 0;
            }
            if (res & (RES_CHANGED | RES_CAPPED)) flag_neighbours(res, tile, g, flags_next, L);
        }
    }
}

// waves per SIMD (= workgroups per CU) a relaxation kernel is compiled for: Op::kWaves, 4 (128 VGPRs) unless the operator says otherwise
template <class Op, class = void>
struct relax_waves : std::integral_constant<int, 4> {};
template <class Op>
struct relax_waves<Op, std::void_t<decltype(Op::kWaves)>> : std::integral_constant<int, Op::kWaves> {};
template <class Body>
__device__ __forceinline__ void round_driver(const uint32_t* __restrict__ list, unsigned long long* __restrict__ count, uint32_t* __restrict__ flags_cur,
                                             uint32_t* __restrict__ flags_next, uint32_t* __restrict__ list_next, unsigned pull_max, const TileGeom& g,
                                             TileLds& L, Body body) {
    round_driver(list, count, flags_cur, flags_next, list_next, pull_max, g, L, body, blockIdx.x, gridDim.x);
}

template <class Op, class = void>
struct has_plain : std::false_type {};
template <class Op>
struct has_plain<Op, std::enable_if_t<Op::kHasPlain>> : std::true_type {};

// Op::kMacroLdsWords + op.macro_update(g, tile, k, lds, L, flags_next) (optional): the operator solves a K x K block of tiles in closed form (TileGeom::blk_k)
template <class Op, class = void>
struct has_macro : std::false_type {};
template <class Op>
struct has_macro<Op, std::enable_if_t<(Op::kMacroLdsWords > 0)>> : std::true_type {};
template <class Op, bool = has_macro<Op>::value>
struct macro_lds_words : std::integral_constant<int, 0> {};
template <class Op>
struct macro_lds_words<Op, true> : std::integral_constant<int, Op::kMacroLdsWords> {};

// The macro blocks of a round (TileGeom::blk_k; flats.hpp: OPEN WATER) have workgroups of their own in the round's launch - the ones from index grid_tiles on -
// instead of a branch inside the tile loop: inlined there, the closed-form update cost the tile kernel two more spilled registers and 35 scalar spills
// (+1.3 ms per 16384^2 step with NO block in sight), as a called function a stack (+2.4 ms).  These workgroups look through the round's list, 256 entries at
// a time, for first tiles of blocks, update them (Op::macro_update raises the neighbours' flags into TL.pend) and append what they activated to the next
// round's list like round_driver does; the tile workgroups pass such entries by.
template <class Op>
__device__ __forceinline__ void macro_role(const Op& op, const TileGeom& g, const uint32_t* __restrict__ list, unsigned long long* __restrict__ count,
                                           uint32_t* __restrict__ flags_next, uint32_t* __restrict__ list_next, int* __restrict__ lds, TileLds& L, unsigned bid, unsigned nblocks,
                                           unsigned long long* __restrict__ dbg) {
    const unsigned nact = unsigned(count[0]);
    if (threadIdx.x == 0) L.npend = 0u;
    auto serve = [&](unsigned nb) {   // the blocks in L.pulled[0 .. nb): update, then append what they activated to the next round's list
        for (unsigned b = 0; b < nb; b++) {
            const int tile = entry_tile(L.pulled[b]), kb = entry_blk_k(L.pulled[b]);
            const unsigned long long tc0 = dbg ? __builtin_readcyclecounter() : 0ull;
            (void)op.macro_update(g, tile, kb, lds, L, flags_next, dbg);   // (ends with a barrier)
            if (dbg && threadIdx.x == 0) { atomicAdd(dbg + 13, __builtin_readcyclecounter() - tc0); atomicAdd(dbg + 14, 1ull); }   // cycles in block updates, updates
            const unsigned npend = L.npend;
            if (threadIdx.x == 0) L.base = npend ? atomicAdd(count + 1, (unsigned long long)npend) : 0ull;
            __syncthreads();
            const unsigned long long lb = L.base;
            for (unsigned q = threadIdx.x; q < npend; q += blockDim.x) list_next[lb + q] = L.pend[q];
            __syncthreads();
            if (threadIdx.x == 0) L.npend = 0u;
        }
    };
    // A round of up to 4096 entries - every round but the first few: EVERY block workgroup numbers the round's blocks in list order (16 entries per thread, one
    // workgroup-wide prefix count) and serves the blocks whose number is its own modulo the number of block workgroups: no workgroup serves two blocks while
    // another idles (with list entries dealt out by index, 150 blocks on 1 024 workgroups met two to four at a time in one of them: rounds of 35 - 76 us).
    if (nact <= 4096u) {
        constexpr int PER = 16;
        uint32_t ent[PER];
        unsigned kk[PER];
        const unsigned first = threadIdx.x * PER;
        (void)first;
#pragma unroll
        for (int j = 0; j < PER; j++) { const unsigned i = unsigned(j) * NTHR + threadIdx.x; ent[j] = i < nact ? list[i] : 0u; }   // (coalesced; any numbering all workgroups agree on will do)
        unsigned mine = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { kk[j] = unsigned(entry_blk_k(ent[j])); mine += kk[j] != 0u; }
        // exclusive prefix of `mine` over the workgroup's 256 threads
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off, 64); if ((threadIdx.x & 63) >= unsigned(off)) incl += t; }
        __syncthreads();
        if ((threadIdx.x & 63) == 63) L.pend[threadIdx.x >> 6] = incl;   // (the pend array is idle here)
        if (threadIdx.x == 0) L.next = 0u;
        __syncthreads();
        unsigned before = incl - mine;
        for (unsigned w = 0; w < (threadIdx.x >> 6); w++) before += L.pend[w];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++)
            if (kk[j] != 0u) {
                if (before % nblocks == bid) { const unsigned slot = atomicAdd(&L.next, 1u); if (slot < unsigned(PULL_MAX)) L.pulled[slot] = ent[j]; }
                before++;
            }
        __syncthreads();
        if (dbg && threadIdx.x == 0 && L.next) atomicMax(dbg + 15, (unsigned long long)L.next);
        serve(L.next < unsigned(PULL_MAX) ? L.next : unsigned(PULL_MAX));   // (at most 4096 / 4 blocks over >= 256 workgroups: never more than 4 each)
        return;
    }
    // 64 list entries per look (at most PULL_MAX = 64 blocks to remember), STRIDED over the list: neighbouring entries - blocks activated by the same front - go
    // to different workgroups (a workgroup that took 64 consecutive entries served a dozen blocks one after the other while the others idled: 500 us per round)
    for (unsigned base = bid; base < nact; base += nblocks * 64u) {
        __syncthreads();                                   // (L.pulled of the previous batch has been read)
        if (threadIdx.x == 0) L.next = 0u;
        __syncthreads();
        if (threadIdx.x < 64u) {
            const unsigned i = base + nblocks * threadIdx.x;
            const uint32_t e = i < nact ? list[i] : 0u;
            if (i < nact && entry_blk_k(e)) L.pulled[atomicAdd(&L.next, 1u)] = e;
        }
        __syncthreads();
        const unsigned nb = L.next;
        if (dbg && threadIdx.x == 0 && nb) atomicMax(dbg + 15, (unsigned long long)nb);   // TDX_DEBUG_ROUNDS=1: most blocks one workgroup served in one look
        serve(nb);
    }
}

// grid_tiles: 0 = every workgroup serves tiles; otherwise the workgroups from index grid_tiles on serve the round's macro blocks (macro_role)
template <class Op, bool REG>
__global__ __launch_bounds__(NTHR, relax_waves<Op>::value) void relax_kernel(Op op, TileGeom g, const uint32_t* __restrict__ list, unsigned long long* __restrict__ count,
                                                    uint32_t* __restrict__ flags_cur, uint32_t* __restrict__ flags_next,
                                                    uint32_t* __restrict__ list_next, unsigned pull_max, unsigned long long* __restrict__ dbg, unsigned grid_tiles) {
    using T = typename Op::T;
    static_assert(sizeof(T) == 4, "tile engine works on 4-byte values");
    constexpr int kLdsWords = REG ? (REG_LDS_WORDS > macro_lds_words<Op>::value ? REG_LDS_WORDS : macro_lds_words<Op>::value) : LH * LP;
    __shared__ T sV[kLdsWords];
    __shared__ TileLds L;
    if constexpr (REG && has_macro<Op>::value) {
        if (grid_tiles != 0u && blockIdx.x >= grid_tiles) {
            macro_role(op, g, list, count, flags_next, list_next, reinterpret_cast<int*>(sV), L, blockIdx.x - grid_tiles, gridDim.x - grid_tiles, dbg);
            return;
        }
    }
    const unsigned nblocks = grid_tiles != 0u ? grid_tiles : gridDim.x;
    round_driver(list, count, flags_cur, flags_next, list_next, pull_max, g, L, [&](int tile, bool full) {
        if constexpr (REG && has_plain<Op>::value) {   // tiles whose masks only say "may move" take the uniform form of the operator (flats.hpp)
            if (!__builtin_amdgcn_readfirstlane(int(op.tile_masked(tile)))) return relax_tile_reg(op.plain(), g, tile, sV, L, dbg);
        }
        return REG ? relax_tile_reg(op, g, tile, sV, L, dbg) : relax_tile(op, g, tile, full, sV, L, dbg);
    }, blockIdx.x, nblocks);
}

// Two independent relaxations of the same operator type (the two level fields of flat resolution) in ONE launch per round: the first
// gridA workgroups serve schedule a, the others schedule b.  The rounds of the two advance in lockstep (a schedule that has ended sees
// empty rounds); nothing depends on two HIP streams being executed side by side.  Opt-in (TDX_FLATS_FUSED=1, flats.hpp): 0.5 ms slower at
// 16384^2 than two streams that do overlap (profiles/r03y_*), because a fused round lasts as long as the slower field's.
struct RoundArgs { const uint32_t* list; unsigned long long* count; uint32_t* flags_cur; uint32_t* flags_next; uint32_t* list_next; };
template <class Op>
__global__ __launch_bounds__(NTHR, relax_waves<Op>::value) void relax_pair_kernel(Op opA, Op opB, TileGeom g, RoundArgs a, RoundArgs b, unsigned gridA,
                                                                                  unsigned pull_max) {
    using T = typename Op::T;
    static_assert(sizeof(T) == 4, "tile engine works on 4-byte values");
    __shared__ T sV[REG_LDS_WORDS];
    __shared__ TileLds L;
    const bool second = blockIdx.x >= gridA;   // uniform
    const Op op = second ? opB : opA;
    const RoundArgs ra = second ? b : a;
    round_driver(ra.list, ra.count, ra.flags_cur, ra.flags_next, ra.list_next, pull_max, g, L,
                 [&](int tile, bool) {
                     if constexpr (has_plain<Op>::value) {
                         if (!__builtin_amdgcn_readfirstlane(int(op.tile_masked(tile)))) return relax_tile_reg(op.plain(), g, tile, sV, L, nullptr);
                     }
                     return relax_tile_reg(op, g, tile, sV, L, nullptr);
                 },
                 second ? blockIdx.x - gridA : blockIdx.x, second ? gridDim.x - gridA : gridA);
}

// ---- schedule 2: asynchronous worklist.  ONE launch: resident workgroups pop tiles from a device queue, relax
// them, and push the neighbours whose halo they changed - a dependency front advances at the pace of single
// tiles instead of one tile per launch.  Hand-off between workgroups follows the agent-scope release / acquire
// recipe of cdna_hip_programming.md (Guideline 16): every wave drains its stores, barrier, lane 0 release-fences
// and only then publishes through atomics; the consumer acquires (L1 invalidate) after popping, before loading.
// Tile state machine (atomic CAS, so that every state change is an RMW on one word):
//   0 idle -> 1 queued -> 2 running -> 0;   a neighbour that changes a running tile's halo turns 2 into 3
//   (running, must run again) and the worker re-queues the tile (3 -> 1) when it finishes.
struct AsyncCtl {
    uint32_t* ring; uint32_t mask;          // ticket ring of tile+1 (0 = empty slot); size mask+1 > number of tiles
    unsigned long long* head; unsigned long long* tail;
    int* pending;                            // tiles queued or running
    uint32_t* state;
    unsigned* error;                         // a bounded spin gave up: the host falls back to schedule 1
    unsigned spin_limit;                     // polls before a spin gives up
    unsigned long long* trace;               // debug counters (may be null): pops, pushes, activations, cas retries
};

#define TDX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void q_push(const AsyncCtl& c, uint32_t tile) {
    const unsigned long long t = __hip_atomic_fetch_add(c.tail, 1ull, TDX_AGENT);
    uint32_t* slot = c.ring + (t & c.mask);
    for (unsigned spins = 0; __hip_atomic_load(slot, TDX_AGENT) != 0u; spins++) {   // previous lap not consumed yet
        __builtin_amdgcn_s_sleep(2);
        if (spins > c.spin_limit) { __hip_atomic_store(c.error, 2u, TDX_AGENT); return; }
    }
    __hip_atomic_store(slot, tile + 1u, TDX_AGENT);
    if (c.trace) atomicAdd(c.trace + 1, 1ull);
}
__device__ __forceinline__ int q_pop(const AsyncCtl& c) {
    const unsigned long long h = __hip_atomic_fetch_add(c.head, 1ull, TDX_AGENT);
    uint32_t* slot = c.ring + (h & c.mask);
    for (unsigned spins = 0;; spins++) {
        const uint32_t v = __hip_atomic_load(slot, TDX_AGENT);
        if (v != 0u) {
            __hip_atomic_store(slot, 0u, TDX_AGENT);
            if (c.trace) atomicAdd(c.trace + 0, 1ull);
            return int(v - 1u);
        }
        if (__hip_atomic_load(c.pending, TDX_AGENT) <= 0) return -1;   // nothing queued or running: no push can follow
        if (__hip_atomic_load(c.error, TDX_AGENT) != 0u) return -1;
        __builtin_amdgcn_s_sleep(8);
        if (spins > c.spin_limit) { __hip_atomic_store(c.error, 1u, TDX_AGENT); return -1; }
    }
}
__device__ __forceinline__ void q_activate(const AsyncCtl& c, uint32_t tile) {
    if (c.trace) atomicAdd(c.trace + 2, 1ull);
    for (unsigned tries = 0;; tries++) {
        if (tries > c.spin_limit) { __hip_atomic_store(c.error, 3u, TDX_AGENT); return; }
        const uint32_t s = __hip_atomic_load(c.state + tile, TDX_AGENT);
        const uint32_t want = (s == 0u) ? 1u : (s == 2u ? 3u : s);
        uint32_t expect = s;
        if (__hip_atomic_compare_exchange_strong(c.state + tile, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (s == 0u) { __hip_atomic_fetch_add(c.pending, 1, TDX_AGENT); q_push(c, tile); }
            return;
        }
    }
}

template <class Op>
__global__ __launch_bounds__(NTHR, 4) void relax_async_kernel(Op op, TileGeom g, AsyncCtl c, unsigned long long* __restrict__ dbg) {
    using T = typename Op::T;
    __shared__ T sV[LH * LP];
    __shared__ TileLds L;
    __shared__ int sTile;
    const int tid = threadIdx.x;
    for (unsigned served = 0;; served++) {
        if (tid == 0) {
            int t = q_pop(c);
            if (served > c.spin_limit) { __hip_atomic_store(c.error, 5u, TDX_AGENT); t = -1; }
            if (t >= 0) {
                __hip_atomic_exchange(c.state + t, 2u, TDX_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's stale L1 lines before the tile is loaded
            }
            sTile = t;
        }
        __syncthreads();
        const int tile = sTile;
        if (tile < 0) break;
        const int res = relax_tile(op, g, tile, true, sV, L, dbg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-back ...
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // ... then one lane writes the L2 back
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (res & RES_CHANGED) {
                const int ddx[8] = {0, 0, -1, 1, -1, 1, -1, 1};
                const int ddy[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
                const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
                for (int k = 0; k < 8; k++) {
                    if (!((res >> k) & 1)) continue;
                    const int ntx = tx + ddx[k], nty = ty + ddy[k];
                    if (ntx >= 0 && ntx < g.tiles_x && nty >= 0 && nty < g.tiles_y) q_activate(c, uint32_t(nty * g.tiles_x + ntx));
                }
            }
            for (unsigned tries = 0;; tries++) {   // leave the running state
                if (tries > c.spin_limit) { __hip_atomic_store(c.error, 4u, TDX_AGENT); break; }
                const uint32_t s = __hip_atomic_load(c.state + tile, TDX_AGENT);
                const bool again = (s == 3u) || (res & RES_CAPPED);
                uint32_t expect = s;
                if (__hip_atomic_compare_exchange_strong(c.state + tile, &expect, again ? 1u : 0u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (again) q_push(c, uint32_t(tile));
                    else __hip_atomic_fetch_sub(c.pending, 1, TDX_AGENT);
                    break;
                }
            }
        }
    }
}

// activation flags -> queue: ring[i] = tile + 1, state = queued; clears the flags; count must be zero on entry
static __global__ __launch_bounds__(256) void async_fill_kernel(uint32_t* __restrict__ flags, int ntiles, AsyncCtl c, unsigned long long* __restrict__ count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bool act = false;
    if (t < ntiles) {
        act = flags[t] != 0u;
        if (act) flags[t] = 0u;
        c.state[t] = act ? 1u : 0u;
    }
    const unsigned long long ballot = __ballot(act);
    if (ballot == 0) return;
    const int lane = int(threadIdx.x & 63), leader = __ffsll((long long)ballot) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned long long)__popcll(ballot));
    base = __shfl(base, leader, 64);
    if (act) c.ring[(base + (unsigned long long)__popcll(ballot & ((1ull << lane) - 1ull))) & c.mask] = uint32_t(t) + 1u;
}
static __global__ void async_start_kernel(AsyncCtl c, const unsigned long long* __restrict__ count) {
    *c.head = 0ull;
    *c.tail = *count;
    *c.pending = int(*count);
    *c.error = 0u;
}

// round 0: activation flags -> tile list (the flags stay: the relaxation kernel consumes them); count must be zero on entry
static __global__ __launch_bounds__(256) void first_list_kernel(const uint32_t* __restrict__ flags, int ntiles, uint32_t* __restrict__ list,
                                                                unsigned long long* __restrict__ count, const uint32_t* __restrict__ remap) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool on = t < ntiles && flags[t] != 0u;
    const unsigned long long pos = block_reserve(on ? 1u : 0u, count);   // one atomic per block: 1024 wave-level atomics on one counter cost 50 us
    uint32_t e = uint32_t(t);
    if (on && remap != nullptr && entry_tile(remap[t]) == t) e = remap[t];   // (a macro block's first tile carries the block bits: TileGeom::remap)
    if (on) list[pos] = e;
}

// Round 0 of TWO schedules that start from the same activation flags (the two level fields of a flat iteration, flats.hpp), in ONE launch: both first flag
// halves = flags0, both second halves cleared, both first lists built.  Replaces two copies, two fills and two first_list_kernel launches; the count rings
// must be zero on entry (flatk::prepare_kernel).
static __global__ __launch_bounds__(256) void pair_start_kernel(const uint32_t* __restrict__ flags0, int ntiles, uint32_t* __restrict__ flagsA, uint32_t* __restrict__ flagsA1,
                                                                uint32_t* __restrict__ listA, unsigned long long* __restrict__ countA, uint32_t* __restrict__ flagsB,
                                                                uint32_t* __restrict__ flagsB1, uint32_t* __restrict__ listB, unsigned long long* __restrict__ countB,
                                                                const uint32_t* __restrict__ remapA, const uint32_t* __restrict__ remapB) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t f = t < ntiles ? flags0[t] : 0u;
    // (macro blocks, TileGeom::remap: only a block's first tile is a node of the schedule - every tile of a block is flagged by the classification, so it is)
    const uint32_t ra = (remapA != nullptr && t < ntiles) ? remapA[t] : uint32_t(t), rb = (remapB != nullptr && t < ntiles) ? remapB[t] : uint32_t(t);
    const uint32_t fa = entry_tile(ra) != t ? 0u : f, fb = entry_tile(rb) != t ? 0u : f;
    if (t < ntiles) { flagsA[t] = fa; flagsB[t] = fb; flagsA1[t] = 0u; flagsB1[t] = 0u; }
    const unsigned long long pa = block_reserve(fa != 0u ? 1u : 0u, countA);
    if (fa != 0u) listA[pa] = ra;    // (a block's first tile: with the block bits)
    const unsigned long long pb = block_reserve(fb != 0u ? 1u : 0u, countB);
    if (fb != 0u) listB[pb] = rb;
}
// Round 0 with EVERY tile active (the first relaxation of a PitRemove level): flags, list, count ring and the second flag half in one launch instead of
// three fills and first_list_kernel.
static __global__ __launch_bounds__(256) void all_start_kernel(int ntiles, uint32_t* __restrict__ flags, uint32_t* __restrict__ flags1, uint32_t* __restrict__ list,
                                                               unsigned long long* __restrict__ counts) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < ntiles) { flags[t] = FLAG_FULL; flags1[t] = 0u; list[t] = uint32_t(t); }
    if (t < 2 * COUNT_RING) counts[t] = t == 0 ? (unsigned long long)ntiles : 0ull;
}

// The ring of per-round counts is full: the pending round's size moves to the front of a cleared ring.
static __global__ __launch_bounds__(256) void ring_wrap_kernel(unsigned long long* __restrict__ counts, int r) {
    const unsigned long long pending = counts[r];
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * COUNT_RING; i += 256) counts[i] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) counts[0] = pending;
}

// a schedule that was stopped with tiles still active: their flags (the schedule's second flag half) back into the first, where a new schedule starts
static __global__ __launch_bounds__(256) void flags_fold_kernel(uint32_t* __restrict__ f_first, uint32_t* __restrict__ f_second, int ntiles) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const uint32_t b = f_second[t];
    if (b) { f_second[t] = 0u; if (b > f_first[t]) f_first[t] = b; }
}
static __global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Sched {
    uint32_t* flags;              // [ntiles] activation flags (input: tiles active in round 0; all zero again when a run returns)
    uint32_t* list;               // [SCHED_LIST_WORDS * ntiles]: tile lists of even / odd rounds, activation flags of odd rounds
    unsigned long long* counts;   // [2 * COUNT_RING]: per-round active counts, then per-round work cursors
};
constexpr size_t SCHED_LIST_WORDS = 3;

}  // namespace tilek

// Finishes a relaxation with the asynchronous worklist from the current activation flags (one launch on `s`).
// Returns TDX_OK and sets *gave_up when a bounded spin gave up (the flags are then reset to "everything active").
template <class Op>
static int tile_relax_async_finish(tdx_context* ctx, hipStream_t s, Op op, tilek::TileGeom g, tilek::Sched sc, int ring_slot, uint64_t* host_mail,
                                   unsigned long long* dbg, bool* gave_up) {
    using namespace tilek;
    const int ntiles = g.tiles_x * g.tiles_y;
    const unsigned cgrid = tdx_blocks_for(size_t(ntiles), 256);
    uint32_t cap = 1;
    while (cap <= uint32_t(ntiles)) cap <<= 1;
    uint32_t* ring = static_cast<uint32_t*>(ctx->scratch(ring_slot, size_t(cap) * 4));
    if (!ring) return TDX_ERR_NOMEM;
    AsyncCtl c;
    c.ring = ring; c.mask = cap - 1u;
    c.head = sc.counts + 0; c.tail = sc.counts + 1;   // control words at the start of the counts area
    c.pending = reinterpret_cast<int*>(sc.counts + 2);
    c.error = reinterpret_cast<unsigned*>(sc.counts + 3);
    c.state = sc.list;   // the round schedule's tile list doubles as the per-tile state
    c.spin_limit = getenv("TDX_ASYNC_SPIN") ? unsigned(atol(getenv("TDX_ASYNC_SPIN"))) : (1u << 21);
    c.trace = dbg ? dbg + 2 : nullptr;
    unsigned long long* count = sc.counts + 4;
    TDX_HIP_CHECK(ctx, hipMemsetAsync(sc.counts, 0, 8 * sizeof(unsigned long long), s));
    TDX_HIP_CHECK(ctx, hipMemsetAsync(ring, 0, size_t(cap) * 4, s));
    hipLaunchKernelGGL(async_fill_kernel, dim3(cgrid), dim3(256), 0, s, sc.flags, ntiles, c, count);
    hipLaunchKernelGGL(async_start_kernel, dim3(1), dim3(1), 0, s, c, count);
    const unsigned grid = unsigned(std::min(ntiles, 4 * ctx->num_cus));
    hipLaunchKernelGGL((relax_async_kernel<Op>), dim3(grid), dim3(NTHR), 0, s, op, g, c, dbg);
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(host_mail, sc.counts, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
    const unsigned err = unsigned(host_mail[3] & 0xffffffffull);
    *gave_up = err != 0;
    if (err) {
        fprintf(stderr, "taudem_amd: asynchronous tile worklist gave up (code %u, head %llu tail %llu pending %d); continuing with the round schedule\n", err,
                (unsigned long long)host_mail[0], (unsigned long long)host_mail[1], int(host_mail[2] & 0xffffffffull));
        hipLaunchKernelGGL(fill_u32_kernel, dim3(cgrid), dim3(256), 0, s, sc.flags, FLAG_FULL, size_t(ntiles));
    }
    return TDX_OK;
}

// The round schedule of ONE relaxation as a resumable object: batches of rounds (one launch each) are enqueued on `s`;
// after the stream has been synchronised collect() reads the batch's per-round tile counts and notes the first empty
// round.  Two runners on two streams interleave two independent relaxations (tile_relax_run_pair).
template <class Op>
struct RoundRunner {
    tdx_context* ctx; hipStream_t s; Op op; tilek::TileGeom g; tilek::Sched sc; uint64_t* h; unsigned long long* dbg;
    int ntiles; unsigned cgrid, grid_full, grid_small;
    bool print_counts = false;
    int ring_len;                        // rounds that fit the count ring before it wraps (TDX_RELAX_RING: test hook)
    bool lds_variant;                    // TDX_RELAX_LDS=1: the LDS-resident tile kernel instead of the register-resident one
    unsigned pull_max;                   // list entries per cursor atomic in large rounds (TDX_RELAX_PULL: test hook)
    // Batches are enqueued and collected through a two-slot queue, so that the host can read the counts of batch k while batch
    // k + 1 is already running (drive() / tile_relax_run_pair): a host round trip between batches costs ~40 us of idle GPU, a
    // relaxation has 5-7 of them, a step a dozen relaxations.  parity: flag half of the first round AFTER the collected batches.
    int r = 0, parity = 0, batch = 4;
    int r_enq = 0, parity_enq = 0, n_enq = 0, n_col = 0, fl_batch[2] = {0, 0};
    bool polled[2] = {false, false};     // the slot's batch reports through the host slot itself (no copy, no event)
    uint64_t fl_tag[2] = {0, 0};         // ... tagged with the batch's sequence number (TileGeom::cnt_tag)
    int ev_base = 0;                     // ctx->ev_batch[ev_base + slot]
    int batch_max = 64;                  // drive() / the pair lower it: with two batches in flight a whole batch of empty rounds follows the last one
    bool done = false;
    int64_t rounds = 0, launches = 0;
    bool short_tail = getenv("TDX_RELAX_LONG_TAIL") == nullptr;   // (A/B hook)
    unsigned long long last_count = 0;   // active tiles of the last non-empty round seen
    // another tile kernel on the same schedule (tile_dep.hpp): launches one round; empty = the relaxation kernel of `op`
    // (rg: the runner's geometry of THIS launch - the caller's geometry plus where the round reports its size; the tile kernel must be given rg)
    std::function<void(const tilek::TileGeom& rg, unsigned grid, hipStream_t st, const uint32_t* list, unsigned long long* count, uint32_t* flags_cur,
                       uint32_t* flags_next, uint32_t* list_next, unsigned pull_max)> custom_launch;
    RoundRunner(tdx_context* c, hipStream_t st, Op o, tilek::TileGeom geom, tilek::Sched sched, uint64_t* host_mail, unsigned long long* d)
        : ctx(c), s(st), op(o), g(geom), sc(sched), h(host_mail), dbg(d) {
        ntiles = g.tiles_x * g.tiles_y;
        cgrid = tdx_blocks_for(size_t(ntiles), 256);
        grid_full = unsigned(std::min(ntiles, 8 * ctx->num_cus));
        static const int gs_env = getenv("TDX_RELAX_GRID_SMALL") ? std::max(1, atoi(getenv("TDX_RELAX_GRID_SMALL"))) : 0;   // (A/B hook)
        grid_small = unsigned(std::min(ntiles, gs_env ? gs_env : ctx->num_cus));
        const char* e = getenv("TDX_RELAX_RING");
        ring_len = e ? std::max(3, std::min(atoi(e), tilek::COUNT_RING)) : tilek::COUNT_RING;
        lds_variant = getenv("TDX_RELAX_LDS") != nullptr;
        const char* pm = getenv("TDX_RELAX_PULL");
        pull_max = pm ? unsigned(std::max(1, std::min(atoi(pm), tilek::PULL_MAX))) : 16u;
    }
    uint32_t* list_of(int p) const { return sc.list + size_t(p) * size_t(ntiles); }
    uint32_t* flags_of(int p) const { return p ? sc.list + 2 * size_t(ntiles) : sc.flags; }
    bool prestarted = false;             // round 0 (flags, second flag half, first list, count ring) was set up by the caller (tilek::pair_start_kernel)
    bool all_tiles = false;              // round 0 = every tile, whatever sc.flags holds (tilek::all_start_kernel)
    int start() {
        using namespace tilek;
        r = 0; parity = 0;
        r_enq = 0; parity_enq = 0; n_enq = 0; n_col = 0;
        if (prestarted) return TDX_OK;
        if (all_tiles) {
            hipLaunchKernelGGL(all_start_kernel, dim3(tdx_blocks_for(size_t(std::max(ntiles, 2 * COUNT_RING)), 256)), dim3(256), 0, s, ntiles, sc.flags, flags_of(1), list_of(0), sc.counts);
            return TDX_OK;
        }
        TDX_HIP_CHECK(ctx, hipMemsetAsync(sc.counts, 0, size_t(2 * COUNT_RING) * sizeof(unsigned long long), s));
        TDX_HIP_CHECK(ctx, hipMemsetAsync(flags_of(1), 0, size_t(ntiles) * 4, s));
        hipLaunchKernelGGL(first_list_kernel, dim3(cgrid), dim3(256), 0, s, sc.flags, ntiles, list_of(0), sc.counts, g.remap);
        r = 0; parity = 0;
        r_enq = 0; parity_enq = 0; n_enq = 0; n_col = 0;
        return TDX_OK;
    }
    int enqueue() {   // at most two batches may be in flight
        using namespace tilek;
        // A monotone relaxation ends; a schedule that does not is a defect (an operator that raises a value, a torn activation flag), and a
        // silent endless loop is the worst way to report it: generous bound (a front can wind through a tile a few dozen times), then an error.
        static const long long round_cap = getenv("TDX_RELAX_MAX_ROUNDS") ? atoll(getenv("TDX_RELAX_MAX_ROUNDS")) : 0;
        if (rounds > (round_cap > 0 ? round_cap : 64ll * ntiles + 65536ll))
            return tdx_fail(ctx, TDX_ERR_HIP, "tile schedule: no fixed point after " + std::to_string(rounds) + " rounds (" + std::to_string(ntiles) + " tiles)");
        if (batch > ring_len - 2) batch = ring_len - 2;
        if (batch > TDX_MAIL_RUN_SLOT) batch = TDX_MAIL_RUN_SLOT;   // a batch reports into ONE host slot of TDX_MAIL_RUN_SLOT words, whatever a caller (TDX_SWEEP_EAGER_ROUNDS) asks for
        if (batch < 1) batch = 1;
        if (r_enq + batch + 1 > ring_len) {   // counts[r + batch] (written by the batch's last round) must be inside the ring
            hipLaunchKernelGGL(ring_wrap_kernel, dim3(1), dim3(256), 0, s, sc.counts, r_enq);
            r_enq = 0;
        }
        const int slot = n_enq & 1;
        uint64_t* hs = h + slot * TDX_MAIL_RUN_SLOT;
        const int r = r_enq, parity = parity_enq;   // (shadow the collected state inside this function)
        const bool timed = ctx->kernel_timing && s == ctx->stream;
        // the tail of a relaxation: a small grid launches faster (any grid size is correct, the cursor covers the list)
        const unsigned grid = (rounds > 0 && last_count <= 256ull) ? grid_small : grid_full;
        // the rounds of this batch report their sizes straight into the host slot (TDX_RELAX_COUNT_COPY=1: a copy after the batch instead - A/B hook)
        static const bool count_copy = getenv("TDX_RELAX_COUNT_COPY") != nullptr;
        if (!count_copy) {
            // Every batch of the context has its own sequence number: the last batch of an EARLIER runner (empty rounds, left in flight when that runner
            // saw its first empty round) may still be storing zeros into this slot; they carry an older number and are not taken for this batch's reports.
            uint32_t seq = ++ctx->run_seq;
            if (seq == 0xffffffffu) seq = ++ctx->run_seq;   // (never the pattern of an unwritten slot)
            fl_tag[slot] = uint64_t(seq) << 32;
            for (int b = 0; b < batch; b++) hs[b] = ~0ull;   // (the slot's previous batch has been collected)
            g.cnt_host = reinterpret_cast<unsigned long long*>(hs);
            g.cnt_dev = sc.counts + r;
            g.cnt_tag = fl_tag[slot];
        } else { g.cnt_host = nullptr; g.cnt_dev = nullptr; g.cnt_tag = 0ull; }
        for (int b = 0; b < batch; b++) {
            const int p = (parity + b) & 1;
            const int sp = timed ? ctx->span_begin(TDX_K_TILEK) : -1;   // this kernel alone: what bench.py's roofline is computed from
            if (custom_launch)
                custom_launch(g, grid, s, list_of(p), sc.counts + r + b, flags_of(p), flags_of(p ^ 1), list_of(p ^ 1), pull_max);
            else if (lds_variant)
                hipLaunchKernelGGL((relax_kernel<Op, false>), dim3(grid), dim3(NTHR), 0, s, op, g, list_of(p), sc.counts + r + b, flags_of(p), flags_of(p ^ 1),
                                   list_of(p ^ 1), pull_max, dbg, 0u);
            else {
                // with macro blocks (TileGeom::blk_k): four more workgroups per CU in the same launch serve the blocks of the round (1 / 2 / 4 / 8 / 16 per CU: 22.07 / 21.97 / 21.86 / 21.81 / 21.87 ms per 16384^2 step, profiles/r06g_macro_wgs.txt)
                static const int macro_wgs = getenv("TDX_MACRO_WGS") ? std::max(1, atoi(getenv("TDX_MACRO_WGS"))) : 4;   // (A/B hook: block workgroups per CU)
                const unsigned gm = (has_macro<Op>::value && g.blk_k != nullptr) ? unsigned(macro_wgs * ctx->num_cus) : 0u;
                hipLaunchKernelGGL((relax_kernel<Op, true>), dim3(grid + gm), dim3(NTHR), 0, s, op, g, list_of(p), sc.counts + r + b, flags_of(p), flags_of(p ^ 1),
                                   list_of(p ^ 1), pull_max, dbg, gm ? grid : 0u);
            }
            ctx->span_end(sp);
            if (timed && ctx->cur_stats) ctx->cur_stats->launches[TDX_K_TILEK]++;
        }
        launches += batch;
        fl_batch[slot] = batch;
        polled[slot] = !count_copy;
        if (count_copy) {
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(hs, sc.counts + r, size_t(batch) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            hipEvent_t& ev = ctx->ev_batch[ev_base + slot];
            if (!ev) TDX_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            TDX_HIP_CHECK(ctx, hipEventRecord(ev, s));   // (an event record is a barrier packet: a ~6 us bubble between two batches)
        }
        r_enq += batch;
        parity_enq = (parity_enq + batch) & 1;
        n_enq++;
        if (batch < batch_max) batch = std::min(2 * batch, batch_max);
        // Two batches are in flight, so a whole batch of empty rounds (4.5 us each) follows the last productive one.  Once the rounds are
        // down to a handful of tiles the end is near (a lone front is followed inside ONE launch: round_driver's solo hand-over), and
        // short batches cost nothing there: the host has the counts of batch k long before batch k + 1 is through.
        if (short_tail && rounds > 0 && last_count <= 8ull) batch = std::min(batch, 4);
        static const int tail_batch = getenv("TDX_RELAX_TAIL_BATCH") ? std::max(2, atoi(getenv("TDX_RELAX_TAIL_BATCH"))) : 0;   // (A/B hook)
        if (tail_batch && rounds > 0 && last_count <= 256ull) batch = std::min(batch, tail_batch);
        return TDX_OK;
    }
    int wait_oldest() {   // the counts of the oldest batch in flight are on the host
        const int slot = n_col & 1;
        if (!polled[slot]) {
            TDX_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_batch[ev_base + slot]));
            return TDX_OK;
        }
        // the batch's rounds wrote their sizes themselves (round_driver): once its LAST round has reported - it has started, so every earlier round has
        // ended - the batch's counts are complete.  No event, no barrier packet in the stream; bounded like every other wait of the library.
        const volatile uint64_t* last = h + slot * TDX_MAIL_RUN_SLOT + (fl_batch[slot] - 1);
        const uint64_t tag = fl_tag[slot];
        auto mine = [&]() { return (*last & 0xffffffff00000000ull) == tag; };
        static const double limit_s = getenv("TDX_COMM_TIMEOUT") ? atof(getenv("TDX_COMM_TIMEOUT")) : 600.0;   // the library's bound for every wait
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; !mine(); spins++) {
            if ((spins & 1023u) == 1023u) {
                if (hipStreamQuery(s) == hipSuccess && !mine())
                    return tdx_fail(ctx, TDX_ERR_HIP, "tile schedule: a round ended without reporting its size");
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
                    return tdx_fail(ctx, TDX_ERR_HIP, "tile schedule: no report from a batch of rounds after " + std::to_string(int(limit_s)) + " s");
            }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);   // the earlier rounds' reports (read by collect()) were stored before the last one's
        return TDX_OK;
    }
    void collect() {   // the oldest batch in flight; its counts must have arrived (wait_oldest() or a synchronised stream)
        const int slot = n_col & 1, nb = fl_batch[slot];
        const volatile uint64_t* hs = h + slot * TDX_MAIL_RUN_SLOT;
        for (int b = 0; b < nb; b++) {
            const uint64_t cnt = polled[slot] ? (hs[b] & 0xffffffffull) : hs[b];   // (a polled slot carries the batch's sequence number above the count)
            if (cnt == 0) { done = true; break; }
            last_count = cnt;
            rounds++;
            if (print_counts) fprintf(stderr, " %llu", (unsigned long long)cnt);   // TDX_DEBUG_ROUNDS: active tiles per round
        }
        r += nb;
        parity = (parity + nb) & 1;
        n_col++;
    }
    // to the fixed point with two batches in flight; on return one batch of empty rounds may still be running on `s`
    static int pipelined_batch_max() {
        static const int v = getenv("TDX_RELAX_BATCH") ? std::max(2, std::min(atoi(getenv("TDX_RELAX_BATCH")), 64)) : 16;
        return v;
    }
    int drive() {
        batch_max = pipelined_batch_max();
        int rc = start();
        if (rc != TDX_OK) return rc;
        rc = enqueue();
        if (rc != TDX_OK) return rc;
        while (!done) {
            rc = enqueue();
            if (rc != TDX_OK) return rc;
            rc = wait_oldest();
            if (rc != TDX_OK) return rc;
            collect();
        }
        return TDX_OK;
    }
};

// One batch of FUSED rounds for two relaxations in lockstep (relax_pair_kernel): A's batch bookkeeping leads, B's follows.
template <class Op>
static int enqueue_fused(RoundRunner<Op>& A, RoundRunner<Op>& B) {
    using namespace tilek;
    tdx_context* ctx = A.ctx;
    hipStream_t s = A.s;
    static const long long round_cap = getenv("TDX_RELAX_MAX_ROUNDS") ? atoll(getenv("TDX_RELAX_MAX_ROUNDS")) : 0;
    if (std::max(A.rounds, B.rounds) > (round_cap > 0 ? round_cap : 64ll * A.ntiles + 65536ll))
        return tdx_fail(ctx, TDX_ERR_HIP, "tile schedule: no fixed point after " + std::to_string(std::max(A.rounds, B.rounds)) + " rounds (" + std::to_string(A.ntiles) + " tiles)");
    int batch = std::min(A.batch, A.ring_len - 2);
    if (A.r_enq + batch + 1 > A.ring_len) {   // both rings wrap together
        hipLaunchKernelGGL(ring_wrap_kernel, dim3(1), dim3(256), 0, s, A.sc.counts, A.r_enq);
        hipLaunchKernelGGL(ring_wrap_kernel, dim3(1), dim3(256), 0, s, B.sc.counts, B.r_enq);
        A.r_enq = B.r_enq = 0;
    }
    const int slot = A.n_enq & 1, r = A.r_enq, parity = A.parity_enq;
    auto grid_of = [](const RoundRunner<Op>& R) { return R.done ? 0u : ((R.rounds > 0 && R.last_count <= 256ull) ? R.grid_small : R.grid_full); };
    const unsigned gA = grid_of(A), gB = grid_of(B);
    for (int b = 0; b < batch; b++) {
        const int p = (parity + b) & 1;
        const RoundArgs ra{A.list_of(p), A.sc.counts + r + b, A.flags_of(p), A.flags_of(p ^ 1), A.list_of(p ^ 1)};
        const RoundArgs rb{B.list_of(p), B.sc.counts + r + b, B.flags_of(p), B.flags_of(p ^ 1), B.list_of(p ^ 1)};
        hipLaunchKernelGGL((relax_pair_kernel<Op>), dim3(std::max(1u, gA + gB)), dim3(NTHR), 0, s, A.op, B.op, A.g, ra, rb, gA, A.pull_max);
    }
    A.launches += batch;
    A.fl_batch[slot] = B.fl_batch[slot] = batch;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(A.h + slot * TDX_MAIL_RUN_SLOT, A.sc.counts + r, size_t(batch) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(B.h + slot * TDX_MAIL_RUN_SLOT, B.sc.counts + r, size_t(batch) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    hipEvent_t& ev = ctx->ev_batch[A.ev_base + slot];
    if (!ev) TDX_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    TDX_HIP_CHECK(ctx, hipEventRecord(ev, s));
    A.r_enq += batch; B.r_enq = A.r_enq;
    A.parity_enq = B.parity_enq = (A.parity_enq + batch) & 1;
    A.n_enq++; B.n_enq = A.n_enq;
    A.batch = batch;
    if (A.batch < A.batch_max) A.batch = std::min(2 * A.batch, A.batch_max);
    const unsigned long long live = std::max(A.done ? 0ull : A.last_count, B.done ? 0ull : B.last_count);
    if (A.short_tail && std::max(A.rounds, B.rounds) > 0 && live <= 8ull) A.batch = std::min(A.batch, 4);
    return TDX_OK;
}

// Two independent relaxations in lockstep on ONE stream, one launch per round for both (see relax_pair_kernel).
template <class Op>
static int tile_relax_run_fused(tdx_context* ctx, Op opA, tilek::Sched scA, Op opB, tilek::Sched scB, tilek::TileGeom g, int64_t* rounds_out,
                                int64_t* launches_out) {
    RoundRunner<Op> A(ctx, ctx->stream, opA, g, scA, ctx->h_mail + TDX_MAIL_RUN_A, nullptr), B(ctx, ctx->stream, opB, g, scB, ctx->h_mail + TDX_MAIL_RUN_B, nullptr);
    A.batch_max = B.batch_max = RoundRunner<Op>::pipelined_batch_max();
    int rc = A.start();
    if (rc != TDX_OK) return rc;
    rc = B.start();
    if (rc != TDX_OK) return rc;
    rc = enqueue_fused(A, B);
    if (rc != TDX_OK) return rc;
    while (!A.done || !B.done) {   // two batches in flight: the host reads one batch's counts while the next one runs
        rc = enqueue_fused(A, B);
        if (rc != TDX_OK) return rc;
        rc = A.wait_oldest();
        if (rc != TDX_OK) return rc;
        if (!A.done) A.collect(); else A.n_col++;   // (a schedule that has ended has written no counts: its slot is only stepped over)
        if (!B.done) B.collect(); else B.n_col++;
    }
    if (rounds_out) *rounds_out += A.rounds + B.rounds;
    if (launches_out) *launches_out += A.launches;
    return TDX_OK;
}

// Two independent relaxations (e.g. the two level fields of flat resolution) side by side on two streams: the rounds of
// one fill the workgroup slots the other leaves idle.  Work already enqueued on the context's stream is waited for.
// flags0 != nullptr: BOTH relaxations start from these activation flags (scA.flags / scB.flags need not be filled in) and both count rings are zero: round 0
// of both schedules is set up by one launch (tilek::pair_start_kernel).
template <class Op>
// gB (optional): the second relaxation's geometry where it differs from the first's (its own macro blocks).
static int tile_relax_run_pair(tdx_context* ctx, Op opA, tilek::Sched scA, Op opB, tilek::Sched scB, tilek::TileGeom g, int64_t* rounds_out,
                               int64_t* launches_out, const uint32_t* flags0 = nullptr, const tilek::TileGeom* gB = nullptr) {
    if (flags0) {
        const int ntiles = g.tiles_x * g.tiles_y;
        hipLaunchKernelGGL(tilek::pair_start_kernel, dim3(tdx_blocks_for(size_t(ntiles), 256)), dim3(256), 0, ctx->stream, flags0, ntiles, scA.flags, scA.list + 2 * size_t(ntiles),
                           scA.list, scA.counts, scB.flags, scB.list + 2 * size_t(ntiles), scB.list, scB.counts, g.remap, gB ? gB->remap : g.remap);
    }
    if (!ctx->stream2) {
        TDX_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        TDX_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    }
    TDX_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    RoundRunner<Op> A(ctx, ctx->stream, opA, g, scA, ctx->h_mail + TDX_MAIL_RUN_A, nullptr), B(ctx, ctx->stream2, opB, gB ? *gB : g, scB, ctx->h_mail + TDX_MAIL_RUN_B, nullptr);
    B.ev_base = 2;
    A.prestarted = B.prestarted = flags0 != nullptr;
    A.batch_max = B.batch_max = RoundRunner<Op>::pipelined_batch_max();
    int rc = A.start();
    if (rc != TDX_OK) return rc;
    rc = B.start();
    if (rc != TDX_OK) return rc;
    // (Making both first lists run before either first round - the second field's list kernel can get stuck for a whole round behind the first field's
    // first launch, profiles/r05zzz_timeline_d8_16384.txt - changes nothing: 12.59-12.66 ms of d8flowdir either way, the device is busy in both orders.)
    rc = A.enqueue();
    if (rc != TDX_OK) return rc;
    rc = B.enqueue();
    if (rc != TDX_OK) return rc;
    while (!A.done || !B.done) {   // two batches in flight per relaxation: the host reads one batch's counts while the next one runs
        if (!A.done) { rc = A.enqueue(); if (rc != TDX_OK) return rc; }
        if (!B.done) { rc = B.enqueue(); if (rc != TDX_OK) return rc; }
        if (!A.done) { rc = A.wait_oldest(); if (rc != TDX_OK) return rc; A.collect(); }
        if (!B.done) { rc = B.wait_oldest(); if (rc != TDX_OK) return rc; B.collect(); }
    }
    // what follows on the context's stream comes after everything on the second one (incl. its last batch of empty rounds)
    TDX_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream2));
    TDX_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
    if (rounds_out) *rounds_out += A.rounds + B.rounds;
    if (launches_out) *launches_out += A.launches + B.launches;
    return TDX_OK;
}

// The round schedule for at most ~max_rounds rounds (one batch in flight, batches of max_rounds): on return *active_left says whether tiles are still
// active - their flags are then in sc.flags, where the next schedule (this function again, after a halo exchange has added its own activations) starts.
// For multi-strip callers whose fronts are many and independent (the upstream closure of outlets): exchanging every few dozen rounds lets the fronts of
// all strips advance side by side instead of strip-local fixed point after strip-local fixed point.
template <class Op>
static int tile_relax_run_bounded(tdx_context* ctx, Op op, tilek::TileGeom g, tilek::Sched sc, int max_rounds, bool* active_left, int64_t* rounds_out,
                                  int64_t* launches_out, bool all_tiles = false) {
    RoundRunner<Op> run(ctx, ctx->stream, op, g, sc, ctx->h_mail + TDX_MAIL_RUN_A, nullptr);
    run.all_tiles = all_tiles;
    run.batch = run.batch_max = std::max(2, std::min(max_rounds, 64));
    static const int debug_level = getenv("TDX_DEBUG_ROUNDS") ? atoi(getenv("TDX_DEBUG_ROUNDS")) : 0;   // (as in tile_relax_run: active tiles per round on stderr)
    run.print_counts = debug_level > 0;
    if (debug_level == 2) fprintf(stderr, "\nrounds(%d tiles, at most %d):", run.ntiles, max_rounds);
    *active_left = false;
    int rc = run.start();
    if (rc != TDX_OK) return rc;
    while (!run.done) {
        rc = run.enqueue();
        if (rc != TDX_OK) return rc;
        rc = run.wait_oldest();
        if (rc != TDX_OK) return rc;
        run.collect();
        if (!run.done && run.rounds >= max_rounds) { *active_left = true; break; }
    }
    if (*active_left && run.parity)   // the active tiles' flags sit in the second flag half: back into the first
        hipLaunchKernelGGL(tilek::flags_fold_kernel, dim3(tdx_blocks_for(size_t(run.ntiles), 256)), dim3(256), 0, ctx->stream, run.flags_of(0), run.flags_of(1), run.ntiles);
    if (rounds_out) *rounds_out += run.rounds;
    if (launches_out) *launches_out += run.launches;
    return TDX_OK;
}

// Runs the relaxation until no tile is active.  `flags` must hold the initially active tiles.
// Default schedule: rounds.  TDX_RELAX_ASYNC=1 selects the asynchronous worklist (one launch); a worklist that gave
// up on a bounded spin continues with rounds (values only ever decrease, so restarting from "all tiles active" is safe).
template <class Op>
static int tile_relax_run(tdx_context* ctx, Op op, tilek::TileGeom g, tilek::Sched sc, int64_t* rounds_out, int64_t* launches_out, bool all_tiles = false) {
    using namespace tilek;
    hipStream_t s = ctx->stream;
    const int ntiles = g.tiles_x * g.tiles_y;
    // TDX_DEBUG_ROUNDS=1: schedule statistics and in-kernel cycle counters on stderr (the counters' atomics slow the
    // kernels down); =2: only the active tiles per round (no kernel-side instrumentation)
    static const int debug_level = getenv("TDX_DEBUG_ROUNDS") ? atoi(getenv("TDX_DEBUG_ROUNDS")) : 0;
    static const bool debug = debug_level == 1;
    // The worklist schedule is opt-in (TDX_RELAX_ASYNC=1): measured on MI355X it does not beat the round schedule
    // (the relaxation is bound by VALU work per tile, not by the number of launches) - see DESIGN.md 4.2.
    static const bool force_rounds = getenv("TDX_RELAX_ASYNC") == nullptr;
    unsigned long long* dbg = nullptr;
    if (debug) {
        dbg = reinterpret_cast<unsigned long long*>(ctx->d_mail) + TDX_MAIL_DBG_RELAX;
        TDX_HIP_CHECK(ctx, hipMemsetAsync(dbg, 0, 128, s));
    }
    int64_t rounds = 0, launches = 0;
    bool need_rounds = force_rounds;
    if (!force_rounds) {
        bool gave_up = false;
        int rc = tile_relax_async_finish(ctx, s, op, g, sc, TDX_S_Q, ctx->h_mail, dbg, &gave_up);
        if (rc != TDX_OK) return rc;
        launches += 1;
        rounds += 1;
        need_rounds = gave_up;
    }
    if (need_rounds) {
        RoundRunner<Op> run(ctx, s, op, g, sc, ctx->h_mail + TDX_MAIL_RUN_A, dbg);
        run.all_tiles = all_tiles && force_rounds;   // (the worklist schedule starts from the flags)
        run.print_counts = debug_level > 0;
        if (debug_level == 2) fprintf(stderr, "\nrounds(%d tiles):", ntiles);
        int rc = run.drive();
        if (rc != TDX_OK) return rc;
        rounds += run.rounds;
        launches += run.launches;
    }
    if (debug) {
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, dbg, 128, hipMemcpyDeviceToHost, s));
        TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
        fprintf(stderr, "tile_relax_run(%d tiles, %s): %lld rounds, %llu tile activations, %llu sweeps; queue pops %llu pushes %llu activations %llu\n", ntiles,
                need_rounds ? "rounds" : "worklist", (long long)rounds, (unsigned long long)ctx->h_mail[1], (unsigned long long)ctx->h_mail[0],
                (unsigned long long)ctx->h_mail[2], (unsigned long long)ctx->h_mail[3], (unsigned long long)ctx->h_mail[4]);
        const double na = double(ctx->h_mail[1] ? ctx->h_mail[1] : 1);
        fprintf(stderr, "    cycles per activation (s_memtime): load %.0f, sweeps %.0f, write-back %.0f\n", double(ctx->h_mail[6]) / na, double(ctx->h_mail[7]) / na,
                double(ctx->h_mail[8]) / na);
        fprintf(stderr, "    load split: issue tile loads %.0f, issue cell loads %.0f, wait+LDS stores %.0f, barrier %.0f\n", double(ctx->h_mail[9]) / na,
                double(ctx->h_mail[10]) / na, double(ctx->h_mail[11]) / na, double(ctx->h_mail[12]) / na);
        if (ctx->h_mail[14]) fprintf(stderr, "    macro blocks: %llu updates, %.0f cycles each (offers %.0f, stores + gain test %.0f, activations %.0f), at most %llu per workgroup and look\n",
                                    (unsigned long long)ctx->h_mail[14], double(ctx->h_mail[13]) / double(ctx->h_mail[14]), double(ctx->h_mail[9]) / double(ctx->h_mail[14]),
                                    double(ctx->h_mail[10]) / double(ctx->h_mail[14]), double(ctx->h_mail[11]) / double(ctx->h_mail[14]), (unsigned long long)ctx->h_mail[15]);
    }
    if (rounds_out) *rounds_out += rounds;
    if (launches_out) *launches_out += launches;
    return TDX_OK;
}
