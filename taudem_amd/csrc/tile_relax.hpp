// Tile relaxation engine shared by PitRemove (minimax-path surface) and flat resolution (the two
// breadth-first level fields of resolveflats).  All three are fixed points of a monotone, order-free
// neighbourhood operator
//       v(c) <- apply(cst(c), v(c), min over the neighbours selected by mask(c) of v(n))
// that only ever DEcreases v from +inf, so any schedule that runs to convergence yields the same
// values.  The schedule used on gfx950:
//
//   * the raster is cut into 64x64 tiles; a 256-thread workgroup stages a tile of v plus a one-cell
//     halo in LDS (66x67 x 4 B = 17.7 KB), every lane owns a 16-row column segment and relaxes it in
//     place ("chaotic" relaxation) in alternating downward / upward sweeps, looking only at cells whose
//     neighbourhood moved, until a sweep moves nothing (details at relax_kernel);
//   * changed cells are written back; a tile whose rim changed raises the activation flag of the
//     neighbouring tiles that see that rim in their halo;
//   * rounds (compact the flags into a tile list, relax the listed tiles) are enqueued in batches
//     with NO host round trip in between: the list length lives in device memory and the resident
//     workgroups pull tiles from a per-round cursor; the host reads back one batch of per-round counts
//     at a time and stops at the first empty round.
//
// HBM traffic per activation of a tile: one tile image of v (+ the per-cell constant) in, changed
// cells out.  Critical path: (longest dependency path measured in tiles) rounds x one launch.
#pragma once
#include <algorithm>
#include <cstdlib>

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
constexpr int MAX_SWEEPS = 32;

struct TileGeom {
    int nx, ny;             // raster (strip incl. halo rows) size
    int tiles_x, tiles_y;
    int y_own0, y_own1;     // rows [y_own0, y_own1) may be updated; others are read-only halo rows
    int max_sweeps;         // sweeps per activation before a tile yields (it re-activates itself)
};

static inline TileGeom make_geom(int nx, int ny, int y_own0, int y_own1) {
    TileGeom g;
    g.nx = nx; g.ny = ny;
    g.tiles_x = (nx + TS - 1) / TS; g.tiles_y = (ny + TS - 1) / TS;
    g.y_own0 = y_own0; g.y_own1 = y_own1;
    static const int ms = getenv("TDX_MAX_SWEEPS") ? atoi(getenv("TDX_MAX_SWEEPS")) : MAX_SWEEPS;
    g.max_sweeps = ms;
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

// Op interface:
//   using T;  (4 bytes)  static T inf();
//   T load(size_t idx) const;            value of an in-grid cell
//   void store(size_t idx, T v) const;   changed cell
//   void cell(size_t idx, T& cst, unsigned& mask) const;   per-cell constant + neighbour mask (0 = never updated)
//   static T apply(T cst, T own, T m);   new value (must be <= own)
//   static bool settled(T cst, T v);     v can never decrease again
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

template <class Op>
__global__ __launch_bounds__(NTHR, 5) void relax_kernel(Op op, TileGeom g, const uint32_t* __restrict__ list,
                                                    unsigned long long* __restrict__ count, uint32_t* __restrict__ flags_next,
                                                    unsigned long long* __restrict__ dbg) {
    using T = typename Op::T;
    static_assert(sizeof(T) == 4, "tile engine works on 4-byte values");
    __shared__ T sV[LH * LP];
    __shared__ int sRim;
    __shared__ unsigned long long sTop[2][NWAVE], sBot[2][NWAVE];
    __shared__ unsigned sAny[2][NWAVE];   // per wave: did any cell move in this sweep   // columns whose first / last segment row changed, per wave
    __shared__ unsigned sNext;
    const unsigned nact = unsigned(count[0]);
    unsigned long long* cursor = count + COUNT_RING;   // per-round work cursor: blocks pull tiles, so the load balances itself
    const int tid = threadIdx.x;
    const int lx = tid & 63;
    const int wv = tid >> 6;
    const int ry0 = wv * RPW;
    for (;;) {
        if (tid == 0) sNext = unsigned(atomicAdd(cursor, 1ull));
        __syncthreads();
        const unsigned it = sNext;
        if (it >= nact) break;
        const int tile = int(list[it]);
        const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
        const int x0 = tx * TS, y0 = ty * TS;
        if (tid == 0) sRim = 0;
        for (int e = tid; e < LH * LH; e += NTHR) {
            const int ly = e / LH, lxx = e - ly * LH;
            const int hx = x0 + lxx - 1, hy = y0 + ly - 1;
            T v = Op::inf();
            if (hx >= 0 && hx < g.nx && hy >= 0 && hy < g.ny) v = op.load(size_t(hy) * size_t(g.nx) + size_t(hx));
            sV[ly * LP + lxx] = v;
        }
        const int gx = x0 + lx;
        T cst[RPW];
        unsigned mk[RPW / 4] = {};
        unsigned live = 0;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const int gy = y0 + ry0 + r;
            T c = Op::inf();
            unsigned m = 0;
            if (gx < g.nx && gy >= g.y_own0 && gy < g.y_own1) op.cell(size_t(gy) * size_t(g.nx) + size_t(gx), c, m);
            cst[r] = c;
            mk[r >> 2] |= (m & 0xFFu) << (8 * (r & 3));
            if (m) live |= 1u << r;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPW; r++)
            if (Op::settled(cst[r], sV[(ry0 + r + 1) * LP + lx + 1])) live &= ~(1u << r);
        unsigned moved = 0;   // rows of this lane that changed during this activation

        bool any_change = false, capped = false;
        unsigned dirty = live;   // the halo may have moved since the last activation: evaluate everything once
        for (int iter = 0;; iter++) {
            unsigned chg = 0;
            const int cur = iter & 1;
            // Within a sweep the rows of a segment are visited in lockstep by the 64 lanes, so a cell that
            // moves in one row can hand its value to the three cells below (above) it in the SAME sweep:
            // `prev` = columns that moved in the row just visited.
            unsigned long long prev = 0ull;
            if ((iter & 1) == 0) {   // downward
#pragma unroll
                for (int r = 0; r < RPW; r++) {
                    const unsigned long long dil = prev | (prev << 1) | (prev >> 1);
                    const bool look = ((((dirty >> r) | unsigned(dil >> lx)) & (live >> r)) & 1u) != 0u;
                    bool ch = false;
                    if (look) {
                        const int c = (ry0 + r + 1) * LP + lx + 1;
                        const T own = sV[c];
                        const T m = masked_min<T>(mask_at(mk, r), Op::inf(), sV[c + 1], sV[c - LP + 1], sV[c - LP], sV[c - LP - 1], sV[c - 1],
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
                }
            } else {                 // upward
#pragma unroll
                for (int r = RPW - 1; r >= 0; r--) {
                    const unsigned long long dil = prev | (prev << 1) | (prev >> 1);
                    const bool look = ((((dirty >> r) | unsigned(dil >> lx)) & (live >> r)) & 1u) != 0u;
                    bool ch = false;
                    if (look) {
                        const int c = (ry0 + r + 1) * LP + lx + 1;
                        const T own = sV[c];
                        const T m = masked_min<T>(mask_at(mk, r), Op::inf(), sV[c + 1], sV[c - LP + 1], sV[c - LP], sV[c - LP - 1], sV[c - 1],
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
                }
            }
            // rows to look at next: 3x3 dilation of everything that moved in this sweep
            moved |= chg;
            const unsigned cl = __shfl_up(chg, 1, 64), cr = __shfl_down(chg, 1, 64);
            const unsigned h = chg | (lx > 0 ? cl : 0u) | (lx < 63 ? cr : 0u);
            unsigned nd = (h | (h << 1) | (h >> 1)) & ((1u << RPW) - 1u);
            const unsigned long long btop = __ballot((chg & 1u) != 0u), bbot = __ballot((chg & (1u << (RPW - 1))) != 0u);
            const unsigned long long bany = __ballot(chg != 0u);
            if (lx == 0) { sTop[cur][wv] = btop; sBot[cur][wv] = bbot; sAny[cur][wv] = bany != 0ull; }
            __syncthreads();   // the only barrier of a sweep: everything a wave needs from the others is read after it
            unsigned any = 0;
#pragma unroll
            for (int w = 0; w < NWAVE; w++) any |= sAny[cur][w];
            if (any) any_change = true;
            if (!any || iter + 1 >= g.max_sweeps) {
                capped = any != 0;
                if (dbg && tid == 0) { atomicAdd(dbg, (unsigned long long)(iter + 1)); atomicAdd(dbg + 1, 1ull); }
                break;
            }
            const unsigned long long above = wv > 0 ? sBot[cur][wv - 1] : 0ull, below = wv < NWAVE - 1 ? sTop[cur][wv + 1] : 0ull;
            if ((lx ? (above >> (lx - 1)) : (above << 1)) & 7ull) nd |= 1u;
            if ((lx ? (below >> (lx - 1)) : (below << 1)) & 7ull) nd |= 1u << (RPW - 1);
            dirty = nd & live;
        }
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
            if (rim) atomicOr(&sRim, rim);
            __syncthreads();
            if (tid < 8 && ((sRim >> tid) & 1)) {
                const int ddx[8] = {0, 0, -1, 1, -1, 1, -1, 1};
                const int ddy[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
                const int ntx = tx + ddx[tid], nty = ty + ddy[tid];
                if (ntx >= 0 && ntx < g.tiles_x && nty >= 0 && nty < g.tiles_y) flags_next[nty * g.tiles_x + ntx] = 1u;
            }
            if (tid == 8 && capped) flags_next[tile] = 1u;   // not yet at its fixed point: run again next round
        }
        __syncthreads();   // LDS is reused by the next listed tile
    }
}

// activation flags -> compact tile list; clears the flags; count must be zero on entry
static __global__ __launch_bounds__(256) void compact_kernel(uint32_t* __restrict__ flags, int ntiles, uint32_t* __restrict__ list,
                                                             unsigned long long* __restrict__ count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bool act = false;
    if (t < ntiles) {
        act = flags[t] != 0u;
        if (act) flags[t] = 0u;
    }
    wave_append(act, uint32_t(t), list, count);
}

static __global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Sched {
    uint32_t* flags;              // [ntiles] activation flags (input: tiles active in round 0)
    uint32_t* list;               // [ntiles]
    unsigned long long* counts;   // [2 * COUNT_RING]: per-round active counts, then per-round work cursors
};

}  // namespace tilek

// Runs rounds until no tile is active.  `flags` must hold the initially active tiles.
template <class Op>
static int tile_relax_run(tdx_context* ctx, Op op, tilek::TileGeom g, tilek::Sched sc, int64_t* rounds_out, int64_t* launches_out) {
    using namespace tilek;
    hipStream_t s = ctx->stream;
    const int ntiles = g.tiles_x * g.tiles_y;
    const unsigned grid = unsigned(std::min(ntiles, 8 * ctx->num_cus));
    const unsigned cgrid = tdx_blocks_for(size_t(ntiles), 256);
    TDX_HIP_CHECK(ctx, hipMemsetAsync(sc.counts, 0, size_t(2 * COUNT_RING) * sizeof(unsigned long long), s));
    int r = 0, batch = 4;
    int64_t rounds = 0, launches = 0;
    static const bool debug = getenv("TDX_DEBUG_ROUNDS") != nullptr;   // active tiles per round on stderr
    unsigned long long* dbg = nullptr;
    if (debug) {
        fprintf(stderr, "tile_relax_run(%d tiles):", ntiles);
        dbg = reinterpret_cast<unsigned long long*>(ctx->d_mail) + 32;
        TDX_HIP_CHECK(ctx, hipMemsetAsync(dbg, 0, 16, s));
    }
    for (;;) {
        if (r + batch > COUNT_RING) {
            TDX_HIP_CHECK(ctx, hipMemsetAsync(sc.counts, 0, size_t(2 * COUNT_RING) * sizeof(unsigned long long), s));
            r = 0;
        }
        for (int b = 0; b < batch; b++) {
            hipLaunchKernelGGL(compact_kernel, dim3(cgrid), dim3(256), 0, s, sc.flags, ntiles, sc.list, sc.counts + r + b);
            hipLaunchKernelGGL((relax_kernel<Op>), dim3(grid), dim3(NTHR), 0, s, op, g, sc.list, sc.counts + r + b, sc.flags, dbg);
        }
        launches += batch;
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, sc.counts + r, size_t(batch) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
        bool done = false;
        for (int b = 0; b < batch; b++) {
            if (ctx->h_mail[b] == 0) { done = true; break; }
            if (debug) fprintf(stderr, " %llu", (unsigned long long)ctx->h_mail[b]);
            rounds++;
        }
        if (done) break;
        r += batch;
        if (batch < 64) batch *= 2;
    }
    if (debug) {
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, dbg, 16, hipMemcpyDeviceToHost, s));
        TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
        fprintf(stderr, "\n   -> %lld rounds, %llu tile activations, %llu sweeps\n", (long long)rounds, (unsigned long long)ctx->h_mail[1],
                (unsigned long long)ctx->h_mail[0]);
    }
    if (rounds_out) *rounds_out += rounds;
    if (launches_out) *launches_out += launches;
    return TDX_OK;
}
