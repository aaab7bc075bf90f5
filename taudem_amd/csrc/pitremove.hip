// PitRemove on gfx950: replaces the compute part of flood() (src/flood.cpp:243-479).
//
// The reference iterates  W <- (Z >= m ? Z : min(W, m)),  m = min over accessible neighbours of W,
// on cells with W > Z, from the seed surface of src/flood.cpp:243-271, with a raster scan followed
// by alternating stacks of still-raised cells (src/flood.cpp:292-479).  The update uses only float
// comparisons, is monotone non-increasing from the FLT_MAX start and has a unique fixed point
// (the minimax-path surface), so ANY schedule that runs to convergence yields
// the same bits.  Schedule used here:
//
//   * pit_seed_kernel    streaming 3x3 stencil: Z -> W0                 (src/flood.cpp:243-271); with the coarse-to-fine start: Z -> first
//                        coarse level, and later Z + relaxed coarse level -> start surface
//   * tilek::relax_kernel<PitOp>  the tile relaxation engine of tile_relax.hpp: one workgroup per
//                        ACTIVE 64x64 tile, W and Z of a lane's 16-row column segment in registers
//                        (neighbour columns = neighbour lanes, LDS only between the four waves),
//                        relaxation to the tile-local fixed point, write-back of changed cells; a tile
//                        whose rim changed re-activates the neighbours it can still improve; rounds are
//                        chained on the device (no host round trip per round)
//
// HBM traffic per activation of a tile = 2 tile images in + changed cells out; rounds ~ longest fill
// path measured in tiles.
#include "context.hpp"
#include "device_common.hpp"
#include "strips.hpp"
#include "tile_relax.hpp"

namespace {

// Seed surface (src/flood.cpp:243-271).  A 256-thread block covers 64 columns x 64 rows; each lane walks a 16-row
// column segment with the 3x3 window of nodata flags in registers: 3 coalesced row loads per output row.
// MODE 0: W0 (FLT_MAX where the cell is not a seed).  The coarse-to-fine start (below) runs the pass twice instead, with nothing written in
// between: MODE 1 only reduces the seed surface to the first coarse level (per 8 x 8 block of the owned rows: Zc = largest valid
// elevation, Wc = Zc if the block holds a seed cell, else FLT_MAX; TDX_FEL_NODATA for a block without valid cells - what
// pit_coarsen_kernel derives from Z and W0), MODE 2 writes the start surface itself, W = seed ? z : Wc[block] (what pit_prolong_kernel
// makes of W0).  6.7 GB of traffic for seed + coarsen + prolong become 3 GB.
constexpr int SEED_ROWS = 16;
constexpr int SEED_CF = 8;   // = CF below
template <int MODE>
__global__ __launch_bounds__(256) void pit_seed_kernel(const float* __restrict__ Z, const int16_t* __restrict__ mask,
                                                       float* __restrict__ W, int nx, int ny, int y_own0, int y_own1, float nodata, int step,
                                                       float* __restrict__ Zc, float* __restrict__ Wc, int nxc, int nyc, int nbx, int xmap) {
    const int bx = tdxk::xcd_block_x(nbx, xmap);
    if (bx < 0) return;
    const int x = bx * 64 + (threadIdx.x & 63);
    const int ybase = y_own0 + blockIdx.y * (4 * SEED_ROWS) + (threadIdx.x >> 6) * SEED_ROWS;
    const bool colok = x < nx;
    const int xc = colok ? x : nx - 1, xm = xc > 0 ? xc - 1 : xc, xp = xc < nx - 1 ? xc + 1 : xc;
    // the lane's window: SEED_ROWS + 2 rows of its own column, every load issued before the first use (addresses clamped, validity applied afterwards -
    // a load behind a row test is a branch, and a lane would then wait for one row at a time); the columns beside it are the neighbouring lanes'
    // (one more load per row, in which only lanes 0 and 63 take part, fetches the columns beside the block); rows outside the array count as nodata
    const int lane = int(threadIdx.x & 63);
    const bool edge_lane = lane == 0 || lane == 63;
    const int xe = lane == 0 ? xm : xp;
    float zm[SEED_ROWS + 2], zedge[SEED_ROWS + 2];
#pragma unroll
    for (int j = 0; j < SEED_ROWS + 2; j++) {
        const int y = ybase - 1 + j, yc = y < 0 ? 0 : (y >= ny ? ny - 1 : y);
        const float* r = Z + size_t(yc) * size_t(nx);
        zm[j] = r[xc];
        zedge[j] = 0.f;
        if (edge_lane) zedge[j] = r[xe];
    }
    float wc[SEED_ROWS / SEED_CF];   // MODE 2: the relaxed coarse value of the blocks this lane crosses
    if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < SEED_ROWS / SEED_CF; q++) {
            const int yc = (ybase - y_own0) / SEED_CF + q, xcb = xc / SEED_CF;
            wc[q] = Wc[size_t(yc < nyc ? yc : nyc - 1) * size_t(nxc) + size_t(xcb < nxc ? xcb : nxc - 1)];
        }
    }
    unsigned nod[SEED_ROWS + 2];   // bit 0 / 1 / 2: (x - 1, x, x + 1) of the row is nodata
#pragma unroll
    for (int j = 0; j < SEED_ROWS + 2; j++) {
        const int y = ybase - 1 + j;
        const bool in = y >= 0 && y < ny;
        const int c = tdxk::is_nodata_f(zm[j], nodata) ? 1 : 0, e = tdxk::is_nodata_f(zedge[j], nodata) ? 1 : 0;
        const int l = tilek::lane_left(c, 0), r = tilek::lane_right(c, 0);
        nod[j] = in ? (unsigned(lane == 0 ? e : l) | (unsigned(c) << 1) | (unsigned(lane == 63 ? e : r) << 2)) : 7u;
    }
    float bmax[SEED_ROWS / SEED_CF];   // MODE 1: this lane's column of each block row it crosses
    bool bany[SEED_ROWS / SEED_CF], bseed[SEED_ROWS / SEED_CF];
#pragma unroll
    for (int q = 0; q < SEED_ROWS / SEED_CF; q++) { bmax[q] = -FLT_MAX; bany[q] = false; bseed[q] = false; }
#pragma unroll
    for (int r = 0; r < SEED_ROWS; r++) {
        const int y = ybase + r;
        const float zc = zm[r + 1];
        const unsigned nn = nod[r], nc = nod[r + 1], ns = nod[r + 2];
        const bool c1 = (nc & 2u) != 0u;
        if (colok && y < y_own1) {
            const size_t idx = size_t(y) * size_t(nx) + size_t(x);
            float w;
            if (c1) w = TDX_FEL_NODATA;
            else if (mask && mask[idx] == 1) w = zc;
            else if (x == 0 || y == 0 || x == nx - 1 || y == ny - 1) w = zc;   // !hasAccess(i+-1,j+-1): global edge ring
            else {
                const bool con = (step == 2) ? (((nc & 5u) | (nn & 2u) | (ns & 2u)) != 0u) : ((nn | nc | ns) != 0u);   // (nc & 2 is clear here)
                w = con ? zc : FLT_MAX;
            }
            if (MODE == 0) W[idx] = w;
            if (MODE == 1 && !c1) {
                bany[r / SEED_CF] = true;
                bmax[r / SEED_CF] = fmaxf(bmax[r / SEED_CF], zc);
                if (w != FLT_MAX) bseed[r / SEED_CF] = true;
            }
            if (MODE == 2) {
                if (w == FLT_MAX) w = wc[r / SEED_CF];
                W[idx] = w;
            }
        }
    }
    if (MODE == 1) {   // the 8 lanes of a block column (aligned: 64 columns per wave), then one store per block
#pragma unroll
        for (int q = 0; q < SEED_ROWS / SEED_CF; q++) {
            float m = bmax[q];
            int fl = (bany[q] ? 1 : 0) | (bseed[q] ? 2 : 0);
#pragma unroll
            for (int off = 1; off < SEED_CF; off <<= 1) {
                m = fmaxf(m, __shfl_xor(m, off, 64));
                fl |= __shfl_xor(fl, off, 64);
            }
            const int yc = (ybase - y_own0) / SEED_CF + q, xcb = x / SEED_CF;
            if ((threadIdx.x & (SEED_CF - 1)) == 0 && xcb < nxc && yc < nyc) {
                const size_t c = size_t(yc) * size_t(nxc) + size_t(xcb);
                Zc[c] = (fl & 1) ? m : TDX_FEL_NODATA;
                Wc[c] = (fl & 1) ? ((fl & 2) ? m : FLT_MAX) : TDX_FEL_NODATA;
            }
        }
    }
}

// ---- coarse-to-fine start surface ------------------------------------------------------------------------------
// Relaxing from the reference's +inf start lowers every cell many times (each newly found saddle re-lowers a whole
// lake).  Any start surface W0 >= W* converges to the same fixed point, so the relaxation is started from a TIGHT
// upper bound obtained on a 8x coarser raster: Zc = max of the valid cells of an 8x8 block; a block is a seed block if
// it holds a seed cell.  A coarse minimax path maps to a fine path that stays inside those blocks (a block's valid
// cells are 8-connected to each other or to a seed next to the block's nodata cells), hence W*(c) <= Wc*(block(c)).
// Used for the 8-neighbour fill of a single strip; recursive.
constexpr int CF = 8;
static_assert(CF == SEED_CF, "pit_seed_kernel reduces to the first coarse level");
__global__ __launch_bounds__(256) void pit_coarsen_kernel(const float* __restrict__ Z, const float* __restrict__ W0, int nx, int ny, int nxc, int nyc,
                                                          float* __restrict__ Zc, float* __restrict__ Wc) {
    const int xc = blockIdx.x * 64 + (threadIdx.x & 63), yc = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xc >= nxc || yc >= nyc) return;
    float zmax = -FLT_MAX;
    bool seed = false, any = false;
    auto cell = [&](float w, float z) {
        if (w == TDX_FEL_NODATA) return;              // nodata cell of the fine raster
        any = true;
        zmax = fmaxf(zmax, z);
        if (w != FLT_MAX) seed = true;                // fine seed: W0 == Z
    };
    if ((nx & 3) == 0 && xc * CF + CF <= nx && yc * CF + CF <= ny) {   // whole block, 16-byte aligned rows: two 128-bit loads per row and array
        for (int j = 0; j < CF; j++) {
            const size_t o = size_t(yc * CF + j) * size_t(nx) + size_t(xc * CF);
            const float4 w0 = *reinterpret_cast<const float4*>(W0 + o), w1 = *reinterpret_cast<const float4*>(W0 + o + 4);
            const float4 z0 = *reinterpret_cast<const float4*>(Z + o), z1 = *reinterpret_cast<const float4*>(Z + o + 4);
            cell(w0.x, z0.x); cell(w0.y, z0.y); cell(w0.z, z0.z); cell(w0.w, z0.w);
            cell(w1.x, z1.x); cell(w1.y, z1.y); cell(w1.z, z1.z); cell(w1.w, z1.w);
        }
    } else {
        for (int j = 0; j < CF; j++) {
            const int y = yc * CF + j;
            if (y >= ny) break;
            for (int i = 0; i < CF; i++) {
                const int x = xc * CF + i;
                if (x >= nx) break;
                cell(W0[size_t(y) * size_t(nx) + size_t(x)], Z[size_t(y) * size_t(nx) + size_t(x)]);
            }
        }
    }
    const size_t c = size_t(yc) * size_t(nxc) + size_t(xc);
    // a block without valid cells behaves like a nodata cell: never updated; its neighbours are seed blocks anyway
    Zc[c] = any ? zmax : TDX_FEL_NODATA;
    Wc[c] = any ? (seed ? zmax : FLT_MAX) : TDX_FEL_NODATA;
}
__global__ __launch_bounds__(256) void pit_prolong_kernel(float* __restrict__ W, int nx, int ny, const float* __restrict__ Wc, int nxc) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    if (W[idx] == FLT_MAX) W[idx] = Wc[size_t(y / CF) * size_t(nxc) + size_t(x / CF)];
}

// ---- one coarse correction in the middle of the fine relaxation ("V-cycle") --------------------------------------------------------------
// The coarse start is an UPPER bound: a lake stands at its block-level spill, which is the largest elevation of the spill's 8 x 8 block, not the spill
// itself.  The fine rounds find the true spill at once - and then lower the whole lake by the difference at one tile per round: the long tail of
// PitRemove (~130 rounds of 16 us at 16384^2) and, across strips, one strip-local fixed point per strip the lake covers.  But what the fine level has
// found can be handed back: U(B) = max of the current fine W over block B bounds the true surface on B, min(Wc, U) is again an upper bound of the
// per-block maxima, the coarse operator keeps that property (a path inside a block never climbs above the block's largest elevation), so the coarse
// level - relaxed again, across the strips, at 1/8 of the rounds - spreads the lowered levels over the lakes, and W <- min(W, Wc[block]) brings them
// back (never below Z: Wc[B] >= the largest elevation of B).  Any upper bound converges to the same bits; only the number of rounds changes.
// Ur keeps what the block was restricted to (= the block's largest fine W: the start surface never exceeds the coarse one): the coarse relaxation that
// follows has work for the fine level exactly where it takes a block BELOW that (pit_prolong_min_kernel); the coarse tiles that see a lowered block are
// flagged (ya0c = array row of the coarse level's first owned row).
__global__ __launch_bounds__(256) void pit_restrict_kernel(const float* __restrict__ W, int nx, int ny, int nxc, int nyc, float* __restrict__ Wc, float* __restrict__ Ur,
                                                           int ya0c, int ny_arr_c, int tiles_x_c, uint32_t* __restrict__ cflags) {
    const int xc = blockIdx.x * 64 + (threadIdx.x & 63), yc = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xc >= nxc || yc >= nyc) return;
    const size_t c = size_t(yc) * size_t(nxc) + size_t(xc);
    const float wc = Wc[c];
    Ur[c] = wc;
    if (wc == TDX_FEL_NODATA) return;   // a block without valid cells
    float u = -FLT_MAX;
    if ((nx & 3) == 0 && xc * CF + CF <= nx && yc * CF + CF <= ny) {
        for (int j = 0; j < CF; j++) {
            const size_t o = size_t(yc * CF + j) * size_t(nx) + size_t(xc * CF);
            const float4 w0 = *reinterpret_cast<const float4*>(W + o), w1 = *reinterpret_cast<const float4*>(W + o + 4);
            const float v[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int i = 0; i < 8; i++) if (v[i] != TDX_FEL_NODATA) u = fmaxf(u, v[i]);
        }
    } else {
        for (int j = 0; j < CF && yc * CF + j < ny; j++)
            for (int i = 0; i < CF && xc * CF + i < nx; i++) {
                const float v = W[size_t(yc * CF + j) * size_t(nx) + size_t(xc * CF + i)];
                if (v != TDX_FEL_NODATA) u = fmaxf(u, v);
            }
    }
    if (u < wc) {
        Wc[c] = u; Ur[c] = u;
        tilek::activate_tiles_around(xc, ya0c + yc, nxc, ny_arr_c, tiles_x_c, cflags);
    }
}
// W <- min(W, Wc[block]) wherever the coarse relaxation took a block below what it was restricted to; the fine tiles that see a lowered cell are activated
// (ya0 = array row of the fine level's first owned row).  One thread per COLUMN of a coarse block (8 cells; lanes = consecutive columns: coalesced rows) - the first
// version, one thread per block walking its 64 cells and raising nine flags per lowered cell, took 0.57 ms at 16384^2, more than the coarse relaxation it follows.
__global__ __launch_bounds__(256) void pit_prolong_min_kernel(float* __restrict__ W, int nx, int nyo, const float* __restrict__ Wc, const float* __restrict__ Ur, int nxc, int nyc,
                                                              int ya0, int ny_arr, int tiles_x, uint32_t* __restrict__ tile_flags) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), yc = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || yc >= nyc) return;
    const size_t c = size_t(yc) * size_t(nxc) + size_t(x / CF);
    const float wc = Wc[c];
    if (wc == TDX_FEL_NODATA || !(wc < Ur[c])) return;
    const int y0 = yc * CF, rows = nyo - y0 < CF ? nyo - y0 : CF;
    float w[CF];
#pragma unroll
    for (int j = 0; j < CF; j++) w[j] = W[size_t(y0 + (j < rows ? j : rows - 1)) * size_t(nx) + size_t(x)];
    bool any = false;
#pragma unroll
    for (int j = 0; j < CF; j++)
        if (j < rows && w[j] != TDX_FEL_NODATA && wc < w[j]) { W[size_t(y0 + j) * size_t(nx) + size_t(x)] = wc; any = true; }
    if (any) {   // the tiles around the column's first and last cell: a superset of the tiles around the cells that moved (a block's rows lie in one tile)
        tilek::activate_tiles_around(x, ya0 + y0, nx, ny_arr, tiles_x, tile_flags);
        tilek::activate_tiles_around(x, ya0 + y0 + rows - 1, nx, ny_arr, tiles_x, tile_flags);
    }
}

// minimax-path operator of flood() (src/flood.cpp:295-330): W <- (Z >= m ? Z : min(W, m)) where W > Z
template <int NBR>   // 8, or 4 for the -4way flag (k = 1,3,5,7)
struct PitOp {
    using T = float;
    static constexpr int kUniform = NBR;
    static constexpr int kWaves = 5;   // 96 VGPRs without spills: five tiles per CU (the bulk rounds stream tiles, one more in flight is worth 1 %)
    const float* Z;
    float* W;
    static __device__ __forceinline__ float inf() { return FLT_MAX; }
    using Raw = float;
    using CellRaw = float;
    __device__ __forceinline__ float load_raw(size_t idx) const { return W[idx]; }
    static __device__ __forceinline__ float decode(float v) { return v; }
    __device__ __forceinline__ void store(size_t idx, float v) const { W[idx] = v; }
    __device__ __forceinline__ float cell_raw(size_t idx) const { return Z[idx]; }
    static __device__ __forceinline__ void cell_decode(float z, float& cst, unsigned& mask) { cst = z; mask = (NBR == 4) ? 0x55u : 0xFFu; }
    // A cell that is settled when the tile is loaded (w <= z: seeds, nodata cells with their marker value) never moves: its floor becomes
    // its own value.  For all others w > z >= ... holds for the whole activation (values only decrease, never below z), so
    // (w > z ? max(z, min(w, m)) : w) is the median of (z, w, m): ONE v_med3_f32 per cell and sweep.
    static __device__ __forceinline__ float cell_floor(float z, float w) { return (w > z) ? z : w; }
    static __device__ __forceinline__ float apply(float z, float w, float m) { return tilek::med3_raw(z, w, m); }
    static __device__ __forceinline__ bool settled(float z, float w) { return !(w > z); }
    // activation filter (tile_relax.hpp): a halo cell that still stands above its elevation comes down for a neighbour value below its own
    static __device__ __forceinline__ float act_never() { return -FLT_MAX; }
    static __device__ __forceinline__ float act_threshold(float z, float w) { return (w > z) ? w : act_never(); }
};

}  // namespace

static int pit_finish_level(tdx_context* ctx, const Strip& ls, float* Zc, float* Wc, int nxc, int nyc, int64_t cells_all, tilek::Sched sc, int depth,
                            int64_t* rounds, int64_t* launches, int64_t* own_rounds = nullptr);

// The strip of a coarse level: the rows the rank's owned rows coarsen to, stacked in rank order, ARE a coarsening of the whole raster (a rank's last
// block row may be lower than CF rows; blocks that touch in the fine raster are neighbours in the stacked coarse raster and vice versa), so a
// coarse level of a multi-strip run is a row-partitioned raster like the fine one: one halo row each side, the neighbours' boundary rows exchanged.
// A single strip keeps arrays without halo rows (cells outside the array are inaccessible, like the raster's edge).
static inline Strip pit_level_strip(const Strip& st, int nxc, int nyc) {
    Strip ls;
    ls.nx = nxc; ls.comm = st.comm;
    if (st.multi()) { ls.ny_arr = nyc + 2; ls.y0 = 1; ls.y1 = nyc + 1; ls.up = st.up; ls.down = st.down; }
    else { ls.ny_arr = nyc; ls.y0 = 0; ls.y1 = nyc; }
    return ls;
}

// One level (fine or coarse) to its fixed point.  Multi-strip: every rank relaxes its strip to the local fixed point with the neighbours' boundary
// rows frozen in its halo rows, then boundary rows are exchanged and the tiles that see a changed halo cell are re-activated; repeat until no halo
// cell changed on any rank (the roles of share() + ringTerm() in src/flood.cpp:344-355,457-468).
template <int NBR>
static int pit_relax_level(tdx_context* ctx, const Strip& ls, const float* Z, float* W, tilek::Sched sc, int64_t* rounds, int64_t* launches, int64_t* outer,
                           bool all_tiles = true) {
    const tilek::TileGeom g = tilek::make_geom(ls.nx, ls.ny_arr, ls.y0, ls.y1);
    for (;;) {
        // round 0: every tile is active (one set-up launch: tilek::all_start_kernel), otherwise / later: the tiles flagged in sc.flags
        int rc = tile_relax_run(ctx, PitOp<NBR>{Z, W}, g, sc, rounds, launches, all_tiles);
        all_tiles = false;
        if (rc != TDX_OK) return rc;
        if (outer) (*outer)++;
        if (!ls.multi()) break;
        int64_t changed = 0;
        rc = strip_exchange<float>(ctx, ls, W, TDX_FEL_NODATA, sc.flags, g.tiles_x, &changed, true);   // halo exchange + the termination vote in one step
        if (rc != TDX_OK) return rc;
        if (changed == 0) break;
    }
    return TDX_OK;
}

// Tightens the start surface of level `depth` (Z, W: its owned rows; 8-neighbour fill) through coarser levels: coarsen, recurse, relax the coarse
// level ACROSS THE STRIPS, prolong.  cells_all = cells of this level in all strips (every rank passes the same number, so every rank builds the
// same levels).  In a multi-strip run the coarse levels are what keeps the fine relaxation local: a strip that relaxes on its own from the seed
// surface can only drain through the raster's edges it owns - an inner strip fills up to its lowest pass to the left / right edge, and the true
// levels then arrive one strip per exchange, each time re-lowering most of the strip (profiles/r05a_*: 104 ms on the critical path of eight
// 65536 x 8192 strips against 8.5 ms for a lone strip).  The coarsest level is a few tiles per rank; its exchanges cost microseconds.
static int pit_coarse_start(tdx_context* ctx, const Strip& st, const float* Z, float* W, int nx, int nyo, int64_t cells_all, tilek::Sched sc, int depth,
                            int64_t* rounds, int64_t* launches) {
    if (depth >= 3 || cells_all < (int64_t(1) << 18) || nyo < 1) return TDX_OK;
    hipStream_t s = ctx->stream;
    const int nxc = (nx + CF - 1) / CF, nyc = (nyo + CF - 1) / CF;
    const Strip ls = pit_level_strip(st, nxc, nyc);
    const size_t nc = size_t(nxc) * size_t(ls.ny_arr), off = size_t(ls.y0) * size_t(nxc);
    float* Zc = static_cast<float*>(ctx->scratch(TDX_S_D + 2 * depth, nc * 4));
    float* Wc = static_cast<float*>(ctx->scratch(TDX_S_E + 2 * depth, nc * 4));
    if (!Zc || !Wc) return TDX_ERR_NOMEM;
    const dim3 gc((nxc + 63) / 64, (nyc + 3) / 4);
    hipLaunchKernelGGL(pit_coarsen_kernel, gc, dim3(256), 0, s, Z, W, nx, nyo, nxc, nyc, Zc + off, Wc + off);
    int rc = pit_finish_level(ctx, ls, Zc, Wc, nxc, nyc, cells_all / (CF * CF), sc, depth, rounds, launches);
    if (rc != TDX_OK) return rc;
    const dim3 gf((nx + 63) / 64, (nyo + 3) / 4);
    hipLaunchKernelGGL(pit_prolong_kernel, gf, dim3(256), 0, s, W, nx, nyo, Wc + off, nxc);
    return TDX_OK;
}
// a coarse level whose owned rows hold the coarsened seed surface: halo rows, coarser levels, relaxation
static int pit_finish_level(tdx_context* ctx, const Strip& ls, float* Zc, float* Wc, int nxc, int nyc, int64_t cells_all, tilek::Sched sc, int depth,
                            int64_t* rounds, int64_t* launches, int64_t* own_rounds) {
    const size_t off = size_t(ls.y0) * size_t(nxc);
    int rc = pit_coarse_start(ctx, ls, Zc + off, Wc + off, nxc, nyc, cells_all, sc, depth + 1, rounds, launches);
    if (rc != TDX_OK) return rc;
    rc = strip_exchange<float>(ctx, ls, Zc, TDX_FEL_NODATA);   // (a block without valid cells is a nodata cell of its level, and so is everything beyond the raster)
    if (rc != TDX_OK) return rc;
    rc = strip_exchange<float>(ctx, ls, Wc, TDX_FEL_NODATA);
    if (rc != TDX_OK) return rc;
    const int64_t before = *rounds;
    rc = pit_relax_level<8>(ctx, ls, Zc, Wc, sc, rounds, launches, nullptr);
    if (own_rounds) *own_rounds = *rounds - before;   // rounds of THIS level's relaxation: ~1/6 of what the next finer level's tail will need
    return rc;
}

// One strip (src/flood.cpp:132-482).
static int pitremove_impl(tdx_context* ctx, const Strip& st, float* d_dem, const int16_t* d_mask, int fourway, float dem_nodata, float* d_fel,
                          tdx_stats* stats) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const tilek::TileGeom geom = tilek::make_geom(st.nx, st.ny_arr, st.y0, st.y1);
    const int ntiles = geom.tiles_x * geom.tiles_y;
    uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_A, size_t(ntiles) * 4));
    uint32_t* list = static_cast<uint32_t*>(ctx->scratch(TDX_S_B, size_t(ntiles) * 4 * tilek::SCHED_LIST_WORDS));
    unsigned long long* counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_C, size_t(tilek::COUNT_RING) * 16));
    if (!flags || !list || !counts) return TDX_ERR_NOMEM;
    const tilek::Sched sched{flags, list, counts};

    ctx->begin_call(stats);
    strip_mark(ctx, st, "pitremove");
    int rc = strip_exchange<float>(ctx, st, d_dem, dem_nodata);   // elevation halo rows
    if (rc != TDX_OK) return rc;
    int64_t rounds = 0, launches = 0, outer = 0;
    const int sbx = (st.nx + 63) / 64, sxmap = 0;   // (64-column blocks on line boundaries: the XCD-aware order of d8flowdir.hip gains nothing here)
    const dim3 sgrid(unsigned(sbx), (st.y1 - st.y0 + 4 * SEED_ROWS - 1) / (4 * SEED_ROWS));
    const bool no_coarse = getenv("TDX_PIT_NO_COARSE") != nullptr;   // (test hook: read per call)
    const int nyo = st.y1 - st.y0;
    int64_t cells_all = int64_t(st.nx) * int64_t(nyo);   // every rank takes the same path: the decision is made on the whole raster's size
    rc = strip_allreduce(ctx, st, &cells_all, 1, TDX_OP_SUM);
    if (rc != TDX_OK) return rc;
    const bool used_coarse = !fourway && !no_coarse && cells_all >= (int64_t(1) << 18);
    int64_t level1_rounds = 0;
    if (used_coarse) {
        // seed surface -> first coarse level -> (coarser levels, relaxed coarse to fine, each across the strips) -> start surface, without a W0 in between
        const int nxc = (st.nx + CF - 1) / CF, nyc = (nyo + CF - 1) / CF;
        const Strip ls = pit_level_strip(st, nxc, nyc);
        const size_t nc = size_t(nxc) * size_t(ls.ny_arr), off = size_t(ls.y0) * size_t(nxc);
        float* Zc = static_cast<float*>(ctx->scratch(TDX_S_D, nc * 4));
        float* Wc = static_cast<float*>(ctx->scratch(TDX_S_E, nc * 4));
        if (!Zc || !Wc) return TDX_ERR_NOMEM;
        {
            TdxSpan sp(ctx, TDX_K_STENCIL);
            hipLaunchKernelGGL(pit_seed_kernel<1>, sgrid, dim3(256), 0, s, d_dem, d_mask, d_fel, st.nx, st.ny_arr, st.y0, st.y1, dem_nodata, 1, Zc + off, Wc + off, nxc, nyc, sbx, sxmap);
            if (stats) stats->launches[TDX_K_STENCIL]++;
        }
        {
            TdxSpan sp(ctx, TDX_K_RELAX);
            ctx->phase = "coarse levels";
            rc = pit_finish_level(ctx, ls, Zc, Wc, nxc, nyc, cells_all / (CF * CF), sched, 0, &rounds, &launches, &level1_rounds);
            if (rc != TDX_OK) return rc;
        }
        {
            TdxSpan sp(ctx, TDX_K_STENCIL);
            hipLaunchKernelGGL(pit_seed_kernel<2>, sgrid, dim3(256), 0, s, d_dem, d_mask, d_fel, st.nx, st.ny_arr, st.y0, st.y1, dem_nodata, 1, Zc + off, Wc + off, nxc, nyc, sbx, sxmap);
            if (stats) stats->launches[TDX_K_STENCIL]++;
        }
    } else {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        hipLaunchKernelGGL(pit_seed_kernel<0>, sgrid, dim3(256), 0, s, d_dem, d_mask, d_fel, st.nx, st.ny_arr, st.y0, st.y1, dem_nodata, fourway ? 2 : 1,
                           static_cast<float*>(nullptr), static_cast<float*>(nullptr), 0, 0, sbx, sxmap);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    ctx->phase = "fine level";
    rc = strip_exchange<float>(ctx, st, d_fel, TDX_FEL_NODATA);   // start surface halo rows
    if (rc != TDX_OK) return rc;
    {
        TdxSpan sp(ctx, TDX_K_RELAX);
        bool all_tiles = true;
        // one coarse correction after the first fine rounds (see pit_restrict_kernel); TDX_PIT_VCYCLE_AFTER=n: after n rounds (0: never)
        const int vc_env = getenv("TDX_PIT_VCYCLE_AFTER") ? std::max(0, atoi(getenv("TDX_PIT_VCYCLE_AFTER"))) : 8;   // (read per call: test hook)
        if (used_coarse && vc_env > 0) {
            bool left = false;
            rc = tile_relax_run_bounded(ctx, PitOp<8>{d_dem, d_fel}, geom, sched, vc_env, &left, &rounds, &launches, true);   // (round 0: every tile)
            if (rc != TDX_OK) return rc;
            all_tiles = false;   // what is still active stays flagged
            // A correction costs a pass over the fine surface, a coarse relaxation and a restart of the fine schedule (~1 ms at 16384^2): it pays when the fine
            // tail is long.  The first coarse level's own relaxation says how long - the fine level needs ~6 x its rounds (13 -> 67 on a 65536 x 8192 strip,
            // 20 -> 123 at 16384^2, 36 -> 245 at 32768^2) -: corrected from 16 coarse rounds on; TDX_PIT_VCYCLE_MIN=n moves the bar.  With neighbours every
            // rank must take the same path, and the strip-by-strip lowering of a shared lake is what the correction is for: always.
            const int vc_min = getenv("TDX_PIT_VCYCLE_MIN") ? atoi(getenv("TDX_PIT_VCYCLE_MIN")) : 16;   // (read per call: test hook)
            if (st.multi() || (left && level1_rounds >= vc_min)) {
                const int nxc = (st.nx + CF - 1) / CF, nyc = (nyo + CF - 1) / CF;
                const Strip ls = pit_level_strip(st, nxc, nyc);
                const size_t off = size_t(ls.y0) * size_t(nxc);
                float* Zc = static_cast<float*>(ctx->scratch(TDX_S_D, size_t(nxc) * size_t(ls.ny_arr) * 4));
                float* Wc = static_cast<float*>(ctx->scratch(TDX_S_E, size_t(nxc) * size_t(ls.ny_arr) * 4));
                const tilek::TileGeom gc = tilek::make_geom(ls.nx, ls.ny_arr, ls.y0, ls.y1);
                const size_t ntc = size_t(gc.tiles_x) * size_t(gc.tiles_y);
                uint32_t* cflags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntc * 4 * (1 + tilek::SCHED_LIST_WORDS)));   // (the fine level's flags are in use)
                unsigned long long* ccounts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
                float* Ur = static_cast<float*>(ctx->scratch(TDX_S_N, size_t(nxc) * size_t(nyc) * 4));
                if (!Zc || !Wc || !cflags || !ccounts || !Ur) return TDX_ERR_NOMEM;
                ctx->phase = "coarse correction";
                TDX_HIP_CHECK(ctx, hipMemsetAsync(cflags, 0, ntc * 4, s));
                hipLaunchKernelGGL(pit_restrict_kernel, dim3((nxc + 63) / 64, (nyc + 3) / 4), dim3(256), 0, s, d_fel + size_t(st.y0) * size_t(st.nx), st.nx, nyo, nxc, nyc, Wc + off, Ur,
                                   ls.y0, ls.ny_arr, gc.tiles_x, cflags);
                {   // the neighbours' restricted boundary rows (the coarse tiles that see a change are flagged by the merge)
                    int64_t changed = 0;
                    rc = strip_exchange<float>(ctx, ls, Wc, TDX_FEL_NODATA, cflags, gc.tiles_x, &changed);
                    if (rc != TDX_OK) return rc;
                }
                rc = pit_relax_level<8>(ctx, ls, Zc, Wc, tilek::Sched{cflags, cflags + ntc, ccounts}, &rounds, &launches, nullptr, false);
                if (rc != TDX_OK) return rc;
                hipLaunchKernelGGL(pit_prolong_min_kernel, dim3((st.nx + 63) / 64, (nyc + 3) / 4), dim3(256), 0, s, d_fel + size_t(st.y0) * size_t(st.nx), st.nx, nyo, Wc + off, Ur, nxc, nyc,
                                   st.y0, st.ny_arr, geom.tiles_x, flags);
                ctx->phase = "fine level";
                if (st.multi()) {   // the neighbours' lowered boundary rows (their tiles are flagged by the merge)
                    int64_t changed = 0;
                    rc = strip_exchange<float>(ctx, st, d_fel, TDX_FEL_NODATA, flags, geom.tiles_x, &changed);
                    if (rc != TDX_OK) return rc;
                }
            }
        }
        rc = fourway ? pit_relax_level<4>(ctx, st, d_dem, d_fel, sched, &rounds, &launches, &outer, all_tiles) : pit_relax_level<8>(ctx, st, d_dem, d_fel, sched, &rounds, &launches, &outer, all_tiles);
        if (rc != TDX_OK) return rc;
        if (stats) stats->launches[TDX_K_RELAX] += launches;
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    if (stats) { stats->rounds = rounds; stats->cells_evaluated = outer; }
    ctx->end_call();
    return TDX_OK;
}

extern "C" int tdx_pitremove_dev(tdx_context* ctx, const float* d_dem, int64_t nx, int64_t ny, float dem_nodata,
                                 const int16_t* d_mask, int fourway, float* d_fel, tdx_stats* stats) {
    if (!ctx || !d_dem || !d_fel || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_pitremove_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return pitremove_impl(ctx, strip_single(int(nx), int(ny)), const_cast<float*>(d_dem), d_mask, fourway, dem_nodata, d_fel, stats);
}

extern "C" int tdx_pitremove_strip(tdx_context* ctx, const tdx_comm* comm, float* d_dem, int64_t nx, int64_t ny_local, float dem_nodata,
                                   const int16_t* d_mask, int fourway, float* d_fel, tdx_stats* stats) {
    if (!ctx || !d_dem || !d_fel || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_pitremove_strip: bad argument");
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return pitremove_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_dem, d_mask, fourway, dem_nodata, d_fel, stats);
}

extern "C" int tdx_pitremove(tdx_context* ctx, const float* dem, int64_t nx, int64_t ny, float dem_nodata,
                             const int16_t* mask, int fourway, float* fel, tdx_stats* stats) {
    if (!ctx || !dem || !fel || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_pitremove: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_z = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_w = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    int16_t* d_m = mask ? static_cast<int16_t*>(ctx->scratch(TDX_S_IO2, n * 2)) : nullptr;
    if (!d_z || !d_w || (mask && !d_m)) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_z, dem, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (mask) TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_m, mask, n * 2, hipMemcpyHostToDevice, ctx->stream));
    int rc = tdx_pitremove_dev(ctx, d_z, nx, ny, dem_nodata, d_m, fourway, d_w, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(fel, d_w, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
