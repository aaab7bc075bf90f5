// AreaD8 on gfx950: replaces the compute part of aread8() (src/aread8.cpp:192-304) and
// initNeighborD8up() (src/commonLib.cpp:240-386).
//
// The reference is a Kahn sweep: a FIFO of cells whose contributing neighbours are all final; on pop
// the cell is evaluated by PULLING its contributors in k = 1..8 order with float32 adds
// (src/aread8.cpp:231-256) and the in-degree of its downstream cell is decremented
// (src/aread8.cpp:261-272).  A cell's value depends only on its inputs, never on queue order, so
// the schedule is free; the k-ordered float32 pull is kept because sums above 2^24 round.
//
//   ad8_setup_kernel   streaming 3x3 stencil over p: in-degree per cell (with the reference's range
//                      checks, incl. the "p == 0 counts at k == 4" quirk), result pre-set to -1
//   ad8_walk_kernel    one lane per cell; lanes on a ready cell (in-degree 0) evaluate it, publish the
//                      value with an agent-scope (sc1) store, drain, decrement the downstream counter
//                      with a device-scope atomic and CONTINUE into the downstream cell iff they were
//                      its last contributor - no queue, no global barrier; critical path = longest
//                      flow path x one atomic round trip
//   outlets            reverse BFS from the outlet cells marks the upstream closure (frontier sweeps)
#include "context.hpp"
#include "device_common.hpp"
#include "d8_sweep.hpp"
#include "flats.hpp"
#include "strips.hpp"

#include <cstdlib>

namespace {
using namespace tdxk;

constexpr int32_t CNT_NOT_PART = 0x40000000;   // never reaches 0: the reference's int16 counter wraps instead (src/aread8.cpp:266-268)
// A cell with no contributor is a SOURCE.  It must stay distinguishable from a cell whose counter was
// driven to 0 by its contributors (that cell is evaluated by its last contributor's lane, and its own
// lane may start later and must not evaluate it again), so sources carry a value no decrement produces.
constexpr int32_t CNT_SOURCE = -1;
constexpr int32_t CNT_DONE = -2;        // evaluated: what a strip neighbour looks for in the exchanged boundary rows

// Outlets mode runs the ordinary sweep on a re-coded direction grid P' (d8_apply_reach_kernel): cells outside the
// upstream closure of the outlets keep "a valid direction" for the contamination test but neither participate nor
// contribute (codes 16..24 = 16 + p); an outlet placed on a cell without direction participates as a pure sink
// (code 32) - the reference evaluates such a cell from the neighbours that drain into it (src/commonLib.cpp:285-359).
constexpr int16_t P_OUTSIDE = 16, P_SINK = 32;

// The per-cell expression of the D8 dependency sweep.  SUM: AreaD8 (src/aread8.cpp:231-256).  MAX / MIN: D8FlowPathExtremeUp
// (src/D8flowpathextremeup.cpp:167-199): the cell's own value of the input grid, then max / min with every contributor.
enum { D8X_SUM = 0, D8X_MAX = 1, D8X_MIN = 2 };
struct D8Expr { int mode; float out_nodata; };
__device__ __forceinline__ bool d8_participates(int16_t p, int16_t nodata) { return !is_nodata_s(p, nodata) && ((p >= 0 && p <= 8) || p == P_SINK); }

// in-degree as in initNeighborD8up (src/commonLib.cpp:251-282)
__device__ __forceinline__ int d8_indegree(const int16_t* __restrict__ P, int nx, int ny, int x, int y, int16_t nodata) {
    int cnt = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        const int xn = x + d1(k), yn = y + d2(k);
        if (xn >= 0 && xn < nx && yn >= 0 && yn < ny) {
            const int16_t pn = P[size_t(yn) * size_t(nx) + size_t(xn)];
            if (!is_nodata_s(pn, nodata) && pn >= 0 && pn <= 8 && (pn - k == 4 || pn - k == -4)) cnt++;
        }
    }
    return cnt;
}

__global__ __launch_bounds__(256) void ad8_setup_kernel(const int16_t* __restrict__ P, int nx, int ny, int y_own0, int y_own1, int16_t nodata,
                                                        int32_t* __restrict__ cnt, float* __restrict__ A, float out_nodata) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = y_own0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= y_own1) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    const int16_t p = P[idx];
    int32_t c = CNT_NOT_PART;
    if (d8_participates(p, nodata)) {
        c = d8_indegree(P, nx, ny, x, y, nodata);
        if (c == 0) c = CNT_SOURCE;
    }
    cnt[idx] = c;
    A[idx] = out_nodata;
}

// ---- outlets: upstream closure through the tile relaxation engine (flats.hpp: reach_closure) ----
// mask of the reachability relaxation: the neighbour a cell drains to; a p == 0 cell counts as draining to its
// south-east neighbour, because initNeighborD8up counts it in that neighbour's in-degree (src/commonLib.cpp:257-266)
__global__ __launch_bounds__(256) void d8_reach_mask_kernel(const int16_t* __restrict__ P, size_t n, int16_t nodata, uint8_t* __restrict__ mask) {
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
// outlet cells (array coordinates; only those in the owned rows): reach = 1 and their tile is activated
__global__ __launch_bounds__(256) void reach_seed_kernel(const int32_t* __restrict__ ox, const int32_t* __restrict__ oy, int nout, int nx, int ny, int y_own0,
                                                         int y_own1, int tiles_x, int32_t* __restrict__ reach, uint32_t* __restrict__ tile_flags) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    const int x = ox[o], y = oy[o];
    if (x < 0 || x >= nx || y < y_own0 || y >= y_own1) return;   // globalToLocal + isInPartition (src/commonLib.cpp:289-291)
    reach[size_t(y) * size_t(nx) + size_t(x)] = 1;
    tilek::activate_tiles_around(x, y, nx, ny, tiles_x, tile_flags);
}
__global__ __launch_bounds__(256) void d8_apply_reach_kernel(const int16_t* __restrict__ P, const int32_t* __restrict__ reach, size_t n, int16_t nodata,
                                                             int16_t* __restrict__ Pout) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const int16_t p = P[i];
    const bool valid = !is_nodata_s(p, nodata) && p >= 0 && p <= 8;
    int16_t q;
    if (reach[i] == 1) q = valid ? p : P_SINK;
    else q = valid ? int16_t(p + P_OUTSIDE) : p;
    Pout[i] = q;
}

// cells that take part in the sweep (d8_participates): the bound of every exact count (aread8_impl).  A few thousand blocks stride over the rows and end
// in ONE atomic each (a word takes ~90 M atomics/s: one atomic per wave was 12.7 ms of a 0.3 ms pass at 65536 x 8192, profiles/r05b_*).
__global__ __launch_bounds__(256) void d8_count_participating_kernel(const int16_t* __restrict__ P, size_t n, int16_t nodata, unsigned long long* __restrict__ total) {
    unsigned c = 0;
    for (size_t i0 = (size_t(blockIdx.x) * 256 + threadIdx.x) * 8; i0 < n; i0 += size_t(gridDim.x) * 2048) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (i0 + j < n && d8_participates(P[i0 + j], nodata)) c++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}

__global__ void fill_i32_kernel(int32_t* p, int32_t v, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void fill_f32_kernel(float* p, float v, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Evaluate cell (x,y) exactly as src/aread8.cpp:231-256.
// The 3x3 window of directions around a cell (read-only during the sweep): pk[0] = the cell itself, pk[k] = neighbour k;
// bit k of `out` = neighbour k lies outside the raster.
struct D8Window { int16_t pk[9]; unsigned out; };
__device__ __forceinline__ void ad8_load_window(const int16_t* __restrict__ P, int nx, int ny, int x, int y, size_t idx, D8Window& w) {
    w.pk[0] = P[idx];
    w.out = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        const int xn = x + d1(k), yn = y + d2(k);
        const bool in = xn >= 0 && xn < nx && yn >= 0 && yn < ny;
        if (!in) w.out |= 1u << k;
        w.pk[k] = P[in ? size_t(yn) * size_t(nx) + size_t(xn) : idx];
    }
}
// a = w(c) [or 1]; for k = 1..8: a += A[neighbour k] if it drains into c (float32, src/aread8.cpp:231-256).  The values of
// all contributors are requested before the first one is used: one memory round trip per cell.
__device__ __forceinline__ float ad8_evaluate(const D8Window& w, const float* __restrict__ Wt, float w_nodata, float* __restrict__ A, int nx, int x,
                                              int y, size_t idx, int16_t nodata, int contcheck, D8Expr ex) {
    float a;
    if (ex.mode != D8X_SUM) a = Wt[idx];   // the input grid's value as it is (src/D8flowpathextremeup.cpp:170)
    else if (Wt) { const float wt = Wt[idx]; a = is_nodata_f(wt, w_nodata) ? TDX_AREA_NODATA : wt; }   // nodata weight: keeps the initial -1
    else a = 1.0f;
    bool con = false;
    unsigned contrib = 0;
    float ak[9];
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        ak[k] = 0.f;
        if ((w.out >> k) & 1u) { con = true; continue; }
        const int16_t pn = w.pk[k];
        if (is_nodata_s(pn, nodata)) { con = true; continue; }
        if (pn - k == 4 || pn - k == -4) {
            contrib |= 1u << k;
            ak[k] = ld_agent(&A[size_t(y + d2(k)) * size_t(nx) + size_t(x + d1(k))]);
        }
    }
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        if (!((contrib >> k) & 1u)) continue;
        if (is_nodata_f(ak[k], ex.out_nodata)) con = true;
        else if (ex.mode == D8X_SUM) a = a + ak[k];
        else if (ex.mode == D8X_MAX) { if (ak[k] > a) a = ak[k]; }
        else { if (ak[k] < a) a = ak[k]; }
    }
    if (con && contcheck == 1) a = ex.out_nodata;
    return a;
}
__device__ __forceinline__ void ad8_walk_from(size_t idx, const int16_t* __restrict__ P, const float* __restrict__ Wt, float w_nodata, int nx, int ny,
                                              int y_own0, int y_own1, int16_t nodata, int contcheck, int32_t* __restrict__ cnt, float* __restrict__ A,
                                              D8Expr ex) {
    int x = int(idx % size_t(nx)), y = int(idx / size_t(nx));
    D8Window w;
    ad8_load_window(P, nx, ny, x, y, idx, w);
    for (;;) {
        const float a = ad8_evaluate(w, Wt, w_nodata, A, nx, x, y, idx, nodata, contcheck, ex);
        st_agent(&A[idx], a);
        cnt[idx] = CNT_DONE;   // nobody decrements an evaluated cell any more
        const int16_t k = w.pk[0];
        if (k < 1 || k > 8) return;
        const int xn = x + d1(k), yn = y + d2(k);
        if (xn < 0 || xn >= nx || yn < y_own0 || yn >= y_own1) return;   // off the raster, or a neighbour rank's row (released there)
        const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
        drain_stores();                 // value must be at the coherence point before the counter moves
        const int32_t old = __hip_atomic_fetch_sub(&cnt[n], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        D8Window wn;                    // the next hop's window travels with the atomic, not after it
        ad8_load_window(P, nx, ny, xn, yn, n, wn);
        if (old != 1) return;           // somebody else is the last contributor
        x = xn; y = yn; idx = n; w = wn;
    }
}

__global__ __launch_bounds__(256) void ad8_walk_kernel(const int16_t* __restrict__ P, const float* __restrict__ Wt, float w_nodata,
                                                       int nx, int ny, int y_own0, int y_own1, int16_t nodata, int contcheck,
                                                       int32_t* __restrict__ cnt, float* __restrict__ A, D8Expr ex) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = y_own0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= y_own1) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    if (cnt[idx] != CNT_SOURCE) return;          // participates and has no contributor
    ad8_walk_from(idx, P, Wt, w_nodata, nx, ny, y_own0, y_own1, nodata, contcheck, cnt, A, ex);
}

// A halo row after an exchange: cells the neighbouring rank has evaluated since the last look release the owned cell
// they drain into (addBorders + queue refill of src/aread8.cpp:282-303)
__global__ __launch_bounds__(256) void ad8_halo_kernel(const int16_t* __restrict__ P, const float* __restrict__ Wt, float w_nodata, int nx, int ny,
                                                       int y_own0, int y_own1, int16_t nodata, int contcheck, int32_t* __restrict__ cnt,
                                                       float* __restrict__ A, int yh, const float* __restrict__ recv_a, const int32_t* __restrict__ recv_cnt,
                                                       unsigned long long* __restrict__ nchanged, D8Expr ex) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    bool ch = false;
    if (x < nx) {
        const size_t h = size_t(yh) * size_t(nx) + size_t(x);
        if (recv_cnt[x] == CNT_DONE && cnt[h] != CNT_DONE) {
            ch = true;
            cnt[h] = CNT_DONE;
            st_agent(&A[h], recv_a[x]);
            const int16_t k = P[h];
            if (!is_nodata_s(k, nodata) && k >= 1 && k <= 8) {
                const int xn = x + d1(k), yn = yh + d2(k);
                if (xn >= 0 && xn < nx && yn >= y_own0 && yn < y_own1) {
                    const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
                    drain_stores();
                    const int32_t old = __hip_atomic_fetch_sub(&cnt[n], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == 1) ad8_walk_from(n, P, Wt, w_nodata, nx, ny, y_own0, y_own1, nodata, contcheck, cnt, A, ex);
                }
            }
        }
    }
    const unsigned long long m = __ballot(ch);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(nchanged, (unsigned long long)__popcll(m));
}


// =====================================================================================================
// Tile-contraction path for the UNWEIGHTED sweep (no -wg, no outlets).
//
// With unit weights every value is an integer cell count, and float32 adds of integers are exact (and
// therefore order-free) up to 2^24.  The dependency-driven walk above is bound by the LONGEST flow path
// (one device-scope atomic round trip per cell); here the path is contracted through 64x64 tiles so that
// the serial chain is only as long as the number of TILE crossings of the longest path:
//
//   A  ad8_tile_local_kernel   per tile, in LDS: Kahn sweep of the flows that stay inside the tile
//                              (the tile staged as one-hot direction codes; lanes walk downstream with ONE
//                              returning 32-bit LDS atomic per hop that carries count + arrival + the
//                              contamination / not-evaluated flags, one loop per lane over all its sources).
//                              Result S_loc(c) -> global; every cell that leaves the tile towards a
//                              participating cell becomes a NODE of the crossing forest; for every
//                              crossing that enters the tile the lane follows the in-tile path to the
//                              exit cell and records next(node).
//   B  ad8_forest_walk_kernel  the walk of ad8_walk_kernel on the crossing forest (nodes = perimeter
//                              cells, ~1/16 of the cells, paths ~64x shorter), integer packed atomics.
//   C  ad8_tile_apply_kernel   per tile, in LDS: add the flow of every entering crossing along its
//                              in-tile path, convert to float32, apply contamination, write ad8.
//   D  big cells (count > 2^24, where float32 adds round and the k order of src/aread8.cpp:239-256
//                              matters) are re-evaluated with the exact k-ordered pull of ad8_evaluate in
//                              dependency order (ad8_big_* kernels): the main stems - a few thousand cells
//                              at 16384^2, ~10^5 per strip at 65536 columns; chains inside a 64-entry chunk
//                              by an in-binade scan (ad8_big_fold_kernel).
//
// "not evaluated" (the reference leaves -1): cells fed by the p == 0 north-west quirk, cells on / below
// a cycle, and everything downstream of them - carried as a flag next to the contamination flag.
// =====================================================================================================
constexpr int TS = 64;
constexpr int TH = TS + 2;
static_assert(TS == 64, "S_TGT() shifts by 6");
// node_indeg == 0xFFFFFFFF: perimeter / halo cell that is not (yet) a node
constexpr uint32_t NODE_DEAD = 0xFFFFFFFEu;      // node whose cell never completes (cycle / poisoned): never fires
constexpr uint32_t NEXT_NONE = 0xFFFFFFFFu;
constexpr uint32_t NEXT_REMOTE_UP = 0xFFFFFFFDu;    // the crossing leaves the strip through the halo row above
constexpr uint32_t NEXT_REMOTE_DOWN = 0xFFFFFFFCu;  //                                           ... below
constexpr unsigned long long BOX_VALID = 1ull << 63;
constexpr float BIG_MARK = -2.0f;                // provisional ad8 of a cell awaiting the exact float re-evaluation

// Geometry of one strip: tiles cover the OWNED rows [y0, y1) of an array of ny_arr rows; the rows y0-1
// and y1 (when inside the array) are halo rows owned by the neighbouring ranks.  Every cell that can
// carry a crossing has a node id: perimeter cells of the tiles first, then the 2 x nx halo cells.
// start of a tile-contraction call: stage counters = 0, node_indeg / node_next = "none" (all ones), out- / in-boxes (4 nx words) and delivered marks (2 nx bytes) = 0
static __global__ __launch_bounds__(256) void ad8_init_kernel(unsigned long long* __restrict__ d_cnt, uint32_t* __restrict__ node_indeg, uint32_t* __restrict__ node_next, size_t nnodes,
                                                              unsigned long long* __restrict__ boxes, uint8_t* __restrict__ delivered, size_t nx) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < 8) d_cnt[i] = 0ull;
    if (i < nnodes) { node_indeg[i] = 0xFFFFFFFFu; node_next[i] = 0xFFFFFFFFu; }
    if (i < 4 * nx) boxes[i] = 0ull;
    if (i < 2 * nx) delivered[i] = 0;
}
struct Ad8Geom {
    int nx, ny_arr, y0, y1, tiles_x, tiles_y;
    uint32_t nnodes_local;   // tiles * 256
};
__device__ __forceinline__ int rows_valid(const Ad8Geom& g, int ty) { const int r = g.y1 - (g.y0 + ty * TS); return r < TS ? r : TS; }

// LDS word of the local sweep (32 bits: a tile holds at most 4096 cells):
// cnt[0:13) arrivals[13:17) contam[17:21) poison[21:25) indeg[25:29)
__device__ __forceinline__ unsigned lw_pack(unsigned cnt, unsigned arr, unsigned con, unsigned poi, unsigned indeg) {
    return cnt | (arr << 13) | (con << 17) | (poi << 21) | (indeg << 25);
}
__device__ __forceinline__ unsigned lw_cnt(unsigned w) { return w & 0x1FFFu; }
__device__ __forceinline__ unsigned lw_arr(unsigned w) { return (w >> 13) & 15u; }
__device__ __forceinline__ unsigned lw_con(unsigned w) { return (w >> 17) & 15u; }
__device__ __forceinline__ unsigned lw_poi(unsigned w) { return (w >> 21) & 15u; }
__device__ __forceinline__ unsigned lw_indeg(unsigned w) { return (w >> 25) & 15u; }
// node word of the forest walk: cnt[0:32) arrivals[32:42) contam[42:52) poison[52:62)
__device__ __forceinline__ unsigned long long nw_pack(unsigned cnt, unsigned arr, unsigned con, unsigned poi) {
    return (unsigned long long)cnt | ((unsigned long long)arr << 32) | ((unsigned long long)con << 42) | ((unsigned long long)poi << 52);
}
__device__ __forceinline__ unsigned nw_arr(unsigned long long w) { return unsigned(w >> 32) & 1023u; }
__device__ __forceinline__ unsigned nw_con(unsigned long long w) { return unsigned(w >> 42) & 1023u; }
__device__ __forceinline__ unsigned nw_poi(unsigned long long w) { return unsigned(w >> 52) & 1023u; }
// per-cell word between phases A and C: cnt[0:30) contam[30] poison[31]
__device__ __forceinline__ uint32_t cw_pack(unsigned cnt, bool con, bool poi) { return cnt | (con ? 1u << 30 : 0u) | (poi ? 1u << 31 : 0u); }

// position of a perimeter cell (lx, ly) of a tile with rv valid rows in its 256-entry node block
__device__ __forceinline__ int perim_pos(int lx, int ly, int rv) {
    if (ly == 0) return lx;
    if (ly == rv - 1) return 64 + lx;
    if (lx == 0) return 128 + (ly - 1);
    return 190 + (ly - 1);
}
// node id of the array cell (gx, gy): a tile perimeter cell or a halo-row cell
__device__ __forceinline__ uint32_t node_id(const Ad8Geom& g, int gx, int gy) {
    if (gy < g.y0) return g.nnodes_local + uint32_t(gx);
    if (gy >= g.y1) return g.nnodes_local + uint32_t(g.nx) + uint32_t(gx);
    const int ty = (gy - g.y0) / TS, ly = (gy - g.y0) % TS;
    return uint32_t(ty * g.tiles_x + gx / TS) * 256u + uint32_t(perim_pos(gx % TS, ly, rows_valid(g, ty)));
}


// One-hot form of a direction code (ad8_tile_local_kernel): bit c for c in 0 .. 8, OH_SINK for the pure sink of the outlets mode, OH_NODATA for nodata and for
// what lies outside the array, nothing for anything else (16 + p: a cell outside the outlets' closure - it neither takes part nor contributes nor contaminates).
constexpr unsigned OH_NODATA = 0x8000u, OH_SINK = 0x4000u, OH_DIRS = 0x1FEu, OH_PART = 0x1FFu | OH_SINK;
__device__ __forceinline__ unsigned p_onehot(int v, int nodata) {
    // (only the two explicit branches reach bits 15 / 14: a raw code 9 .. 15 in p is a cell that neither takes part nor contaminates - like everywhere else
    // in the pipeline, d8_participates and the reference, which ignores codes outside 0 .. 8 - and not a sink or a nodata cell)
    const unsigned sel = v == nodata ? 15u : (v == int(P_SINK) ? 14u : (unsigned(v) <= 8u ? unsigned(v) : 31u));
    return (1u << sel) & (0x1FFu | OH_SINK | OH_NODATA);
}
// Stage P (tile + ring; ya0 = array row of the tile's first row; outside the array = nodata) into LDS, every cell converted ONCE (each cell is in the 3 x 3 window of
// nine cells: converted where it is used, the conversion cost a quarter of the topology pass).  Addresses are clamped and validity is applied afterwards: loads inside a
// loop with a bounds branch wait for each other.
__device__ __forceinline__ void stage_p_onehot(const int16_t* __restrict__ P, int nx, int ny_arr, int x0, int ya0, int16_t nodata, uint16_t* sO) {
    constexpr int NIT = (TH * TH + 255) / 256, HALF = (NIT + 1) / 2;
    // two batches of loads (a converted value needs a register of its own where two raw int16 shared one: all 17 in flight at once spilled eleven registers;
    // six tiles per CU hide the second memory latency)
#pragma unroll
    for (int b = 0; b < 2; b++) {
        int16_t v[HALF];
        unsigned ok = 0;
#pragma unroll
        for (int i = 0; i < HALF; i++) {
            const int e = int(threadIdx.x) + (b * HALF + i) * 256, ec = e < TH * TH ? e : TH * TH - 1;
            const int ly = ec / TH, lx = ec - ly * TH;
            const int gx = x0 + lx - 1, gy = ya0 + ly - 1;
            const int gxc = gx < 0 ? 0 : (gx >= nx ? nx - 1 : gx), gyc = gy < 0 ? 0 : (gy >= ny_arr ? ny_arr - 1 : gy);
            v[i] = P[size_t(gyc) * size_t(nx) + size_t(gxc)];
            if (gx == gxc && gy == gyc) ok |= 1u << i;
        }
#pragma unroll
        for (int i = 0; i < HALF; i++) {
            const int e = int(threadIdx.x) + (b * HALF + i) * 256;
            if (b * HALF + i < NIT && e < TH * TH) sO[e] = uint16_t(((ok >> i) & 1u) ? p_onehot(int(v[i]), int(nodata)) : OH_NODATA);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// (a cell takes part in the sweep - d8_participates, src/commonLib.cpp:240-386 - iff its code is 0 .. 8 or the outlets mode's pure sink: OH_PART)
__device__ __forceinline__ bool in_tile(int lx, int ly, int rv) { return lx >= 0 && lx < TS && ly >= 0 && ly < rv; }

// ring cell j of a tile with rv valid rows: top row, bottom row, left column, right column
__device__ __forceinline__ bool ring_cell(int j, int rv, int& hx, int& hy) {
    if (j < TH) { hx = j - 1; hy = -1; return true; }
    if (j < 2 * TH) { hx = j - TH - 1; hy = rv; return true; }
    j -= 2 * TH;
    if (j < rv) { hx = -1; hy = j; return true; }
    j -= rv;
    if (j < rv) { hx = TS; hy = j; return true; }
    return false;
}

__device__ unsigned long long g_ad8_dbg[8];   // TDX_AD8_DEBUG=1: phase cycles of ad8_tile_local_kernel summed over the tiles (thread 0)
template <bool DBG, bool FLAT>
__global__ __launch_bounds__(256, 6) void ad8_tile_local_kernel(const int16_t* __restrict__ P, Ad8Geom g, int16_t nodata,
                                                             uint32_t* __restrict__ cellw, unsigned long long* __restrict__ node_acc,
                                                             uint32_t* __restrict__ node_indeg, uint32_t* __restrict__ node_next) {
    unsigned long long tcs[6];
#define AD8_MARK(i) do { if (DBG) tcs[i] = clock64(); } while (0)
    AD8_MARK(0);
    // 26.7 KB of LDS per tile (six workgroups per CU use 160 of its 163.8 KB).  The staged tile holds ONE-HOT direction codes, 66-pitch with the ring; once every
    // lane has derived its topology from it, the table of in-tile targets takes its place at 64-pitch - a hop addresses the table and sAcc with the same cell
    // index, two shifts - so the ring cells' codes, which the entry search needs afterwards, are set aside first: sRing, in ring_cell() order.
    __shared__ uint16_t sO[TH * TH];
    __shared__ unsigned sAcc[TS * TS];
    __shared__ unsigned sIn[256];   // crossings that end at each perimeter cell
    __shared__ uint16_t sRing[4 * TH];
    int16_t* const sT = reinterpret_cast<int16_t*>(sO);
    const int tile = blockIdx.x;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    const int x0 = tx * TS, ya0 = g.y0 + ty * TS, rv = rows_valid(g, ty);
    const int tid = threadIdx.x, lx = tid & 63, ry0 = (tid >> 6) * 16;
    stage_p_onehot(P, g.nx, g.ny_arr, x0, ya0, nodata, sO);
    sIn[tid] = 0u;
    __syncthreads();
    AD8_MARK(1);
    for (int j = tid; j < 4 * TH; j += 256) {
        int hx, hy;
        sRing[j] = ring_cell(j, rv, hx, hy) ? sO[(hy + 1) * TH + hx + 1] : uint16_t(0);
    }
    unsigned src = 0;       // rows of this lane that start a walk
    unsigned exit_up = 0, exit_down = 0;   // rows whose crossing leaves towards the row above / below the cell
    int16_t tgt[16];
    {
        // the lane's window of directions in registers (four quarter bands of 4 rows: 3 x 6 unconditional LDS reads each, so that
        // 6 workgroups per CU still fit the register file), then branch-free topology: initNeighborD8up
        // (src/commonLib.cpp:251-282) and the contamination test of src/aread8.cpp:241-242
        const unsigned colmask = (lx > 0 ? 0xFFu : 0xC7u) & (lx < TS - 1 ? 0xFFu : 0x7Cu);   // neighbours 4 5 6 lie left of the tile's first column, 1 2 8 right of its last
#pragma unroll
        for (int h = 0; h < 4; h++) {
            // The window in one-hot form (p_onehot): "neighbour k drains into the cell" is bit opposite(k) of the neighbour's word, the eight tests one
            // and-or chain per side of the compass, the in-degree a population count under the mask of the neighbours that lie in the tile - instead of
            // eight compare / select chains on lane masks per cell (~100 vector and 50 scalar instructions per cell row; the kernel is issue-bound:
            // profiles/r05m_*, r05n_*).
            unsigned oh[3][6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
#pragma unroll
                for (int i = 0; i < 3; i++) oh[i][j] = sO[(ry0 + 4 * h + j) * TH + lx + i];
            }
            unsigned ohtgt[4];   // code of the cell the row drains to (speculative read, address from a clamped direction)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned dirs = oh[1][q + 1] & OH_DIRS;
                const int pc = dirs ? __ffs(int(dirs)) - 1 : 1;
                ohtgt[q] = sO[(ry0 + 4 * h + q + d2(pc) + 1) * TH + lx + d1(pc) + 1];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = 4 * h + q, ly = ry0 + r;
                const unsigned dirs = oh[1][q + 1] & OH_DIRS;
                const int p = dirs ? __ffs(int(dirs)) - 1 : 0;   // 1 .. 8, or 0: no direction
                const bool part = ly < rv && (oh[1][q + 1] & OH_PART) != 0u;
                // neighbours 1 .. 8 = E NE N NW W SW S SE (src/commonLib.h:83-84); k <= 4 drains into the cell with code k + 4, k >= 5 with k - 4
                const unsigned oE = oh[2][q + 1], oNE = oh[2][q], oN = oh[1][q], oNW = oh[0][q], oW = oh[0][q + 1], oSW = oh[0][q + 2], oS = oh[1][q + 2],
                               oSE = oh[2][q + 2];
                const unsigned hi = (oE & 0x20u) | (oNE & 0x40u) | (oN & 0x80u) | (oNW & 0x100u);   // codes 5 6 7 8 of neighbours 1 2 3 4
                const unsigned lo = (oW & 0x2u) | (oSW & 0x4u) | (oS & 0x8u) | (oSE & 0x10u);       // codes 1 2 3 4 of neighbours 5 6 7 8
                const unsigned drain = (hi >> 5) | (lo << 3);                                        // bit k - 1: neighbour k drains into the cell
                const unsigned rowmask = (ly > 0 ? 0xFFu : 0xF1u) & (ly + 1 < rv ? 0xFFu : 0x1Fu);  // neighbours 2 3 4 lie above the tile's first row, 6 7 8 below its last
                const unsigned indeg = unsigned(__popc(drain & colmask & rowmask));
                const bool con = (((oE | oNE | oN) | (oNW | oW | oSW) | (oS | oSE)) & OH_NODATA) != 0u;
                const bool poison = (oNW & 1u) != 0u;   // k == 4 with p == 0: counted in the in-degree but never decremented (src/commonLib.cpp:257-266, src/aread8.cpp:262)
                int t = -1;
                if (part && p >= 1 && (ohtgt[q] & OH_PART) != 0u) {
                    const int tlx = lx + d1(p), tly = ly + d2(p);
                    t = in_tile(tlx, tly, rv) ? tly * TS + tlx : -2;
                    if (t == -2) {
                        if (d2(p) < 0) exit_up |= 1u << r;
                        if (d2(p) > 0) exit_down |= 1u << r;
                    }
                }
                tgt[r] = int16_t(t);
                sAcc[ly * TS + lx] = part ? lw_pack(1u, 0u, con ? 1u : 0u, poison ? 1u : 0u, indeg) : lw_pack(0u, 0u, 0u, 0u, 15u);
                if (part && indeg == 0) src |= 1u << r;
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps the next quarter's reads behind this quarter's arithmetic)
        }
    }
    __syncthreads();   // every lane is done reading directions: the target table takes the tile's place
#define S_TGT(c) sT[c]
#pragma unroll
    for (int r = 0; r < 16; r++)
        if (ry0 + r < rv) sT[(ry0 + r) * TS + lx] = tgt[r];
    __syncthreads();
    AD8_MARK(2);
    // Kahn sweep of the in-tile flows: one returning 32-bit LDS atomic per hop; the target of the NEXT hop is read alongside the
    // atomic (both only need the current target), so a hop is one LDS round trip, not two
    if (FLAT) {
        // ONE loop per lane: a lane whose walk has ended starts its next source in the very next step, whatever the other lanes of the wave are doing.  (With
        // a walk loop nested in a loop over the lane's sources - the `else` branch, TDX_AD8_KAHN_NESTED=1 - a wave pays, source after source, for the LONGEST
        // walk any of its lanes makes from its k-th source: 271 steps per tile against 165 in a step model of the schedule on a 2048^2 fractal raster's
        // directions; the tile's longest in-tile path is 79 hops.)  Starting a source is a hop like any other: the lane "arrives" at the source cell itself with
        // nothing (add = 0), finds arrivals == in-degree == 0 in the returned word and goes on with that word towards the cell's target.
        int t = -1;
        unsigned add = 0u;
        while (t >= 0 || src != 0u) {
            if (t < 0) {
                const int r = __ffs(int(src)) - 1;
                src &= src - 1u;
                t = (ry0 + r) * TS + lx;
                add = 0u;
            }
            const int tn = S_TGT(t);
            const unsigned nw = __hip_atomic_fetch_add(&sAcc[t], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + add;
            // what this cell hands on: its count, one arrival, "contaminated" / "not evaluated" as 0 / 1 (min(field, 1 << shift): the fields count contributors)
            add = (nw & 0x1FFFu) | (1u << 13) | min(nw & (15u << 17), 1u << 17) | min(nw & (15u << 21), 1u << 21);
            t = lw_arr(nw) == lw_indeg(nw) ? tn : -1;   // not the last contributor: someone else will go on from here
        }
    } else {
    while (src) {
        const int r = __ffs(int(src)) - 1;
        src &= src - 1u;
        const int c = (ry0 + r) * TS + lx;
        unsigned w = sAcc[c];
        int t = S_TGT(c);
        while (t >= 0) {
            const int tn = S_TGT(t);
            const unsigned add = lw_pack(lw_cnt(w), 1u, lw_con(w) ? 1u : 0u, lw_poi(w) ? 1u : 0u, 0u);
            const unsigned nw = __hip_atomic_fetch_add(&sAcc[t], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + add;
            if (lw_arr(nw) != lw_indeg(nw)) break;   // someone else will be the last contributor
            w = nw; t = tn;
        }
    }
    }
    __syncthreads();
    AD8_MARK(3);
    // crossings that enter the tile: follow the in-tile path of the entry cell to where it leaves
    for (int j = tid; j < 4 * TH; j += 256) {
        int hx, hy;
        if (!ring_cell(j, rv, hx, hy)) continue;
        const unsigned dirs = sRing[j] & OH_DIRS;   // the ring cell's direction, set aside before the table took the tile's place
        if (!dirs) continue;
        const int ph = __ffs(int(dirs)) - 1;
        const int vx = hx + d1(ph), vy = hy + d2(ph);
        if (!in_tile(vx, vy, rv)) continue;
        int cur = vy * TS + vx, hops = 0;
        if (lw_indeg(sAcc[cur]) == 15u) continue;        // the entry cell does not participate
        while (S_TGT(cur) >= 0 && hops < TS * TS) { cur = S_TGT(cur); hops++; }
        uint32_t nxt = NEXT_NONE;
        if (S_TGT(cur) == -2) {
            const int pp = perim_pos(cur % TS, cur / TS, rv);
            atomicAdd(&sIn[pp], 1u);
            nxt = uint32_t(tile) * 256u + uint32_t(pp);
        }
        node_next[node_id(g, x0 + hx, ya0 + hy)] = nxt;
    }
    __syncthreads();
    AD8_MARK(4);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int ly = ry0 + r, c = ly * TS + lx;
        const int gx = x0 + lx, gy = ya0 + ly;
        if (gx >= g.nx || ly >= rv) continue;
        const unsigned w = sAcc[c];
        const bool part = lw_indeg(w) != 15u;
        const bool complete = part && lw_arr(w) == lw_indeg(w);
        cellw[size_t(gy) * size_t(g.nx) + size_t(gx)] = part ? cw_pack(lw_cnt(w), lw_con(w) != 0u, lw_poi(w) != 0u || !complete) : 0u;
        if (part && tgt[r] == -2) {
            const int pp = perim_pos(lx, ly, rv);
            const uint32_t nid = uint32_t(tile) * 256u + uint32_t(pp);
            node_acc[nid] = nw_pack(lw_cnt(w), 0u, lw_con(w) ? 1u : 0u, lw_poi(w) ? 1u : 0u);
            node_indeg[nid] = complete ? sIn[pp] : NODE_DEAD;
            const int tgy = gy + (((exit_up >> r) & 1u) ? -1 : (((exit_down >> r) & 1u) ? 1 : 0));
            if (tgy < g.y0) node_next[nid] = NEXT_REMOTE_UP;        // the tile that would record next(node) lives on another rank
            else if (tgy >= g.y1) node_next[nid] = NEXT_REMOTE_DOWN;
        }
    }
#undef S_TGT
    if (DBG) {
        __syncthreads();
        AD8_MARK(5);
        if (tid == 0)
            for (int i = 0; i < 5; i++) atomicAdd(&g_ad8_dbg[i], tcs[i + 1] - tcs[i]);
    }
#undef AD8_MARK
}

// Walk the crossing forest from the completed node u (word w): the last arrival at a node continues.
// A node whose crossing leaves the strip drops its word into the out-box of that side.
__device__ __forceinline__ void forest_walk_from(uint32_t u, unsigned long long w, const Ad8Geom& g, unsigned long long* __restrict__ node_acc,
                                                 const uint32_t* __restrict__ node_indeg, const uint32_t* __restrict__ node_next,
                                                 unsigned long long* __restrict__ outbox) {
    // One memory round trip per hop: the node AFTER the next one (node_next[n], read-only during the walks) and the next node's in-degree are requested
    // together with the returning atomic on the next node - read at the top of the loop they were a second, dependent round trip per hop.
    uint32_t n = node_next[u];
    for (;;) {
        if (n == NEXT_NONE) return;
        if (n == NEXT_REMOTE_UP || n == NEXT_REMOTE_DOWN) {
            // u is a perimeter node in the first / last tile row; its column is tile column * 64 + (pos & 63)
            const uint32_t tile = u >> 8, pos = u & 255u;
            const uint32_t gx = (tile % uint32_t(g.tiles_x)) * TS + (pos & 63u);
            outbox[(n == NEXT_REMOTE_UP ? 0u : uint32_t(g.nx)) + gx] = (w & ~BOX_VALID) | BOX_VALID;
            return;
        }
        const uint32_t nn = node_next[n], ind = node_indeg[n];
        const unsigned long long add = nw_pack(unsigned(w), 1u, nw_con(w) ? 1u : 0u, nw_poi(w) ? 1u : 0u);
        const unsigned long long nw = __hip_atomic_fetch_add(&node_acc[n], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
        if (nw_arr(nw) != ind) return;
        u = n; w = nw; n = nn;
    }
}

// nodes without entering crossings start
__global__ __launch_bounds__(256) void ad8_forest_walk_kernel(Ad8Geom g, unsigned long long* __restrict__ node_acc, const uint32_t* __restrict__ node_indeg,
                                                              const uint32_t* __restrict__ node_next, unsigned long long* __restrict__ outbox) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= g.nnodes_local || node_indeg[i] != 0u) return;
    forest_walk_from(uint32_t(i), node_acc[i], g, node_acc, node_indeg, node_next, outbox);
}

// crossings that arrived from the neighbouring ranks: inbox[side * nx + x] = out-box word of the cell in
// the halo row above (side 0) / below (side 1); each is delivered once and continues the walk here
__global__ __launch_bounds__(256) void ad8_forest_deliver_kernel(Ad8Geom g, const unsigned long long* __restrict__ inbox, uint8_t* __restrict__ delivered,
                                                                 unsigned long long* __restrict__ node_acc, uint32_t* __restrict__ node_indeg,
                                                                 const uint32_t* __restrict__ node_next, unsigned long long* __restrict__ outbox,
                                                                 unsigned long long* __restrict__ ndelivered) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool did = false;
    if (i < 2 * g.nx) {
        const unsigned long long b = inbox[i];
        if ((b & BOX_VALID) && !delivered[i]) {
            delivered[i] = 1;
            did = true;
            const uint32_t hn = g.nnodes_local + uint32_t(i);
            const unsigned long long w = b & ~BOX_VALID;
            // the halo node is complete by construction: arrivals field 0 == in-degree 0
            const unsigned long long hw = nw_pack(unsigned(w), 0u, nw_con(w) ? 1u : 0u, nw_poi(w) ? 1u : 0u);
            node_acc[hn] = hw;
            node_indeg[hn] = 0u;
            forest_walk_from(hn, hw, g, node_acc, node_indeg, node_next, outbox);
        }
    }
    const unsigned long long m = __ballot(did);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(ndelivered, (unsigned long long)__popcll(m));
}

// LDS word of the apply pass: cnt[0:32) contam[32:44) poison[44:56)
// (93 VGPRs: five tiles per CU.  Held to 80 - six tiles, as the LDS allows - it spills 15 registers in the first phase and is 13 % slower: profiles/r05l_*)
__global__ __launch_bounds__(256) void ad8_tile_apply_kernel(const int16_t* __restrict__ P, Ad8Geom g, int16_t nodata,
                                                             uint32_t* __restrict__ cellw, const unsigned long long* __restrict__ node_acc,
                                                             const uint32_t* __restrict__ node_indeg, int contcheck, unsigned big_threshold,
                                                             float* __restrict__ A, uint32_t* __restrict__ biglist,
                                                             unsigned long long* __restrict__ nbig) {
    // 26 KB of LDS per tile: counts (32 bit), the two flags as a bit set, and the staged tile of one-hot direction codes, which the 64-pitch table of
    // in-tile targets replaces once the topology has been derived (the ring's codes set aside in sRing) - as in ad8_tile_local_kernel
    __shared__ uint16_t sO[TH * TH];
    __shared__ unsigned sCnt[TS * TS];
    __shared__ unsigned sFlag[TS * TS / 16];   // 2 bits per cell: contaminated, not evaluated
    __shared__ uint16_t sRing[4 * TH];
    int16_t* const sT = reinterpret_cast<int16_t*>(sO);
    const int tile = blockIdx.x;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    const int x0 = tx * TS, ya0 = g.y0 + ty * TS, rv = rows_valid(g, ty);
    const int tid = threadIdx.x, lx = tid & 63, ry0 = (tid >> 6) * 16;
    stage_p_onehot(P, g.nx, g.ny_arr, x0, ya0, nodata, sO);
    sFlag[tid] = 0u;
    __syncthreads();
    for (int j = tid; j < 4 * TH; j += 256) {
        int hx, hy;
        sRing[j] = ring_cell(j, rv, hx, hy) ? sO[(hy + 1) * TH + hx + 1] : uint16_t(0);
    }
    int16_t tgt[16];
    unsigned partmask = 0;
    {
        // the 16 words of phase A and the directions (own cell, cell drained to) are fetched unconditionally, back to back
        uint32_t cw[16];
        unsigned oown[16], otgt[16];
        const int gx = x0 + lx, gxc = gx < g.nx ? gx : g.nx - 1;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ly = ry0 + r, lyc = ly < rv ? ly : rv - 1;
            cw[r] = cellw[size_t(ya0 + lyc) * size_t(g.nx) + size_t(gxc)];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) oown[r] = sO[(ry0 + r + 1) * TH + lx + 1];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const unsigned dirs = oown[r] & OH_DIRS;
            const int pc = dirs ? __ffs(int(dirs)) - 1 : 1;
            otgt[r] = sO[(ry0 + r + d2(pc) + 1) * TH + lx + d1(pc) + 1];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ly = ry0 + r;
            const unsigned dirs = oown[r] & OH_DIRS;
            const int p = dirs ? __ffs(int(dirs)) - 1 : 0;   // 1 .. 8, or 0: no direction
            int t = -1;
            if (ly < rv && (oown[r] & OH_PART) != 0u) {
                partmask |= 1u << r;
                if (p >= 1) {
                    const int tlx = lx + d1(p), tly = ly + d2(p);
                    if (in_tile(tlx, tly, rv) && (otgt[r] & OH_PART) != 0u) t = tly * TS + tlx;
                }
            }
            tgt[r] = int16_t(t);
            const uint32_t w = (gx < g.nx && ly < rv) ? cw[r] : 0u;
            sCnt[ly * TS + lx] = w & 0x3FFFFFFFu;
            const unsigned fl = w >> 30;   // bit 0 contaminated, bit 1 not evaluated
            if (fl) atomicOr(&sFlag[(ly * TS + lx) >> 4], fl << (2 * ((ly * TS + lx) & 15)));
        }
    }
    __syncthreads();   // every lane is done reading directions: the target table takes the tile's place
#define S_TGT(c) sT[c]
    // (entry cells are looked up through the ring's codes in sRing)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        // participation of an entry cell: remembered as target code -3 ("participates, leaves the tile or ends") vs -1
        if (ry0 + r < rv) sT[(ry0 + r) * TS + lx] = (tgt[r] >= 0) ? tgt[r] : (((partmask >> r) & 1u) ? int16_t(-3) : int16_t(-1));
    }
    __syncthreads();
    for (int j = tid; j < 4 * TH; j += 256) {
        int hx, hy;
        if (!ring_cell(j, rv, hx, hy)) continue;
        const unsigned dirs = sRing[j] & OH_DIRS;
        if (!dirs) continue;
        const int ph = __ffs(int(dirs)) - 1;
        const int vx = hx + d1(ph), vy = hy + d2(ph);
        if (!in_tile(vx, vy, rv)) continue;
        int cur = vy * TS + vx, hops = 0;
        if (S_TGT(cur) == -1) continue;   // the entry cell does not participate
        const uint32_t nid = node_id(g, x0 + hx, ya0 + hy);
        const uint32_t ind = node_indeg[nid];
        const unsigned long long w = node_acc[nid];
        unsigned addc = 0, addf = 2u;     // a crossing that never delivers leaves everything below it unevaluated
        if (ind < NODE_DEAD && nw_arr(w) == ind) { addc = unsigned(w); addf = (nw_con(w) ? 1u : 0u) | (nw_poi(w) ? 2u : 0u); }
        for (;;) {
            if (addc) __hip_atomic_fetch_add(&sCnt[cur], addc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (addf) atomicOr(&sFlag[cur >> 4], addf << (2 * (cur & 15)));
            const int t = S_TGT(cur);
            if (t < 0 || ++hops >= TS * TS) break;
            cur = t;
        }
    }
    __syncthreads();
    unsigned bigmask = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int ly = ry0 + r, c = ly * TS + lx;
        const int gx = x0 + lx, gy = ya0 + ly;
        if (gx >= g.nx || ly >= rv) continue;
        const unsigned cnt = sCnt[c];
        const unsigned fl = (sFlag[c >> 4] >> (2 * (c & 15))) & 3u;
        const bool part = (partmask >> r) & 1u;
        const bool con = (fl & 1u) != 0u, poi = (fl & 2u) != 0u;
        float a = TDX_AREA_NODATA;
        if (part && !poi && !(con && contcheck == 1)) {
            if (cnt > big_threshold) { a = BIG_MARK; bigmask |= 1u << r; cellw[size_t(gy) * size_t(g.nx) + size_t(gx)] = cnt; }
            else a = (float)cnt;   // exact: cnt <= 2^24
        }
        A[size_t(gy) * size_t(g.nx) + size_t(gx)] = a;
    }
#undef S_TGT
    const unsigned long long pos0 = block_reserve(unsigned(__popc(bigmask)), nbig);
    unsigned long long pos = pos0;
#pragma unroll
    for (int r = 0; r < 16; r++)
        if (bigmask & (1u << r)) biglist[pos++] = uint32_t(size_t(ya0 + ry0 + r) * size_t(g.nx) + size_t(x0 + lx));
}

// ---- big cells: exact k-ordered float32 re-evaluation in dependency order ----
// The exact integer count of a cell is larger than the count of every cell that drains into it, so ascending count
// is a dependency order.  The big cells (a few thousand main-stem cells) are sorted by count and evaluated by ONE
// wave, 64 consecutive list entries at a time: all operands of the 64 cells are fetched in parallel; a cell whose
// big contributors were finished in earlier chunks folds its sum at once; the (rare) cells that depend on a cell of
// the same chunk are folded one after the other, handing values through LDS.  The fold itself is the reference's:
// a = 1.0f; for k = 1..8: a += ad8[neighbour k] (float32)  (src/aread8.cpp:231-256).
__global__ __launch_bounds__(256) void ad8_big_keys_kernel(const uint32_t* __restrict__ biglist, unsigned long long nbig, const uint32_t* __restrict__ cellw,
                                                           uint32_t* __restrict__ keys) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q < nbig) keys[q] = cellw[biglist[q]];
}
// (the list kernels below take the list's length from DEVICE memory: the list shrinks from one outer round to the next - ad8_big_compact_kernel - and
// nobody waits for the host to learn by how much; their grids cover the first round's length)
__global__ __launch_bounds__(256) void ad8_big_pos_kernel(const uint32_t* __restrict__ sorted, const unsigned long long* __restrict__ nbig_dev, uint32_t* __restrict__ pos) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q < *nbig_dev) pos[sorted[q]] = uint32_t(q);
}
// The entries of `in` whose cell is still pending (ad8 == BIG_MARK), in list order -> out; *nout = their number.  One workgroup: a list is at most a few
// 10^5 entries (630 542 cells above 2^24 in the eight strips of BASELINE.json configs[3]) and every outer round only keeps what the neighbouring strips
// still block.  Ascending exact count stays a dependency order of what is left.
__global__ __launch_bounds__(1024) void ad8_big_compact_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ root_in, const unsigned long long* __restrict__ nin_dev,
                                                               const float* __restrict__ A, uint32_t* __restrict__ out, uint32_t* __restrict__ root_out,
                                                               unsigned long long* __restrict__ nout) {
    __shared__ unsigned wsum[16];
    const unsigned long long n = *nin_dev;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long base = 0;
    for (unsigned long long q0 = 0; q0 < n; q0 += 1024) {
        const unsigned long long q = q0 + threadIdx.x;
        const uint32_t cell = q < n ? in[q] : 0u;
        const bool keep = q < n && A[cell] == BIG_MARK;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wsum[w] = unsigned(__popcll(m));
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { const unsigned v = wsum[i]; before += i < w ? v : 0u; total += v; }
        if (keep) {
            const unsigned long long o = base + before + unsigned(__popcll(m & ((1ull << lane) - 1ull)));
            out[o] = cell;
            root_out[o] = root_in[q];   // (the tree a cell belongs to: what is left of a tree stays one group)
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *nout = base;
}

// ---- the forest of big cells, tree by tree ----
// The cell a big cell drains to is big as well (its count is larger) unless the flow leaves the strip or ends: the big cells form a forest, a big
// contributor of a big cell lies in the same tree, and trees do not depend on each other.  One wave folding ALL of them in count order costs ~0.12 us per
// cell - 12 ms for the ~10^5 big cells of a 65536 x 8192 strip of BASELINE.json configs[3], the largest item of AreaD8's critical path across the eight
// strips (profiles/r05c_*) - so the list is grouped by tree (root by pointer jumping, stable sort by root: ascending count survives inside a tree) and
// every tree gets a wave of its own.
__global__ __launch_bounds__(256) void ad8_big_next_kernel(const int16_t* __restrict__ P, int nx, int y_own0, int y_own1, const uint32_t* __restrict__ sorted,
                                                           unsigned long long nbig, const uint32_t* __restrict__ pos, const float* __restrict__ A,
                                                           uint32_t* __restrict__ nxt) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nbig) return;
    const size_t c = size_t(sorted[q]);
    const int x = int(c % size_t(nx)), y = int(c / size_t(nx));
    const int p = P[c];
    uint32_t r = uint32_t(q);   // a root: the flow ends, leaves the strip, or goes on into a cell that is not big
    if (p >= 1 && p <= 8) {
        const int xn = x + d1(p), yn = y + d2(p);
        if (xn >= 0 && xn < nx && yn >= y_own0 && yn < y_own1) {
            const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
            if (A[n] == BIG_MARK) r = pos[n];
        }
    }
    nxt[q] = r;
}
__global__ __launch_bounds__(256) void ad8_big_jump_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, unsigned long long nbig) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q < nbig) out[q] = in[in[q]];
}
// first list position of every group of equal roots (the list is grouped by root) -> segbeg[0 .. nseg), segbeg[nseg] = n; one workgroup like ad8_big_compact_kernel
__global__ __launch_bounds__(1024) void ad8_big_segments_kernel(const uint32_t* __restrict__ root, const unsigned long long* __restrict__ n_dev,
                                                                uint32_t* __restrict__ segbeg, unsigned long long* __restrict__ nseg) {
    __shared__ unsigned wsum[16];
    const unsigned long long n = *n_dev;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long base = 0;
    for (unsigned long long q0 = 0; q0 < n; q0 += 1024) {
        const unsigned long long q = q0 + threadIdx.x;
        const bool first = q < n && (q == 0 || root[q] != root[q - 1]);
        const unsigned long long m = __ballot(first);
        if (lane == 0) wsum[w] = unsigned(__popcll(m));
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { const unsigned v = wsum[i]; before += i < w ? v : 0u; total += v; }
        if (first) segbeg[base + before + unsigned(__popcll(m & ((1ull << lane) - 1ull)))] = uint32_t(q);
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { segbeg[base] = uint32_t(n); *nseg = base; }
}

// Step 1 (parallel over the sorted list): everything about a big cell that does not depend on other pending cells - which
// neighbours drain into it, the values of the contributors that are final, the list positions of those that are pending.
constexpr uint32_t BIG_NODEP = 0xFFFFFFFFu;
constexpr uint32_t BIGF_PENDING = 1u, BIGF_BLOCKED = 2u, BIGF_CON = 4u;   // bits 8-15: contributor mask
__global__ __launch_bounds__(256) void ad8_big_gather_kernel(const int16_t* __restrict__ P, int nx, int ny, int y_own0, int y_own1, int16_t nodata,
                                                             const uint32_t* __restrict__ sorted, const unsigned long long* __restrict__ nbig_dev,
                                                             const uint32_t* __restrict__ pos, const float* __restrict__ A, float* __restrict__ vals,
                                                             uint32_t* __restrict__ deps, uint32_t* __restrict__ flags, float* __restrict__ bigval) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= *nbig_dev) return;
    const size_t c = size_t(sorted[q]);
    const int x = int(c % size_t(nx)), y = int(c / size_t(nx));
    const float mine = A[c];
    bigval[q] = mine;   // final value, or BIG_MARK while pending
    uint32_t f = 0;
    if (mine == BIG_MARK) {
        f = BIGF_PENDING;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            float v = 0.f;
            uint32_t dep = BIG_NODEP;
            const int xn = x + d1(k), yn = y + d2(k);
            if (xn < 0 || xn >= nx || yn < 0 || yn >= ny) f |= BIGF_CON;
            else {
                const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
                const int16_t pn = P[n];
                if (is_nodata_s(pn, nodata)) f |= BIGF_CON;
                else if (pn - k == 4 || pn - k == -4) {
                    f |= 0x100u << (k - 1);
                    v = A[n];
                    if (v == BIG_MARK) {
                        // a pending own cell is earlier in the list (smaller count); a pending cell of a neighbouring strip blocks this round
                        if (yn >= y_own0 && yn < y_own1) dep = pos[n];
                        else f |= BIGF_BLOCKED;
                    }
                }
            }
            vals[q * 8 + (k - 1)] = v;
            deps[q * 8 + (k - 1)] = dep;
        }
    }
    flags[q] = f;
}

// Step 2 (ONE wave per tree, 64 consecutive list entries at a time): the fold of the reference in k order.  The values of the
// last BIG_LDS list entries live in LDS, as a ring (an agent-scope round trip per chunk would dominate, and a cell's pending
// contributors are mostly the cells just before it in count order: the next cell up its own stem); a contributor further
// back - a tributary that joins a much larger stem - is read from bigval, where a tree longer than the ring keeps a copy
// (written by this wave long before: stores are drained every 64 chunks, the ring spans 240).  Values of the same chunk are
// handed on in list order.  A cell with a blocked contributor stays pending (BIG_MARK) for the next outer round.
constexpr unsigned BIG_LDS = 15360;   // ring entries (a multiple of 64)
static_assert(BIG_LDS % 64 == 0, "chunks must not straddle the ring's end");
__global__ __launch_bounds__(64) void ad8_big_fold_kernel(int contcheck, int scan_min, unsigned ring, const uint32_t* __restrict__ sorted, const unsigned long long* __restrict__ nbig_dev,
                                                          const uint32_t* __restrict__ segbeg, const unsigned long long* __restrict__ nseg_dev,
                                                          const float* __restrict__ vals, const uint32_t* __restrict__ deps,
                                                          const uint32_t* __restrict__ flags, float* __restrict__ bigval, float* __restrict__ A,
                                                          unsigned long long* __restrict__ nfinal) {
    __shared__ float s_big[BIG_LDS];
    const int lane = threadIdx.x;
    unsigned long long done = 0;
    // The list is grouped by TREE of the big-cell forest (segbeg: first list position of every tree, ascending exact count inside a tree; null: one
    // group).  A big contributor of a big cell lies in the same tree, so trees are independent: one wave per tree, as many waves as the launch has.
    const unsigned long long nseg = segbeg ? *nseg_dev : 1ull;
    for (unsigned long long seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const unsigned long long b0 = segbeg ? (unsigned long long)segbeg[seg] : 0ull, nbig = segbeg ? (unsigned long long)segbeg[seg + 1] : *nbig_dev;   // this wave's part: [b0, nbig)
    const bool spill = nbig - b0 > ring;   // a tree longer than the ring (ring <= BIG_LDS, a multiple of 64; TDX_AD8_BIG_RING: test hook): folded values also go to bigval
    unsigned base = 0u;                    // ring slot of list position chunk0
    float hint = 0.f;   // the value of the last cell folded in this tree: values ascend with the list, so its binade is the guess for the next chunk's scan
    // operands of the first chunk; the next chunk's are fetched while the current one is folded
    uint32_t f = 0, c = 0, dp[8];
    float ak[8], bv = BIG_MARK;
    auto fetch = [&](unsigned long long q) {
        f = 0; c = 0;
        if (q < nbig) {
            f = flags[q]; c = sorted[q]; bv = bigval[q];   // (bigval as the gather left it: the final value of a cell that is not pending)
#pragma unroll
            for (int k = 0; k < 8; k++) { ak[k] = vals[q * 8 + k]; dp[k] = deps[q * 8 + k]; }
        }
    };
    fetch(b0 + (unsigned long long)lane);
    for (unsigned long long chunk0 = b0; chunk0 < nbig; chunk0 += 64) {
        const unsigned long long q = chunk0 + (unsigned long long)lane;
        const uint32_t fl = f, cell = c;
        const float own = bv;
        float a8[8];
        uint32_t d8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { a8[k] = ak[k]; d8[k] = dp[k]; }
        fetch(q + 64);
        const bool pending = (fl & BIGF_PENDING) != 0u;
        bool blocked = (fl & BIGF_BLOCKED) != 0u;
        unsigned inchunk = 0;
        if (pending) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (d8[k] == BIG_NODEP) continue;
                if ((unsigned long long)d8[k] >= q) blocked = true;                // (cannot happen: a contributor's count is smaller)
                else if ((unsigned long long)d8[k] >= chunk0) inchunk |= 1u << k;
                else {   // (the ring holds the entries [chunk0 - ring, chunk0); entry chunk0 - delta sits delta slots before `base`)
                    const unsigned delta = unsigned(chunk0 - (unsigned long long)d8[k]);
                    const unsigned slot = base >= delta ? base - delta : base + ring - delta;
                    a8[k] = delta <= ring ? s_big[slot < ring ? slot : 0u] : ld_agent(&bigval[d8[k]]);
                }
            }
        }
        float result = BIG_MARK;
        auto fold = [&]() {
            float a = 1.0f;
            bool c2 = (fl & BIGF_CON) != 0u, blk = false;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!((fl >> (8 + k)) & 1u)) continue;
                const float v = a8[k];
                if (v == BIG_MARK) blk = true;
                else if (is_nodata_f(v, TDX_AREA_NODATA)) c2 = true;
                else a = a + v;
            }
            if (c2 && contcheck == 1) a = TDX_AREA_NODATA;
            return blk ? BIG_MARK : a;
        };
        if (pending && !blocked && inchunk == 0u) result = fold();
        // Cells that depend on a cell of the same chunk (consecutive main-stem cells): one after the other in list order.
        // The hand-over is register to register (v_readlane with wave-uniform lane numbers), no LDS round trip per step.
        // The common case - exactly ONE contributor in the chunk, at neighbour kd - is prepared in parallel so that a serial
        // step is a dozen instructions: pre = 1 + (contributors before kd), then the handed-over value, then the
        // contributors after kd; a non-contributor adds 0.0f, which leaves a sum >= 1 unchanged, so the order of the
        // reference's float32 additions is kept.
        const bool single = __popc(inchunk) == 1;
        float pre = 1.0f, suf[8];
        bool c2pre = (fl & BIGF_CON) != 0u, blkpre = false;
        int jdep = 0;
        {
            const int kd = single ? __ffs(int(inchunk)) - 1 : 8;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float v = 0.f;
                if (((fl >> (8 + k)) & 1u) && k != kd) {
                    v = a8[k];
                    if (v == BIG_MARK) { blkpre = true; v = 0.f; }
                    else if (is_nodata_f(v, TDX_AREA_NODATA)) { c2pre = true; v = 0.f; }
                }
                pre = pre + (k < kd ? v : 0.f);
                suf[k] = k > kd ? v : 0.f;
                if (k == kd) jdep = int(d8[k] - unsigned(chunk0));
            }
        }
        unsigned long long serial = __ballot(pending && !blocked && inchunk != 0u);
        const unsigned long long singles = __ballot(single);
        // The serial steps as a SCAN.  Inside one binade [B, 2B) (ulp u) a float32 addition is RN(c + x) = c + R(x) where R(x) is x rounded to a multiple of u and
        // depends on c only through the PARITY of c / u (ties go to the even sum).  So the fold of a cell with one contributor in the chunk, as a function of
        // that contributor's value v, is v + D[parity(v)] as long as v and the result stay inside the binade - and D[0], D[1] are what the fold itself returns
        // for the two representatives B and B + u (the reference's additions, in the reference's order).  Functions of this form compose
        // ((g o f)[p] = f[p] + g[p ^ parity(f[p])], all sums exact: multiples of u below 2B), so the in-chunk dependencies - chains of main-stem cells, several
        // of them interleaved - are resolved by pointer jumping over the wave (<= 6 steps) instead of one cell after the other (~0.12 us each).  B is a guess
        // (values ascend with the list: the binade of the largest value known so far); every scanned cell then checks that its contributor's value and its own
        // lie in [B, 2B) - if not (a chunk that straddles a power of two, a contaminated cell, two contributors in the chunk) the serial loop below does the chunk as before.
        if (scan_min > 0 && serial != 0ull && (serial & ~singles) == 0ull && __popcll(serial) >= scan_min) {
            const bool mine = ((serial >> lane) & 1ull) != 0ull;
            float top = (!mine && result >= 1.0f) ? result : 0.f;
#pragma unroll
            for (int o = 32; o; o >>= 1) top = fmaxf(top, __shfl_xor(top, o, 64));
            top = fmaxf(top, hint);
            if (top < 1.0f) top = 16777216.f;   // nothing of this tree is known yet (everything so far waits for a neighbouring strip): any binade will do for what is blocked, and a value that is not fails the check below
            {
                const unsigned bB = __float_as_uint(top) & 0xFF800000u;
                const float B = __uint_as_float(bB), B1 = __uint_as_float(bB + 1u), twoB = B + B;
                float D0 = 0.f, D1 = 0.f;
                if (mine) {
                    float a0 = pre + B, a1 = pre + B1;
#pragma unroll
                    for (int k = 0; k < 8; k++) { a0 = a0 + suf[k]; a1 = a1 + suf[k]; }
                    D0 = a0 - B; D1 = a1 - B1;
                }
                // "still waiting for a cell of a neighbouring strip" (BIG_MARK) travels down a chain the same way: in an outer round whose stem is blocked at its
                // upstream end every cell below it used to be visited one after the other just to stay pending (eight strips of BASELINE.json configs[3]:
                // ~2 ms per outer round and rank, nine rounds on the last rank - profiles/r05k_segments_8strips_d8_*.json)
                bool res = !mine || blkpre;
                float val = (mine && blkpre) ? BIG_MARK : result;
                int par = mine ? jdep : lane;
                for (int it = 0; it < 8 && __ballot(!res) != 0ull; it++) {
                    const int pres = __shfl(int(res), par, 64), pp = __shfl(par, par, 64);
                    const float pv = __shfl(val, par, 64), pD0 = __shfl(D0, par, 64), pD1 = __shfl(D1, par, 64);
                    if (!res) {
                        if (pres) { val = pv == BIG_MARK ? BIG_MARK : pv + ((__float_as_uint(pv) & 1u) ? D1 : D0); res = true; }
                        else {
                            const bool o0 = (__float_as_uint(B + pD0) & 1u) != 0u, o1 = (__float_as_uint(B + pD1) & 1u) != 0u;
                            const float n0 = pD0 + (o0 ? D1 : D0), n1 = pD1 + (o1 ? D0 : D1);
                            D0 = n0; D1 = n1; par = pp;
                        }
                    }
                }
                const float pin = __shfl(val, mine ? jdep : lane, 64);   // the contributor's final value
                const bool good = !mine || (res && (val == BIG_MARK || (!(c2pre && contcheck == 1) && pin >= B && pin < twoB && val < twoB)));
                if (__ballot(!good) == 0ull) {
                    if (mine) result = val;
                    serial = 0ull;
                }
            }
        }
        while (serial) {
            const int i = __ffsll((long long)serial) - 1;
            serial &= serial - 1ull;
            if ((singles >> i) & 1ull) {
                const int j = __builtin_amdgcn_readlane(jdep, i);
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, result), j));
                if (lane == i) {
                    const bool bad = v == BIG_MARK, nod = is_nodata_f(v, TDX_AREA_NODATA);
                    float a = pre + ((bad || nod) ? 0.f : v);
#pragma unroll
                    for (int k = 0; k < 8; k++) a = a + suf[k];   // (only the additions that add something, behind scalar branches: 20 % SLOWER - profiles/r05f_*)
                    if ((c2pre || nod) && contcheck == 1) a = TDX_AREA_NODATA;
                    result = (bad || blkpre) ? BIG_MARK : a;
                }
                continue;
            }
            const unsigned ic = unsigned(__builtin_amdgcn_readlane(int(inchunk), i));
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!((ic >> k) & 1u)) continue;
                const int j = int(unsigned(__builtin_amdgcn_readlane(int(d8[k]), i)) - unsigned(chunk0));
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, result), j));
                if (lane == i) a8[k] = v;
            }
            if (lane == i) result = fold();
        }
        if (q < nbig) s_big[base + unsigned(lane)] = pending ? result : own;   // every entry enters the ring: folded, still pending (BIG_MARK) or final since an earlier round
        base = base + 64u >= ring ? 0u : base + 64u;
        if (pending && result != BIG_MARK) {
            if (spill) st_agent(&bigval[q], result);
            A[cell] = result;   // read again only after this kernel (exchange / host)
            done++;
        }
        if (spill && (((chunk0 - b0) >> 6) & 63ull) == 63ull) drain_stores();   // what falls out of the ring (240 chunks later) is read back through the L2
        else if (spill && ring < 64u * 130u) drain_stores();                        // (a ring shortened for tests: every chunk)
        {
            const float last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, result), 63));
            if (last >= 1.0f) hint = last;
        }
    }
    }   // trees
    if (done) atomicAdd(nfinal, done);
}


}  // namespace


// unweighted, no outlets: tile contraction (see the block comment above).  Multi-strip: crossings that
// leave the strip are handed to the neighbouring rank through per-column out-boxes, and the exact
// re-evaluation of big cells continues across strips through exchanged boundary rows of ad8 - both in
// outer rounds that end when no rank received anything new (the role of the outer while loop with
// share()/addBorders() in src/aread8.cpp:282-303).
int tdx_sort_pairs_u32(tdx_context* ctx, int scratch_slot, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                       size_t n);   // sort_pairs.hip

static int aread8_tiled(tdx_context* ctx, const Strip& st, int16_t* d_p, int16_t p_nodata, int contcheck, float* d_ad8, tdx_stats* stats) {
    hipStream_t s = ctx->stream;
    const int inx = st.nx;
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    Ad8Geom g;
    g.nx = st.nx; g.ny_arr = st.ny_arr; g.y0 = st.y0; g.y1 = st.y1;
    g.tiles_x = (st.nx + TS - 1) / TS; g.tiles_y = (st.y1 - st.y0 + TS - 1) / TS;
    const size_t ntiles = size_t(g.tiles_x) * size_t(g.tiles_y);
    g.nnodes_local = uint32_t(ntiles * 256);
    const size_t nnodes = size_t(g.nnodes_local) + 2 * size_t(st.nx);   // + the halo-row nodes
    uint32_t* cellw = static_cast<uint32_t*>(ctx->scratch(TDX_S_A, n * 4));          // later reused as the big-cell counters
    unsigned long long* node_acc = static_cast<unsigned long long*>(ctx->scratch(TDX_S_B, nnodes * 8));
    uint32_t* node_indeg = static_cast<uint32_t*>(ctx->scratch(TDX_S_C, nnodes * 4));
    uint32_t* node_next = static_cast<uint32_t*>(ctx->scratch(TDX_S_D, nnodes * 4));
    const unsigned big_threshold = getenv("TDX_AD8_BIG_THRESHOLD") ? unsigned(atol(getenv("TDX_AD8_BIG_THRESHOLD"))) : (1u << 24);   // test hook
    const size_t bigcap = big_threshold < (1u << 24) ? n : n / 8 + 4096;
    uint32_t* biglist = static_cast<uint32_t*>(ctx->scratch(TDX_S_E, bigcap * 4));
    unsigned long long* boxes = static_cast<unsigned long long*>(ctx->scratch(TDX_S_F, size_t(st.nx) * 4 * 8));   // out-box[2nx], in-box[2nx]
    uint8_t* delivered = static_cast<uint8_t*>(ctx->scratch(TDX_S_G, size_t(st.nx) * 2));
    if (!cellw || !node_acc || !node_indeg || !node_next || !biglist || !boxes || !delivered) return TDX_ERR_NOMEM;
    unsigned long long *outbox = boxes, *inbox = boxes + 2 * size_t(st.nx);
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    ctx->begin_call(stats);
    strip_mark(ctx, st, "aread8");
    int rc = strip_exchange<int16_t>(ctx, st, d_p, p_nodata);   // directions of the neighbours' boundary rows
    if (rc != TDX_OK) return rc;
    // (one launch instead of five runtime fills: the stage counters, the node arrays, the per-column boxes and their "delivered" marks)
    hipLaunchKernelGGL(ad8_init_kernel, dim3(tdx_blocks_for(std::max(nnodes, size_t(st.nx) * 4), 256)), dim3(256), 0, s, d_cnt, node_indeg, node_next, nnodes, boxes,
                       delivered, size_t(st.nx));
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        static const bool ad8_debug = getenv("TDX_AD8_DEBUG") != nullptr;
        if (ad8_debug) {
            unsigned long long z[8] = {};
            TDX_HIP_CHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_ad8_dbg), z, sizeof(z)));
            hipLaunchKernelGGL((ad8_tile_local_kernel<true, true>), dim3(unsigned(ntiles)), dim3(256), 0, s, d_p, g, p_nodata, cellw, node_acc, node_indeg, node_next);
            TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
            TDX_HIP_CHECK(ctx, hipMemcpyFromSymbol(z, HIP_SYMBOL(g_ad8_dbg), sizeof(z)));
            fprintf(stderr, "ad8_tile_local: cycles per tile: stage %.0f, topology %.0f, targets %.0f, Kahn walks %.0f, entry walks %.0f, write-back %.0f\n",
                    double(z[0]) / double(ntiles), double(z[1]) / double(ntiles) , 0.0, double(z[2]) / double(ntiles), double(z[3]) / double(ntiles), double(z[4]) / double(ntiles));
        } else if (getenv("TDX_AD8_KAHN_NESTED"))   // (A/B hook, read per call: the walk loop nested in the loop over a lane's sources)
            hipLaunchKernelGGL((ad8_tile_local_kernel<false, false>), dim3(unsigned(ntiles)), dim3(256), 0, s, d_p, g, p_nodata, cellw, node_acc, node_indeg, node_next);
        else
            hipLaunchKernelGGL((ad8_tile_local_kernel<false, true>), dim3(unsigned(ntiles)), dim3(256), 0, s, d_p, g, p_nodata, cellw, node_acc, node_indeg, node_next);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    int64_t outer = 1;
    ctx->phase = "forest";
    {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        hipLaunchKernelGGL(ad8_forest_walk_kernel, dim3(tdx_blocks_for(size_t(g.nnodes_local), 256)), dim3(256), 0, s, g, node_acc, node_indeg, node_next,
                           outbox);
        if (stats) stats->launches[TDX_K_ACCUM]++;
        while (st.multi()) {
            const size_t rowb = size_t(st.nx) * 8;
            rc = strip_exchange_buffers(ctx, st, outbox, outbox + st.nx, inbox, inbox + st.nx, rowb);
            if (rc != TDX_OK) return rc;
            TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt + 1, 0, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(ad8_forest_deliver_kernel, dim3(tdx_blocks_for(size_t(2 * st.nx), 256)), dim3(256), 0, s, g, inbox, delivered, node_acc,
                               node_indeg, node_next, outbox, d_cnt + 1);
            int64_t got = 0;
            rc = strip_allreduce_device(ctx, st, d_cnt + 1, 1, TDX_OP_SUM, &got);   // the vote: device counter -> all ranks -> host, one synchronisation
            if (rc != TDX_OK) return rc;
            if (stats) stats->launches[TDX_K_ACCUM]++;
            if (got == 0) break;
            outer++;
        }
    }
    const int64_t outer_forest = outer;
    ctx->phase = "apply";
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        hipLaunchKernelGGL(ad8_tile_apply_kernel, dim3(unsigned(ntiles)), dim3(256), 0, s, d_p, g, p_nodata, cellw, node_acc, node_indeg, contcheck,
                           big_threshold, d_ad8, biglist, d_cnt);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
    const unsigned long long nbig = ctx->h_mail[0];
    if (nbig > bigcap) return tdx_fail(ctx, TDX_ERR_NOMEM, "AreaD8: big-cell list exhausted");
    int64_t nbig_all = int64_t(nbig);
    rc = strip_allreduce(ctx, st, &nbig_all, 1, TDX_OP_SUM);
    if (rc != TDX_OK) return rc;
    ctx->phase = "big cells";
    if (nbig_all > 0) {
        TdxSpan sp(ctx, TDX_K_MISC);
        // dependency order = ascending exact count (left in cellw by the apply pass); list arrays of nbig words: sort keys, sorted keys, the list in count
        // order, two root arrays (pointer jumping; later: the compacted list's roots), the list grouped by tree, its second buffer, the trees' first positions
        const size_t nb1 = size_t(nbig ? nbig : 1);
        uint32_t* keys = static_cast<uint32_t*>(ctx->scratch(TDX_S_H, (nb1 * 8 + 4) * 4));
        if (!keys) return TDX_ERR_NOMEM;
        uint32_t *keys_sorted = keys + nb1, *sorted = keys_sorted + nb1, *rootA = sorted + nb1, *rootB = rootA + nb1, *grouped = rootB + nb1, *listB = grouped + nb1,
                 *segbeg = listB + nb1;
        uint32_t* pos = cellw;   // list position of every big cell (the per-cell words are no longer needed once the keys are out)
        const bool no_trees = getenv("TDX_AD8_BIG_ONE_WAVE") != nullptr;                   // (A/B hook, read per call: one wave folds the whole list in count order)
        // count order, one wave per tree.  (Round 5 also built the fold in LEVEL order - distance to the tree's root, no serial step inside a chunk: bit-identical and
        // 3 - 5 x slower, a level step is a chain of dependent LDS round trips where this fold hands a value on in registers; docs/experiments_r05.md section 1.  Retired in round 6.)
        static const bool no_incremental = getenv("TDX_AD8_BIG_FULL_ROUNDS") != nullptr;   // (A/B hook: every outer round on the whole list)
        // in-chunk dependencies of the count-order fold by pointer jumping when at least this many cells of a 64-entry chunk wait for another cell of the chunk
        // (TDX_AD8_BIG_SCAN=0: always one after the other - A/B hook; read per call)
        const int big_scan = getenv("TDX_AD8_BIG_SCAN") ? std::max(0, atoi(getenv("TDX_AD8_BIG_SCAN"))) : 4;
        // entries of the fold's LDS ring (TDX_AD8_BIG_RING: test hook, rounded to a multiple of 64 in [64, BIG_LDS]: trees longer than that at test sizes)
        const unsigned big_ring = getenv("TDX_AD8_BIG_RING") ? std::min(BIG_LDS, std::max(64u, unsigned(atoi(getenv("TDX_AD8_BIG_RING"))) / 64u * 64u)) : BIG_LDS;
        uint32_t *cur = sorted, *cur_root = rootA, *other = listB, *other_root = rootB;
        unsigned long long* n_cur = d_cnt;            // the apply pass's counter = the first list's length
        const unsigned gb = tdx_blocks_for(nb1, 256);
        if (nbig) {
            hipLaunchKernelGGL(ad8_big_keys_kernel, dim3(gb), dim3(256), 0, s, biglist, nbig, cellw, keys);
            rc = tdx_sort_pairs_u32(ctx, TDX_S_I, keys, keys_sorted, biglist, sorted, size_t(nbig));
            if (rc != TDX_OK) return rc;
            if (!no_trees) {
                hipLaunchKernelGGL(ad8_big_pos_kernel, dim3(gb), dim3(256), 0, s, sorted, n_cur, pos);
                hipLaunchKernelGGL(ad8_big_next_kernel, dim3(gb), dim3(256), 0, s, d_p, inx, st.y0, st.y1, sorted, nbig, pos, d_ad8, rootA);
                uint32_t *ra = rootA, *rb = rootB;
                for (unsigned long long reach = 1; reach < nbig; reach *= 2) {   // after j steps a pointer spans 2^j cells of its chain
                    hipLaunchKernelGGL(ad8_big_jump_kernel, dim3(gb), dim3(256), 0, s, ra, rb, nbig);
                    std::swap(ra, rb);
                }
                // stable sort by root: the trees become groups, ascending count inside each (the keys arrays are free again)
                rc = tdx_sort_pairs_u32(ctx, TDX_S_I, ra, keys, sorted, grouped, size_t(nbig));
                if (rc != TDX_OK) return rc;
                cur = grouped; cur_root = keys; other = listB; other_root = keys_sorted;
            }
        }
        // per big cell: 8 contributor values, 8 dependency positions, flags, current value
        float* big_vals = static_cast<float*>(ctx->scratch(TDX_S_J, nb1 * 4 * 18));
        if (!big_vals) return TDX_ERR_NOMEM;
        uint32_t* big_deps = reinterpret_cast<uint32_t*>(big_vals) + nb1 * 8;
        uint32_t* big_flags = big_deps + nb1 * 8;
        float* big_val = reinterpret_cast<float*>(big_flags + nb1);
        rc = strip_exchange<float>(ctx, st, d_ad8, TDX_AREA_NODATA);   // which halo cells await re-evaluation
        if (rc != TDX_OK) return rc;
        // Outer rounds: a cell whose contributor in a neighbouring strip is still pending stays pending; what the round finished travels in the exchanged
        // boundary rows.  Every round works on the list of what is STILL pending (groups and their order survive: ad8_big_compact_kernel), not on all big
        // cells again.
        int flip = 0;
        for (;;) {
            if (nbig) {
                hipLaunchKernelGGL(ad8_big_pos_kernel, dim3(gb), dim3(256), 0, s, cur, n_cur, pos);
                hipLaunchKernelGGL(ad8_big_gather_kernel, dim3(gb), dim3(256), 0, s, d_p, inx, st.ny_arr, st.y0, st.y1, p_nodata, cur, n_cur, pos, d_ad8, big_vals,
                                   big_deps, big_flags, big_val);
                if (no_trees)
                    hipLaunchKernelGGL(ad8_big_fold_kernel, dim3(1), dim3(64), 0, s, contcheck, big_scan, big_ring, cur, n_cur, static_cast<const uint32_t*>(nullptr),
                                       static_cast<const unsigned long long*>(nullptr), big_vals, big_deps, big_flags, big_val, d_ad8, d_cnt + 2);
                else {
                    hipLaunchKernelGGL(ad8_big_segments_kernel, dim3(1), dim3(1024), 0, s, cur_root, n_cur, segbeg, d_cnt + 6);
                    hipLaunchKernelGGL(ad8_big_fold_kernel, dim3(unsigned(std::min<unsigned long long>(nbig, 2ull * unsigned(ctx->num_cus)))), dim3(64), 0, s, contcheck, big_scan, big_ring, cur,
                                       n_cur, segbeg, d_cnt + 6, big_vals, big_deps, big_flags, big_val, d_ad8, d_cnt + 2);
                }
                if (st.multi() && !no_incremental) {
                    unsigned long long* n_next = d_cnt + 4 + flip;
                    hipLaunchKernelGGL(ad8_big_compact_kernel, dim3(1), dim3(1024), 0, s, cur, cur_root, n_cur, d_ad8, other, other_root, n_next);
                    std::swap(cur, other);
                    std::swap(cur_root, other_root);
                    n_cur = n_next;
                    flip ^= 1;
                }
            }
            if (stats) stats->launches[TDX_K_MISC]++;
            if (!st.multi()) break;
            int64_t changed = 0;   // halo cells that became final on the neighbouring ranks
            rc = strip_exchange<float>(ctx, st, d_ad8, TDX_AREA_NODATA, nullptr, 0, &changed);
            if (rc != TDX_OK) return rc;
            rc = strip_allreduce(ctx, st, &changed, 1, TDX_OP_SUM);
            if (rc != TDX_OK) return rc;
            if (changed == 0) break;
            outer++;
        }
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    if (st.multi()) {
        static const bool trace = getenv("TDX_COMM_TRACE") != nullptr && atoi(getenv("TDX_COMM_TRACE")) != 0;
        if (trace)
            fprintf(stderr, "taudem_amd[rank %d/%d] aread8: %lld outer rounds of the crossing forest, %lld of the big cells; %llu big cells here, %lld in all strips\n",
                    st.comm->rank, st.comm->size, (long long)outer_forest, (long long)(outer - outer_forest + 1), nbig, (long long)nbig_all);
    }
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) { stt->rounds = outer; stt->cells_evaluated = nbig_all; }
    return TDX_OK;
}

// One strip of aread8() (src/aread8.cpp:175-307).  Outlets (array coordinates) restrict the sweep to their upstream
// closure; unit weights without TDX_AD8_WALK take the tile-contraction path, everything else the exact pull walk.
static int aread8_impl(tdx_context* ctx, const Strip& st, int16_t* d_p, int16_t p_nodata, const float* d_w, float w_nodata, int contcheck,
                       const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_ad8, tdx_stats* stats,
                       D8Expr ex = D8Expr{D8X_SUM, TDX_AREA_NODATA}) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = st.nx, iny = st.ny_arr;
    const size_t n = size_t(inx) * size_t(iny);
    const bool force_walk = getenv("TDX_AD8_WALK") != nullptr;
    bool tiled = ex.mode == D8X_SUM && !d_w && !force_walk;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    int16_t* p_use = d_p;
    int rc;
    if (n_outlets >= 0) {
        // upstream closure of the outlets, then the ordinary sweep on the re-coded directions
        rc = strip_exchange<int16_t>(ctx, st, d_p, p_nodata);
        if (rc != TDX_OK) return rc;
        const tilek::TileGeom geom = tilek::make_geom(inx, iny, st.y0, st.y1);
        const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y);
        int32_t* reach = static_cast<int32_t*>(ctx->scratch(TDX_S_N, n * 4));
        uint8_t* mask = static_cast<uint8_t*>(ctx->scratch(TDX_S_O, n));
        int16_t* pprime = static_cast<int16_t*>(ctx->scratch(TDX_S_P, n * 2));
        uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));
        unsigned long long* counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
        int32_t* d_oxy = static_cast<int32_t*>(ctx->scratch(TDX_S_R, size_t(n_outlets ? n_outlets : 1) * 8));
        if (!reach || !mask || !pprime || !flags || !counts || !d_oxy) return TDX_ERR_NOMEM;
        TDX_HIP_CHECK(ctx, hipMemsetAsync(reach, 0, n * 4, s));
        TDX_HIP_CHECK(ctx, hipMemsetAsync(flags, 0, ntiles * 4, s));
        hipLaunchKernelGGL(d8_reach_mask_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_p, n, p_nodata, mask);
        if (n_outlets > 0) {
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_oxy, outlet_x, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_oxy + n_outlets, outlet_y, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(reach_seed_kernel, dim3(tdx_blocks_for(size_t(n_outlets), 256)), dim3(256), 0, s, d_oxy, d_oxy + n_outlets, int(n_outlets), inx,
                               iny, st.y0, st.y1, geom.tiles_x, reach, flags);
        }
        int64_t rr = 0, ll = 0;
        rc = reach_closure(ctx, st, reach, mask, flags, flags + ntiles, counts, &rr, &ll);
        if (rc != TDX_OK) return rc;
        hipLaunchKernelGGL(d8_apply_reach_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_p, reach, n, p_nodata, pprime);
        p_use = pprime;
    }
    // The tile contraction carries exact cell counts in 32-bit words (node_acc, the apply pass's LDS counters, the big-cell
    // sort keys).  A count is at most the number of PARTICIPATING cells of the whole raster (all strips; with outlets: of their
    // upstream closure), so that number must stay below 2^32: a raster of fewer cells passes at once, a larger one - 65536 x 65536,
    // BASELINE.json configs[3], is exactly 2^32 cells, 4 294 700 699 of them with a direction - has its participating cells counted
    // (one streaming pass over p).  Beyond that the tile dependency sweep runs, whose float32 adds have no such limit.
    // TDX_AD8_COUNT_LIMIT: test hook for that switch.
    // (cell indices of the tile contraction are uint32: the strip's ARRAY must hold at most 2^32 - 1 cells whatever the participating count says - the entry
    // points refuse larger strips; this keeps the two conditions side by side)
    if (n > 0xffffffffull) return tdx_fail(ctx, TDX_ERR_ARG, "aread8: strip array larger than 2^32 - 1 cells");
    if (tiled) {
        int64_t cells = int64_t(st.nx) * int64_t(st.y1 - st.y0);
        int rc0 = strip_allreduce(ctx, st, &cells, 1, TDX_OP_SUM);
        if (rc0 != TDX_OK) return rc0;
        const int64_t limit = getenv("TDX_AD8_COUNT_LIMIT") ? atoll(getenv("TDX_AD8_COUNT_LIMIT")) : int64_t(0xFFFFFFFFll);
        if (cells > limit) {
            TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt + 3, 0, sizeof(unsigned long long), s));
            const size_t first = size_t(st.y0) * size_t(inx), nown = size_t(st.y1 - st.y0) * size_t(inx);
            hipLaunchKernelGGL(d8_count_participating_kernel, dim3(std::min(tdx_blocks_for((nown + 7) / 8, 256), 2048u)), dim3(256), 0, s, p_use + first, nown, p_nodata, d_cnt + 3);
            int64_t npart = 0;
            rc0 = strip_allreduce_device(ctx, st, d_cnt + 3, 1, TDX_OP_SUM, &npart);
            if (rc0 != TDX_OK) return rc0;
            if (npart > limit) tiled = false;
        }
    }
    if (tiled && getenv("TDX_AD8_SWEEP") == nullptr) return aread8_tiled(ctx, st, p_use, p_nodata, contcheck, d_ad8, stats);

    if (!force_walk) {
        // ---- tile dependency sweep (weights, extremes, rasters beyond the 32-bit counts; TDX_AD8_SWEEP=1: always) ----
        uint32_t* info = static_cast<uint32_t*>(ctx->scratch(TDX_S_A, n * 4));
        const tilek::TileGeom geom = tilek::make_geom(inx, iny, st.y0, st.y1);
        const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y);
        uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));
        unsigned long long* counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
        if (!info || !flags || !counts) return TDX_ERR_NOMEM;
        ctx->begin_call(stats);
        strip_mark(ctx, st, "aread8");
        rc = strip_exchange<int16_t>(ctx, st, p_use, p_nodata);
        if (rc != TDX_OK) return rc;
        {
            TdxSpan sp(ctx, TDX_K_STENCIL);
            hipLaunchKernelGGL(d8sweep::setup_kernel, dim3((inx + 63) / 64, (iny + 3) / 4), dim3(256), 0, s, p_use, inx, iny, p_nodata, 0, nullptr, 0, nullptr, info);
            const size_t first = size_t(st.y0) * size_t(inx), nown = size_t(st.y1 - st.y0) * size_t(inx);
            hipLaunchKernelGGL(d8sweep::init_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, info, d_ad8, first, nown, ex.out_nodata);
            if (stats) stats->launches[TDX_K_STENCIL] += 2;
        }
        rc = strip_exchange<float>(ctx, st, d_ad8, ex.out_nodata);
        if (rc != TDX_OK) return rc;
        int64_t rounds = 0, launches = 0, outer = 1;
        {
            TdxSpan sp(ctx, TDX_K_ACCUM);
            d8sweep::SumMaxMin alg{ex.mode, ex.out_nodata, w_nodata, contcheck, d_w != nullptr};
            d8sweep::Arrays<d8sweep::SumMaxMin> A{d_ad8, d_w, nullptr, nullptr, info};
            rc = d8sweep::run(ctx, st, alg, A, flags, counts, &rounds, &launches, &outer);
            if (rc != TDX_OK) return rc;
            const size_t first = size_t(st.y0) * size_t(inx), nown = size_t(st.y1 - st.y0) * size_t(inx);
            hipLaunchKernelGGL(d8sweep::finish_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, d_ad8, first, nown, ex.out_nodata);   // never evaluated: nodata
            if (stats) stats->launches[TDX_K_ACCUM] += launches;
        }
        TDX_HIP_CHECK(ctx, hipGetLastError());
        tdx_stats* stt = stats;
        ctx->end_call();
        if (stt) { stt->rounds = outer; stt->cells_evaluated = rounds; }
        return TDX_OK;
    }

    // ---- exact pull walk (TDX_AD8_WALK: A/B hook) ----
    int32_t* cnt = static_cast<int32_t*>(ctx->scratch(TDX_S_A, n * 4));
    float* recvbuf = static_cast<float*>(ctx->scratch(TDX_S_K, size_t(inx) * 4 * 4));
    if (!cnt || !recvbuf) return TDX_ERR_NOMEM;
    const dim3 grid2d((inx + 63) / 64, (st.y1 - st.y0 + 3) / 4);
    ctx->begin_call(stats);
    strip_mark(ctx, st, "aread8");
    rc = strip_exchange<int16_t>(ctx, st, p_use, p_nodata);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        hipLaunchKernelGGL(ad8_setup_kernel, grid2d, dim3(256), 0, s, p_use, inx, iny, st.y0, st.y1, p_nodata, cnt, d_ad8, ex.out_nodata);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    rc = strip_exchange<float>(ctx, st, d_ad8, ex.out_nodata);
    if (rc != TDX_OK) return rc;
    rc = strip_exchange<int32_t>(ctx, st, cnt, CNT_NOT_PART);
    if (rc != TDX_OK) return rc;
    int64_t outer = 1;
    {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        hipLaunchKernelGGL(ad8_walk_kernel, grid2d, dim3(256), 0, s, p_use, d_w, w_nodata, inx, iny, st.y0, st.y1, p_nodata, contcheck, cnt, d_ad8, ex);
        if (stats) stats->launches[TDX_K_ACCUM]++;
        while (st.multi()) {
            const size_t rowb = size_t(inx) * 4;
            float *r_a_up = recvbuf, *r_a_dn = recvbuf + inx;
            int32_t *r_c_up = reinterpret_cast<int32_t*>(recvbuf + 2 * size_t(inx)), *r_c_dn = reinterpret_cast<int32_t*>(recvbuf + 3 * size_t(inx));
            rc = strip_exchange_buffers(ctx, st, d_ad8 + size_t(st.y0) * inx, d_ad8 + size_t(st.y1 - 1) * inx, r_a_up, r_a_dn, rowb);
            if (rc != TDX_OK) return rc;
            rc = strip_exchange_buffers(ctx, st, cnt + size_t(st.y0) * inx, cnt + size_t(st.y1 - 1) * inx, r_c_up, r_c_dn, rowb);
            if (rc != TDX_OK) return rc;
            TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt + 1, 0, sizeof(unsigned long long), s));
            const unsigned gx = tdx_blocks_for(size_t(inx), 256);
            if (st.up)
                hipLaunchKernelGGL(ad8_halo_kernel, dim3(gx), dim3(256), 0, s, p_use, d_w, w_nodata, inx, iny, st.y0, st.y1, p_nodata, contcheck, cnt, d_ad8,
                                   st.y0 - 1, r_a_up, r_c_up, d_cnt + 1, ex);
            if (st.down)
                hipLaunchKernelGGL(ad8_halo_kernel, dim3(gx), dim3(256), 0, s, p_use, d_w, w_nodata, inx, iny, st.y0, st.y1, p_nodata, contcheck, cnt, d_ad8,
                                   st.y1, r_a_dn, r_c_dn, d_cnt + 1, ex);
            int64_t changed = 0;
            rc = strip_allreduce_device(ctx, st, d_cnt + 1, 1, TDX_OP_SUM, &changed);   // the vote: device counter -> all ranks -> host, one synchronisation
            if (rc != TDX_OK) return rc;
            if (stats) stats->launches[TDX_K_ACCUM]++;
            if (changed == 0) break;
            outer++;
        }
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) stt->rounds = outer;
    return TDX_OK;
}

extern "C" int tdx_aread8_strip(tdx_context* ctx, const tdx_comm* comm, int16_t* d_p, int64_t nx, int64_t ny_local, int16_t p_nodata,
                                   const float* d_w, float w_nodata, int contcheck, const int32_t* outlet_x, const int32_t* outlet_row,
                                   int64_t n_outlets, float* d_ad8, tdx_stats* stats) {
    if (!ctx || !d_p || !d_ad8 || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_aread8_strip: bad argument");
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    if (n_outlets > 0 && (!outlet_x || !outlet_row)) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_aread8_strip: outlets missing");
    return aread8_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_p, p_nodata, d_w, w_nodata, contcheck, outlet_x, outlet_row, n_outlets, d_ad8,
                       stats);
}

extern "C" int tdx_aread8_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata,
                              const float* d_w, float w_nodata, int contcheck,
                              const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                              float* d_ad8, tdx_stats* stats) {
    if (!ctx || !d_p || !d_ad8 || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_aread8_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    if (n_outlets > 0 && (!outlet_x || !outlet_y)) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_aread8_dev: outlets missing");
    return aread8_impl(ctx, strip_single(int(nx), int(ny)), const_cast<int16_t*>(d_p), p_nodata, d_w, w_nodata, contcheck, outlet_x, outlet_y, n_outlets,
                       d_ad8, stats);
}

extern "C" int tdx_aread8(tdx_context* ctx, const int16_t* p, int64_t nx, int64_t ny, int16_t p_nodata,
                          const float* w, float w_nodata, int contcheck,
                          const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                          float* ad8, tdx_stats* stats) {
    if (!ctx || !p || !ad8 || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_aread8: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    int16_t* d_p = static_cast<int16_t*>(ctx->scratch(TDX_S_IO0, n * 2));
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_w = w ? static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4)) : nullptr;
    if (!d_p || !d_a || (w && !d_w)) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_p, p, n * 2, hipMemcpyHostToDevice, ctx->stream));
    if (w) TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_w, w, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = tdx_aread8_dev(ctx, d_p, nx, ny, p_nodata, d_w, w_nodata, contcheck, outlet_x, outlet_y, n_outlets, d_a, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ad8, d_a, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}

// ---- D8FlowPathExtremeUp (src/D8flowpathextremeup.cpp:58-285): the same dependency sweep, outlets closure and strip
// protocol as the weighted AreaD8, with max / min instead of the sum.  sa: the grid whose upstream extreme is sought; ssa: result, nodata
// -FLT_MAX (MISSINGFLOAT, src/commonLib.h:80).
static int extremeup_check(tdx_context* ctx, const void* p, const void* sa, const void* ssa, int64_t nx, int64_t ny, const char* who) {
    if (!ctx || !p || !sa || !ssa || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, who);
    return TDX_OK;
}
extern "C" int tdx_d8flowpathextremeup_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata, const float* d_sa, int usemax,
                                           int contcheck, const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_ssa,
                                           tdx_stats* stats) {
    int rc = extremeup_check(ctx, d_p, d_sa, d_ssa, nx, ny, "tdx_d8flowpathextremeup_dev: bad argument");
    if (rc != TDX_OK) return rc;
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    if (n_outlets > 0 && (!outlet_x || !outlet_y)) return tdx_fail(ctx, TDX_ERR_ARG, "outlets missing");
    return aread8_impl(ctx, strip_single(int(nx), int(ny)), const_cast<int16_t*>(d_p), p_nodata, d_sa, 0.f, contcheck, outlet_x, outlet_y, n_outlets, d_ssa, stats,
                       D8Expr{usemax ? D8X_MAX : D8X_MIN, -FLT_MAX});
}
extern "C" int tdx_d8flowpathextremeup_strip(tdx_context* ctx, const tdx_comm* comm, int16_t* d_p, int64_t nx, int64_t ny_local, int16_t p_nodata,
                                             const float* d_sa, int usemax, int contcheck, const int32_t* outlet_x, const int32_t* outlet_row,
                                             int64_t n_outlets, float* d_ssa, tdx_stats* stats) {
    int rc = extremeup_check(ctx, d_p, d_sa, d_ssa, nx, ny_local, "tdx_d8flowpathextremeup_strip: bad argument");
    if (rc != TDX_OK) return rc;
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    if (n_outlets > 0 && (!outlet_x || !outlet_row)) return tdx_fail(ctx, TDX_ERR_ARG, "outlets missing");
    return aread8_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_p, p_nodata, d_sa, 0.f, contcheck, outlet_x, outlet_row, n_outlets, d_ssa, stats,
                       D8Expr{usemax ? D8X_MAX : D8X_MIN, -FLT_MAX});
}
extern "C" int tdx_d8flowpathextremeup(tdx_context* ctx, const int16_t* p, int64_t nx, int64_t ny, int16_t p_nodata, const float* sa, int usemax, int contcheck,
                                       const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* ssa, tdx_stats* stats) {
    int rc = extremeup_check(ctx, p, sa, ssa, nx, ny, "tdx_d8flowpathextremeup: bad argument");
    if (rc != TDX_OK) return rc;
    const size_t n = size_t(nx) * size_t(ny);
    int16_t* d_p = static_cast<int16_t*>(ctx->scratch(TDX_S_IO0, n * 2));
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_w = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    if (!d_p || !d_a || !d_w) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_p, p, n * 2, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_w, sa, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = tdx_d8flowpathextremeup_dev(ctx, d_p, nx, ny, p_nodata, d_w, usemax, contcheck, outlet_x, outlet_y, n_outlets, d_a, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ssa, d_a, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
