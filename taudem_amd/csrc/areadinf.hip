// AreaDinf and DinfDecayAccum on gfx950: replace the compute parts of area() (src/areadinf.cpp:151-265)
// and dmarea() (src/dinfdecayaccum.cpp:178-291) plus initNeighborDinfup() (src/commonLib.cpp:92-238).
//
// Same dependency-driven evaluation as AreaD8 (aread8.hip): a cell is evaluated, by PULLING its
// contributors in k = 1..8 order with the reference's mixed float/double arithmetic, at the moment its
// last contributor finishes.  D-infinity splits flow between up to two downslope neighbours
// (prop(), src/commonLib.cpp:76-91), so the lane that finishes a cell may release two cells: it
// continues into the first and parks the second on a small private stack (overflow goes to a global
// list that a follow-up launch drains).
//
// prop() needs atan2(dy,dx) only through the per-row table aref[] - computed on the host with the host
// libm (one value per row); everything else is exactly-rounded +,-,/ on the device.
#include "context.hpp"
#include "d8_sweep.hpp"
#include "device_common.hpp"
#include "dinf_outlets.hpp"
#include "dinf_prop.hpp"
#include "flats.hpp"
#include "strips.hpp"

#include <type_traits>

namespace {
using namespace tdxk;

constexpr int32_t CNT_NOT_PART = 0x40000000;
constexpr int32_t CNT_SOURCE = -1;
// Result of a participating cell before it has been evaluated: a quiet NaN with a payload no arithmetic produces.  A
// walker publishes a cell's value and decrements the downstream counters WITHOUT waiting for the store in between (that
// wait was one of three memory round trips per hop); the lane that later evaluates a downstream cell re-reads a
// contributor that still shows this pattern (the store was issued before the decrement it has observed, so it lands).
constexpr uint32_t DINF_PENDING_BITS = 0x7FC0DEADu;
constexpr int32_t CNT_DONE = -2;       // evaluated: what a strip neighbour looks for in the exchanged boundary rows
constexpr int WALK_STACK = 48;
constexpr float ANG_OUTSIDE = TDX_ANG_OUTSIDE, ANG_SINK = TDX_ANG_SINK;   // re-coded angles of outlets mode (dinf_outlets.hpp)

// does neighbour k of (x,y) drain into (x,y)?  returns the proportion (>0) or a value <= 0
__device__ __forceinline__ double inflow_prop(const float* __restrict__ ANG, const RowProp* __restrict__ rows, int nx, int ny, int x, int y,
                                              int k, float nodata, size_t* nidx, bool* missing) {
    const int xn = x + d1(k), yn = y + d2(k);
    *missing = false;
    if (xn < 0 || xn >= nx || yn < 0 || yn >= ny) { *missing = true; return -1.; }
    const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
    const float an = ANG[n];
    if (is_nodata_f(an, nodata)) { *missing = true; return -1.; }
    if (an == ANG_SINK) *missing = true;   // an outlet on a cell without angle: a sink for itself, still "no angle" for the contamination test beside it
    *nidx = n;
    return prop_dev(an, (k + 4) % 8, rows[yn].a2);
}

// Per cell, once: which neighbours drain into it (bit k-1 of the low byte; the test of initNeighborDinfup,
// src/commonLib.cpp:99-131) and whether any neighbour is missing (bit 8: outside the raster or nodata - the
// contamination test of src/areadinf.cpp:196-199).  The walk then touches only real contributors: 1-3 proportion
// evaluations per cell instead of 16.  With `cnt` the in-degree of every cell is initialised too.
__global__ __launch_bounds__(256) void dinf_setup_kernel(const float* __restrict__ ANG, int nx, int ny, int y_own0, int y_own1, float nodata,
                                                         const RowProp* __restrict__ rows, uint16_t* __restrict__ info, int32_t* __restrict__ cnt,
                                                         float* __restrict__ OUT, float out_nodata) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = y_own0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= y_own1) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    // the masks are kept for EVERY cell: an outlet may sit on a cell without an angle (e.g. the edge ring) and is then
    // evaluated from the neighbours that drain into it (src/commonLib.cpp:165-233)
    int32_t c = 0;
    unsigned inf = 0;
    for (int k = 1; k <= 8; k++) {
        size_t n; bool miss;
        const float p = (float)inflow_prop(ANG, rows, nx, ny, x, y, k, nodata, &n, &miss);   // `float p` in the reference (commonLib.cpp:99)
        if (miss) inf |= 0x100u;
        else if (p > 0.0f) { c++; inf |= 1u << (k - 1); }
    }
    if (c == 0) c = CNT_SOURCE;
    const float ang = ANG[idx];
    if (is_nodata_f(ang, nodata) || ang == ANG_OUTSIDE) c = CNT_NOT_PART;
    else {
        // where this cell sends flow: prop(ang, k) can be positive only for the two directions that bracket the angle
        // (src/commonLib.cpp:83-88); kept with the cell so that the sweep needs no proportion to find its targets
        const double a2 = rows[y].a2;
        const int s1 = dinf_sector(ang, a2);
        inf |= unsigned(s1 - 1) << 9;
        if (prop_dev(ang, s1, a2) > 0.0) inf |= 1u << 12;
        if (prop_dev(ang, s1 % 8 + 1, a2) > 0.0) inf |= 1u << 13;
    }
    info[idx] = uint16_t(inf);
    if (cnt) cnt[idx] = c;
    OUT[idx] = (c == CNT_NOT_PART) ? out_nodata : __uint_as_float(DINF_PENDING_BITS);
}

__global__ void fill_i32_kernel(int32_t* p, int32_t v, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void fill_f32_kernel(float* p, float v, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- outlets: upstream closure through the tile relaxation engine (flats.hpp: reach_closure) ----
// Outlets mode runs the ordinary sweep on re-coded angles: cells outside the closure keep "a valid angle" for the
// contamination test but neither participate nor contribute (ANG_OUTSIDE, for which prop() is 0 in every direction);
// an outlet on a cell without angle participates as a pure sink (ANG_SINK) - src/commonLib.cpp:165-233.

// mask of the reachability relaxation: the (at most two) neighbours a cell sends flow to
__global__ __launch_bounds__(256) void dinf_reach_mask_kernel(const float* __restrict__ ANG, int nx, int ny, float nodata, const RowProp* __restrict__ rows,
                                                              uint8_t* __restrict__ mask) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    const float ang = ANG[idx];
    unsigned m = 0;
    if (!is_nodata_f(ang, nodata)) {
        const double a2 = rows[y].a2;
        const int s1 = dinf_sector(ang, a2);
        if (prop_dev(ang, s1, a2) > 0.0) m |= 1u << (s1 - 1);
        const int s2 = s1 % 8 + 1;
        if (prop_dev(ang, s2, a2) > 0.0) m |= 1u << (s2 - 1);
    }
    mask[idx] = uint8_t(m);
}
__global__ __launch_bounds__(256) void dinf_reach_seed_kernel(const int32_t* __restrict__ ox, const int32_t* __restrict__ oy, int nout, int nx, int ny, int y_own0,
                                                              int y_own1, int tiles_x, int32_t* __restrict__ reach, uint32_t* __restrict__ tile_flags) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    const int x = ox[o], y = oy[o];
    if (x < 0 || x >= nx || y < y_own0 || y >= y_own1) return;
    reach[size_t(y) * size_t(nx) + size_t(x)] = 1;
    tilek::activate_tiles_around(x, y, nx, ny, tiles_x, tile_flags);
}
__global__ __launch_bounds__(256) void dinf_apply_reach_kernel(const float* __restrict__ ANG, const int32_t* __restrict__ reach, size_t n, float nodata,
                                                               float* __restrict__ out) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = ANG[i];
    const bool nd = is_nodata_f(a, nodata);
    out[i] = (reach[i] == 1) ? (nd ? ANG_SINK : a) : (nd ? a : ANG_OUTSIDE);
}

__device__ __forceinline__ float ld_ready(const float* p) {
    float v = ld_agent(p);
    for (unsigned spins = 0; __float_as_uint(v) == DINF_PENDING_BITS && spins < (1u << 20); spins++) {   // bounded: a NaN result, never a hang
        __builtin_amdgcn_s_sleep(1);
        v = ld_agent(p);
    }
    return v;
}
__global__ __launch_bounds__(256) void dinf_finish_kernel(float* __restrict__ OUT, size_t first, size_t n, float out_nodata) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < first + n && __float_as_uint(OUT[i]) == DINF_PENDING_BITS) OUT[i] = out_nodata;   // never evaluated (cycle, or upstream of one)
}

// ---- flow algebra ---------------------------------------------------------------------------------
struct AreaAlg {   // src/areadinf.cpp:187-217
    const float* W;
    __device__ __forceinline__ float evaluate(const float* __restrict__ ANG, const RowProp* __restrict__ rows, float* __restrict__ OUT,
                                              int nx, int x, int y, size_t idx, unsigned inf, int contcheck) const {
        float areares = 0.f;
        bool con = (inf & 0x100u) != 0u;
        // all contributor values are requested before the first one is used: one memory round trip per cell, not one per contributor
        float vk[8], ak[8];
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            vk[k - 1] = 0.f; ak[k - 1] = 0.f;
            if (!((inf >> (k - 1)) & 1u)) continue;
            const size_t n = size_t(y + d2(k)) * size_t(nx) + size_t(x + d1(k));
            ak[k - 1] = ANG[n];
            vk[k - 1] = ld_agent(&OUT[n]);
        }
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (!((inf >> (k - 1)) & 1u)) continue;
            const int yn = y + d2(k);
            const double p = prop_dev(ak[k - 1], (k + 4) % 8, rows[yn].a2);
            float v = vk[k - 1];
            if (__float_as_uint(v) == DINF_PENDING_BITS) v = ld_ready(&OUT[size_t(yn) * size_t(nx) + size_t(x + d1(k))]);
            if (is_nodata_f(v, TDX_AREA_NODATA)) con = true;
            else areares = (float)(areares + p * v);
        }
        if (W) areares = areares + W[idx];
        else areares = (float)(areares + rows[y].dx);
        return (con && contcheck == 1) ? TDX_AREA_NODATA : areares;
    }
};

struct DecayAlg {   // src/dinfdecayaccum.cpp:213-245
    const float* W;
    const float* DM;
    float dm_nodata;
    __device__ __forceinline__ float evaluate(const float* __restrict__ ANG, const RowProp* __restrict__ rows, float* __restrict__ OUT,
                                              int nx, int x, int y, size_t idx, unsigned inf, int contcheck) const {
        float acc = W ? W[idx] : (float)rows[y].dx;
        bool con = (inf & 0x100u) != 0u;
        float vk[8], ak[8], dk[8];
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            vk[k - 1] = 0.f; ak[k - 1] = 0.f; dk[k - 1] = 0.f;
            if (!((inf >> (k - 1)) & 1u)) continue;
            const size_t n = size_t(y + d2(k)) * size_t(nx) + size_t(x + d1(k));
            ak[k - 1] = ANG[n];
            dk[k - 1] = DM[n];
            vk[k - 1] = ld_agent(&OUT[n]);
        }
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (!((inf >> (k - 1)) & 1u)) continue;
            const int yn = y + d2(k);
            const double p = prop_dev(ak[k - 1], (k + 4) % 8, rows[yn].a2);
            float area = vk[k - 1];
            if (__float_as_uint(area) == DINF_PENDING_BITS) area = ld_ready(&OUT[size_t(yn) * size_t(nx) + size_t(x + d1(k))]);
            const float dm = dk[k - 1];
            if (is_nodata_f(area, TDX_ANG_NODATA) || is_nodata_f(dm, dm_nodata)) con = true;
            else acc = acc + (float)(dm * area * p);   // (dm*area) in float, times p in double
        }
        return (con && contcheck == 1) ? TDX_ANG_NODATA : acc;
    }
};

// walk from `start` (a ready cell) downstream while this lane keeps being the last contributor
template <class Alg>
__device__ __forceinline__ unsigned long long dinf_walk(Alg alg, const float* __restrict__ ANG, const RowProp* __restrict__ rows, int nx, int y_own0,
                                                        int y_own1, const uint16_t* __restrict__ info, int contcheck, int32_t* __restrict__ cnt,
                                                        float* __restrict__ OUT, uint32_t* __restrict__ ovf, unsigned long long* __restrict__ ovf_count,
                                                        unsigned long long ovf_cap, size_t start) {
    uint32_t stack[WALK_STACK];
    int sp = 0;
    unsigned long long done = 0;
    size_t idx = start;
    bool go = true;
    float ang = ANG[idx];
    unsigned inf = info[idx];
    while (go) {
        const int x = int(idx % size_t(nx)), y = int(idx / size_t(nx));
        const double a2 = rows[y].a2;
        const float v = alg.evaluate(ANG, rows, OUT, nx, x, y, idx, inf, contcheck);
        st_agent(&OUT[idx], v);   // not waited for: see DINF_PENDING_BITS
        cnt[idx] = CNT_DONE;      // nobody decrements an evaluated cell any more
        done++;
        go = false;
        // prop(ang, k) can be positive only for the two directions that bracket the angle (src/commonLib.cpp:83-88):
        // sector i = number of aref[1..8] that are <= ang; candidates k = i and i % 8 + 1
        const int s1 = dinf_sector(ang, a2);
        size_t tn[2];
        bool tv[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int k = (t == 0) ? s1 : (s1 % 8 + 1);
            const int xn = x + d1(k), yn = y + d2(k);
            // a target in a neighbour rank's row is released there, once this cell shows up as done in its halo row
            tv[t] = prop_dev(ang, k, a2) > 0.0 && xn >= 0 && xn < nx && yn >= y_own0 && yn < y_own1;
            tn[t] = tv[t] ? size_t(yn) * size_t(nx) + size_t(xn) : 0;
        }
        // both decrements are in flight together: one atomic round trip per cell, not one per downslope neighbour.  Release / acquire:
        // the value stored above is visible to whoever observes the decrement, and the lane that takes a counter to zero sees the values of
        // every other contributor (the hand-over of the HIP memory model; the re-read of a pending pattern below stays as a second line)
        int32_t old[2] = {0, 0};
        if (tv[0]) old[0] = __hip_atomic_fetch_sub(&cnt[tn[0]], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (tv[1]) old[1] = __hip_atomic_fetch_sub(&cnt[tn[1]], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        // what the next hop needs from its cell (read-only during the sweep) travels with the atomics, not after them
        float pang[2] = {0.f, 0.f};
        unsigned pinf[2] = {0u, 0u};
        if (tv[0]) { pang[0] = ANG[tn[0]]; pinf[0] = info[tn[0]]; }
        if (tv[1]) { pang[1] = ANG[tn[1]]; pinf[1] = info[tn[1]]; }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (tv[t] && old[t] == 1) {
                const size_t n = tn[t];
                if (!go) { idx = n; ang = pang[t]; inf = pinf[t]; go = true; }
                else if (sp < WALK_STACK) stack[sp++] = uint32_t(n);
                else {
                    const unsigned long long slot = atomicAdd(ovf_count, 1ull);
                    if (slot < ovf_cap) ovf[slot] = uint32_t(n);
                }
            }
        }
        if (!go && sp > 0) { idx = stack[--sp]; ang = ANG[idx]; inf = info[idx]; go = true; }
    }
    return done;
}

template <class Alg>
__global__ __launch_bounds__(256) void dinf_walk_kernel(Alg alg, const float* __restrict__ ANG, const RowProp* __restrict__ rows, int nx, int y_own0,
                                                        int y_own1, const uint16_t* __restrict__ info, int contcheck, int32_t* __restrict__ cnt, float* __restrict__ OUT,
                                                        uint32_t* __restrict__ ovf, unsigned long long* __restrict__ ovf_count,
                                                        unsigned long long ovf_cap) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = y_own0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= y_own1) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    if (cnt[idx] != CNT_SOURCE) return;
    dinf_walk(alg, ANG, rows, nx, y_own0, y_own1, info, contcheck, cnt, OUT, ovf, ovf_count, ovf_cap, idx);
}

template <class Alg>
__global__ __launch_bounds__(256) void dinf_walk_list_kernel(Alg alg, const float* __restrict__ ANG, const RowProp* __restrict__ rows, int nx, int y_own0,
                                                             int y_own1, const uint16_t* __restrict__ info, int contcheck, int32_t* __restrict__ cnt, float* __restrict__ OUT,
                                                             const uint32_t* __restrict__ list, unsigned long long nlist,
                                                             uint32_t* __restrict__ ovf, unsigned long long* __restrict__ ovf_count,
                                                             unsigned long long ovf_cap) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nlist) return;
    dinf_walk(alg, ANG, rows, nx, y_own0, y_own1, info, contcheck, cnt, OUT, ovf, ovf_count, ovf_cap, size_t(list[q]));
}

// A halo row after an exchange: cells that the neighbouring rank has evaluated since the last look release the owned
// cells they drain into (the role of addBorders() + the queue refill of src/areadinf.cpp:241-262).
template <class Alg>
__global__ __launch_bounds__(256) void dinf_halo_kernel(Alg alg, const float* __restrict__ ANG, const RowProp* __restrict__ rows, int nx, int y_own0,
                                                        int y_own1, const uint16_t* __restrict__ info, int contcheck, int32_t* __restrict__ cnt,
                                                        float* __restrict__ OUT, int yh, const float* __restrict__ recv_out,
                                                        const int32_t* __restrict__ recv_cnt, uint32_t* __restrict__ ovf,
                                                        unsigned long long* __restrict__ ovf_count, unsigned long long ovf_cap,
                                                        unsigned long long* __restrict__ nchanged) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    bool ch = false;
    if (x < nx) {
        const size_t h = size_t(yh) * size_t(nx) + size_t(x);
        if (recv_cnt[x] == CNT_DONE && cnt[h] != CNT_DONE) {
            ch = true;
            cnt[h] = CNT_DONE;
            st_agent(&OUT[h], recv_out[x]);
            drain_stores();
            const float ang = ANG[h];
            const double a2 = rows[yh].a2;
            const int s1 = dinf_sector(ang, a2);
            for (int t = 0; t < 2; t++) {
                const int k = (t == 0) ? s1 : (s1 % 8 + 1);
                const int xn = x + d1(k), yn = yh + d2(k);
                if (prop_dev(ang, k, a2) > 0.0 && xn >= 0 && xn < nx && yn >= y_own0 && yn < y_own1) {
                    const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
                    const int32_t old = __hip_atomic_fetch_sub(&cnt[n], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == 1) dinf_walk(alg, ANG, rows, nx, y_own0, y_own1, info, contcheck, cnt, OUT, ovf, ovf_count, ovf_cap, n);
                }
            }
        }
    }
    const unsigned long long m = __ballot(ch);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(nchanged, (unsigned long long)__popcll(m));
}


// =====================================================================================================================
// Tile dependency sweep (default path).  The walk above pays two device-scope memory round trips (~2.5 us) per cell of the
// LONGEST flow path.  A cell's value depends only on its contributors' values (the k-ordered fold of the reference), never
// on the schedule, so the same bits come out of any evaluation order that respects the dependencies.  Here:
//   * a 256-thread workgroup stages a 64 x 64 tile + one ring of result, angle (decay multiplier, weight) and the per-cell
//     inflow / target bits in LDS; a cell is READY when none of its contributors still shows the "pending" pattern - counted
//     from the staged values, no global counters at all;
//   * lanes walk downstream from the ready cells inside LDS (evaluate by pulling contributors in k order, publish in LDS,
//     decrement the packed byte counters of the <= 2 targets with one returning LDS atomic each, continue into a target
//     that became ready; a second ready target goes to a small LDS queue that the workgroup drains in phases) - a hop costs
//     LDS latency plus the fp64 proportions, not HBM round trips;
//   * evaluated cells are written back; a tile whose finished cells drain into a neighbouring tile raises that tile's
//     activation flag for the next ROUND (the schedule of tile_relax.hpp: lists, flags, device-chained launches), which
//     re-counts its perimeter from the new ring values.  Rounds = tile crossings of the longest dependency chain.
// Cells that can never be evaluated (cycles, downstream of cycles) stay pending and become nodata in dinf_finish_kernel,
// like the reference's never-queued cells.
// =====================================================================================================================
namespace dsweep {
constexpr int QCAP = 512;

// Per cell, once (streaming, all rows of the array): everything the sweep needs to evaluate the cell and to find its targets
// without a proportion of its own -
//   info  [0:8) neighbour k drains into the cell (the test of initNeighborDinfup, src/commonLib.cpp:99-131: `float p > 0`)
//         [8]   a neighbour is missing (outside the raster or nodata: the contamination test of src/areadinf.cpp:196-199)
//         [9:12) s1 - 1, [12] / [13] the cell sends flow to neighbour s1 / s1 % 8 + 1 (prop() can be positive only for the two
//                directions that bracket the angle, src/commonLib.cpp:83-88)
//         [16:24) contributor k reaches this cell through ITS second target (selects which of its two proportions applies)
//   P     the two proportions of the cell's own outflow as doubles (the fp64 divisions of prop() leave the serial path)
// and the result array starts as "pending" on participating owned cells.
// Two passes (dinf_prop.hpp): the cell's own outflow (two fp64 divisions) -> P and a code byte; then a byte stencil -> info.
__global__ __launch_bounds__(256) void setup_out_kernel(const float* __restrict__ ANG, size_t n, int nx, int y_own0, int y_own1, float nodata,
                                                        const RowProp* __restrict__ rows, uint8_t* __restrict__ code, double2* __restrict__ P,
                                                        float* __restrict__ OUT, float out_nodata) {
    const size_t idx = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n) return;
    const int y = int(idx / size_t(nx));
    const float ang = ANG[idx];
    const bool nd = is_nodata_f(ang, nodata), part = !(nd || ang == ANG_OUTSIDE);
    double2 pp;
    code[idx] = uint8_t(dinf_code(ang, nd, part, rows[y].a2, &pp.x, &pp.y));
    P[idx] = pp;
    if (y >= y_own0 && y < y_own1) OUT[idx] = part ? __uint_as_float(DINF_PENDING_BITS) : out_nodata;
}
// four cells per thread (a raster whose width is a multiple of 4 on 16-byte boundaries): one 16-byte load, four 16-byte proportion stores, a 4-byte code store, one 16-byte
// store of the pending pattern - a quarter of the memory instructions of the one-cell form
__global__ __launch_bounds__(256) void setup_out4_kernel(const float* __restrict__ ANG, size_t n, int nx, int y_own0, int y_own1, float nodata,
                                                         const RowProp* __restrict__ rows, uint8_t* __restrict__ code, double2* __restrict__ P,
                                                         float* __restrict__ OUT, float out_nodata) {
    const size_t idx = (size_t(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (idx >= n) return;
    const int y = int(idx / size_t(nx));   // (nx % 4 == 0: the four cells are in one row)
    const float4 a4 = *reinterpret_cast<const float4*>(ANG + idx);
    const float a[4] = {a4.x, a4.y, a4.z, a4.w};
    const double a2 = rows[y].a2;
    unsigned cw = 0;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool nd = is_nodata_f(a[i], nodata), part = !(nd || a[i] == ANG_OUTSIDE);
        double2 pp;
        cw |= (unsigned(dinf_code(a[i], nd, part, a2, &pp.x, &pp.y)) & 0xFFu) << (8 * i);
        P[idx + i] = pp;
        o[i] = part ? __uint_as_float(DINF_PENDING_BITS) : out_nodata;
    }
    *reinterpret_cast<uint32_t*>(code + idx) = cw;
    if (y >= y_own0 && y < y_own1) *reinterpret_cast<float4*>(OUT + idx) = make_float4(o[0], o[1], o[2], o[3]);
}
// per cell: which neighbours send to it (and whether as their first or second target), whether a neighbour is missing, its own targets - from the code bytes of its 3 x 3
// window.  62-column window like the streaming passes of d8flowdir.hip: a lane loads ONE code byte per window row (18 for 16 output rows, all before the first use), the
// columns beside it are the neighbouring lanes' (the per-cell form loaded nine bytes per cell: 4.8 ms at 32768^2).
constexpr int SIN_COLS = 62, SIN_ROWS = 16;
__global__ __launch_bounds__(256) void setup_in_kernel(const uint8_t* __restrict__ code, int nx, int ny, uint32_t* __restrict__ info, int nbx, int xmap) {
    using tilek::lane_left;
    using tilek::lane_right;
    const int bx = tdxk::xcd_block_x(nbx, xmap);
    if (bx < 0) return;
    const int lx = threadIdx.x & 63;
    const int x = bx * SIN_COLS - 1 + lx;
    const int ybase = __builtin_amdgcn_readfirstlane(int(blockIdx.y) * (4 * SIN_ROWS) + int(threadIdx.x >> 6) * SIN_ROWS);
    const bool mine = lx >= 1 && lx <= SIN_COLS && x < nx;
    const bool inx = x >= 0 && x < nx;
    const int xc = x < 0 ? 0 : (x >= nx ? nx - 1 : x);
    int cw[SIN_ROWS + 2];
#pragma unroll
    for (int j = 0; j < SIN_ROWS + 2; j++) {
        const int y = ybase - 1 + j, yc = y < 0 ? 0 : (y >= ny ? ny - 1 : y);
        cw[j] = code[size_t(yc) * size_t(nx) + size_t(xc)];
    }
#pragma unroll
    for (int j = 0; j < SIN_ROWS + 2; j++) {
        const int y = ybase - 1 + j;
        if (!inx || y < 0 || y >= ny) cw[j] = DINF_CODE_NODATA;   // outside the raster
    }
#pragma unroll
    for (int r = 0; r < SIN_ROWS; r++) {
        const int y = ybase + r;
        // window in the order of the neighbour index k = 1 .. 8 (E NE N NW W SW S SE)
        const int n1 = cw[r], c1 = cw[r + 1], s1 = cw[r + 2];
        unsigned c[9];
        c[1] = unsigned(lane_right(c1, int(DINF_CODE_NODATA)));
        c[2] = unsigned(lane_right(n1, int(DINF_CODE_NODATA)));
        c[3] = unsigned(n1);
        c[4] = unsigned(lane_left(n1, int(DINF_CODE_NODATA)));
        c[5] = unsigned(lane_left(c1, int(DINF_CODE_NODATA)));
        c[6] = unsigned(lane_left(s1, int(DINF_CODE_NODATA)));
        c[7] = unsigned(s1);
        c[8] = unsigned(lane_right(s1, int(DINF_CODE_NODATA)));
        if (!(mine && y < ny)) continue;
        const unsigned own = unsigned(c1);
        unsigned inf = 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (dinf_code_missing(c[k])) inf |= 0x100u;          // (a sink on a cell without angle: missing for the contamination test, sends nothing)
            if (c[k] == DINF_CODE_NODATA) continue;
            const int kk = (k + 4) % 8;                                   // direction from the neighbour to this cell
            const int via = dinf_code_sends(c[k], kk == 0 ? 8 : kk);
            if (via) inf |= 1u << (k - 1);
            if (via == 2) inf |= 1u << (16 + k - 1);                       // not its first target: its second
        }
        if (own != DINF_CODE_NODATA && (own & DINF_CODE_PART)) {
            inf |= (own & 7u) << 9;
            if (own & DINF_CODE_P1) inf |= 1u << 12;
            if (own & DINF_CODE_P2) inf |= 1u << 13;
        }
        info[size_t(y) * size_t(nx) + size_t(x)] = inf;
    }
}

__device__ __forceinline__ bool pending(float v) { return __float_as_uint(v) == DINF_PENDING_BITS; }

// The neighbourhood of a cell as registers: all 8 result values and proportion pairs are requested unconditionally and
// together (ONE LDS latency); a read inside a data-dependent branch gets its own basic block and its own wait - eight of
// them per cell were most of a hop's time.
struct Nbr8 { float v[9]; double2 p[9]; float dm[9]; };
template <class L, bool HAS_DM, int LH>
__device__ __forceinline__ void load_nbrs(const L& S, int cl, Nbr8& nb) {
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        const int n = cl + d2(k) * LH + d1(k);
        nb.v[k] = S.out[n];
        nb.p[k] = S.p[n];
        if (HAS_DM) nb.dm[k] = S.dm[n];
    }
}
__device__ __forceinline__ unsigned pending_bits(const Nbr8& nb) {
    unsigned b = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) b |= pending(nb.v[k]) ? 1u << (k - 1) : 0u;
    return b;
}

struct AreaEval {   // src/areadinf.cpp:187-217
    // the same expression contributor by contributor (the walks fold the contributors that exist, in k order: eval_walk in dinf_sweep_tile.inc)
    __device__ __forceinline__ float begin(float, double, bool) const { return 0.f; }
    __device__ __forceinline__ void fold(float& areares, bool& con, float v, double p, float) const {
        if (is_nodata_f(v, TDX_AREA_NODATA)) con = true;
        else areares = (float)(areares + p * v);
    }
    __device__ __forceinline__ float end(float areares, bool con, float w, double dx, bool has_w, int contcheck) const {
        if (has_w) areares = areares + w;
        else areares = (float)(areares + dx);
        return (con && contcheck == 1) ? TDX_AREA_NODATA : areares;
    }
    __device__ __forceinline__ float eval(const Nbr8& nb, float w, double dx, unsigned inf, int contcheck, bool has_w) const {
        float areares = 0.f;
        bool con = (inf & 0x100u) != 0u;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (!((inf >> (k - 1)) & 1u)) continue;
            const double p = ((inf >> (16 + k - 1)) & 1u) ? nb.p[k].y : nb.p[k].x;
            const float v = nb.v[k];
            if (is_nodata_f(v, TDX_AREA_NODATA)) con = true;
            else areares = (float)(areares + p * v);
        }
        if (has_w) areares = areares + w;
        else areares = (float)(areares + dx);
        return (con && contcheck == 1) ? TDX_AREA_NODATA : areares;
    }
};
struct DecayEval {   // src/dinfdecayaccum.cpp:213-245
    float dm_nodata;
    __device__ __forceinline__ float begin(float w, double dx, bool has_w) const { return has_w ? w : (float)dx; }
    __device__ __forceinline__ void fold(float& acc, bool& con, float area, double p, float dm) const {
        if (is_nodata_f(area, TDX_ANG_NODATA) || is_nodata_f(dm, dm_nodata)) con = true;
        else acc = acc + (float)(dm * area * p);   // (dm*area) in float, times p in double
    }
    __device__ __forceinline__ float end(float acc, bool con, float, double, bool, int contcheck) const { return (con && contcheck == 1) ? TDX_ANG_NODATA : acc; }
    __device__ __forceinline__ float eval(const Nbr8& nb, float w, double dx, unsigned inf, int contcheck, bool has_w) const {
        float acc = has_w ? w : (float)dx;
        bool con = (inf & 0x100u) != 0u;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            if (!((inf >> (k - 1)) & 1u)) continue;
            const double p = ((inf >> (16 + k - 1)) & 1u) ? nb.p[k].y : nb.p[k].x;
            const float area = nb.v[k], dm = nb.dm[k];
            if (is_nodata_f(area, TDX_ANG_NODATA) || is_nodata_f(dm, dm_nodata)) con = true;
            else acc = acc + (float)(dm * area * p);   // (dm*area) in float, times p in double
        }
        return (con && contcheck == 1) ? TDX_ANG_NODATA : acc;
    }
};

// Verifier of the D-infinity sweep (TDX_SWEEP_VERIFY=1; the contract is spelled out in d8_sweep.hpp): one streaming pass over the
// quiescent result that gathers every owned cell's neighbourhood straight from global memory - no tiles, no LDS, nothing shared with
// the sweep but the policy's expression - and checks that a pending cell has a pending contributor and that an evaluated cell is,
// bit for bit, what its contributors' final values give.  out[0] cells checked, out[1] mismatches, out[2] first mismatching index.
template <class Eval, bool HAS_W, bool HAS_DM>
__global__ __launch_bounds__(256) void verify_kernel(Eval ev, const float* __restrict__ ANG, float ang_nodata, int nx, int ny, int y_own0, int y_own1,
                                                     const double2* __restrict__ P, const float* __restrict__ W, const float* __restrict__ DM,
                                                     const uint32_t* __restrict__ INFO, const RowProp* __restrict__ rows, const float* __restrict__ OUT,
                                                     float out_nodata, int contcheck, unsigned long long* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = y_own0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    bool checked = false, wrong = false;
    size_t idx = 0;
    if (x < nx && y < y_own1) {
        idx = size_t(y) * size_t(nx) + size_t(x);
        const float ang = ANG[idx], me = OUT[idx];
        const bool part = !(is_nodata_f(ang, ang_nodata) || ang == ANG_OUTSIDE);
        checked = true;
        if (!part) wrong = __float_as_uint(me) != __float_as_uint(out_nodata);
        else {
            const unsigned inf = INFO[idx];
            Nbr8 nb;
#pragma unroll
            for (int k = 1; k <= 8; k++) {
                const int xn = x + d1(k), yn = y + d2(k);
                const bool in = xn >= 0 && xn < nx && yn >= 0 && yn < ny;
                const size_t n = size_t(in ? yn : y) * size_t(nx) + size_t(in ? xn : x);
                nb.v[k] = in ? OUT[n] : out_nodata;
                nb.p[k] = in ? P[n] : make_double2(0., 0.);
                nb.dm[k] = (HAS_DM && in) ? DM[n] : 0.f;
            }
            const unsigned pb = pending_bits(nb) & inf & 0xFFu;
            if (pending(me)) wrong = pb == 0u;        // ready, never evaluated
            else if (pb != 0u) wrong = true;          // evaluated ahead of a contributor
            else wrong = __float_as_uint(ev.eval(nb, HAS_W ? W[idx] : 0.f, rows[y].dx, inf, contcheck, HAS_W)) != __float_as_uint(me);
        }
    }
    const unsigned long long bc = __ballot(checked), bw = __ballot(wrong);
    if (bw) atomicMin(out + 2, wrong ? (unsigned long long)idx : ~0ull);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 0, (unsigned long long)__popcll(bc));
        if (bw) atomicAdd(out + 1, (unsigned long long)__popcll(bw));
    }
}

// activation flags between the two tile geometries (a 64 x 64 tile = 2 x 2 tiles of 32 x 32, sh = 1, or 4 x 4 tiles of 16 x 16, sh = 2)
static __global__ __launch_bounds__(256) void flags_down_kernel(const uint32_t* __restrict__ f64, int tiles_x64, uint32_t* __restrict__ f32, int tiles_x32,
                                                                int tiles_y32, int sh) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles_x32 * tiles_y32) return;
    const int tx = t % tiles_x32, ty = t / tiles_x32;
    f32[t] = f64[(ty >> sh) * tiles_x64 + (tx >> sh)] ? tilek::FLAG_FULL : 0u;
}
static __global__ __launch_bounds__(256) void flags_up_kernel(uint32_t* __restrict__ f32, int tiles_x32, int tiles_y32, uint32_t* __restrict__ f64, int tiles_x64, int sh) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles_x32 * tiles_y32) return;
    if (f32[t]) {
        f32[t] = 0u;
        const int tx = t % tiles_x32, ty = t / tiles_x32;
        f64[(ty >> sh) * tiles_x64 + (tx >> sh)] = tilek::FLAG_HALO;   // walks only: the bulk sweeps have run
    }
}
}  // namespace dsweep

#define DSWEEP_NS dsweep64
#define DSWEEP_TS 64
#define DSWEEP_NT 1024
#include "dinf_sweep_tile.inc"
#undef DSWEEP_NS
#undef DSWEEP_TS
#undef DSWEEP_NT
#define DSWEEP_NS dsweep32
#define DSWEEP_TS 32
#define DSWEEP_NT 256
#include "dinf_sweep_tile.inc"
#undef DSWEEP_NS
#undef DSWEEP_TS
#undef DSWEEP_NT
// 16 x 16 tiles, ONE wave per tile (no workgroup barrier in the hop loop, 10 KB of LDS: fifteen tiles per CU) - the variant VERDICT r04 asked to be
// measured for the bulk rounds (TDX_DINF_BULK_TILE=16; docs/experiments_r05.md has what it did)
#define DSWEEP_NS dsweep16
#define DSWEEP_TS 16
#define DSWEEP_NT 64
#include "dinf_sweep_tile.inc"
#undef DSWEEP_NS
#undef DSWEEP_TS
#undef DSWEEP_NT



template <class Alg>
int run_dinf_accum(tdx_context* ctx, Alg alg, const Strip& st, float* d_ang, float ang_nodata, const double* dxc, const double* dyc, int contcheck,
                   const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_out, float out_nodata, float* d_dm, float dm_nodata,
                   tdx_stats* stats) {
    if (n_outlets > 0 && (!outlet_x || !outlet_y)) return tdx_fail(ctx, TDX_ERR_ARG, "outlets missing");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = st.nx, iny = st.ny_arr;
    const size_t n = size_t(inx) * size_t(iny);
    std::vector<RowProp> rows;
    rows.resize(size_t(iny));
    for (int j = 0; j < iny; j++) { rows[size_t(j)].a2 = atan2(dyc[j], dxc[j]); rows[size_t(j)].dx = dxc[j]; }
    RowProp* d_rows = static_cast<RowProp*>(ctx->scratch(TDX_S_J, rows.size() * sizeof(RowProp)));
    // default: the tile dependency sweep (LDS tiles on the round schedule, no device-scope atomics).  TDX_DINF_WALK=1: the
    // atomic pull walk instead (A/B hook; tests/test_gpu_dinf.py checks that both give the same bits)
    const bool use_walk = getenv("TDX_DINF_WALK") != nullptr;
    int32_t* cnt = use_walk ? static_cast<int32_t*>(ctx->scratch(TDX_S_A, n * 4)) : reinterpret_cast<int32_t*>(ctx->d_mail);
    uint16_t* info = use_walk ? static_cast<uint16_t*>(ctx->scratch(TDX_S_I, n * 2)) : reinterpret_cast<uint16_t*>(ctx->d_mail);
    const unsigned long long ovf_cap = n / 4 + 1024;
    uint32_t* ovfa = static_cast<uint32_t*>(ctx->scratch(TDX_S_G, size_t(ovf_cap) * 4));
    uint32_t* ovfb = static_cast<uint32_t*>(ctx->scratch(TDX_S_H, size_t(ovf_cap) * 4));
    float* recvbuf = static_cast<float*>(ctx->scratch(TDX_S_K, size_t(inx) * 4 * 4));   // received OUT rows (up, down) and counter rows (up, down)
    if (!d_rows || !cnt || !info || !ovfa || !ovfb || !recvbuf) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(RowProp), hipMemcpyHostToDevice, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    const dim3 grid2d((inx + 63) / 64, (st.y1 - st.y0 + 3) / 4);

    ctx->begin_call(stats);
    strip_mark(ctx, st, "areadinf / dinfdecayaccum");
    int rc = strip_exchange<float>(ctx, st, d_ang, ang_nodata);   // angles of the neighbours' boundary rows
    if (rc != TDX_OK) return rc;
    if (d_dm) { rc = strip_exchange<float>(ctx, st, d_dm, dm_nodata); if (rc != TDX_OK) return rc; }
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    float* ang_use = d_ang;
    if (n_outlets >= 0) {
        ctx->phase = "outlets' closure";
        rc = dinf_outlet_recode(ctx, st, d_ang, ang_nodata, d_rows, outlet_x, outlet_y, n_outlets, &ang_use, stats);
        if (rc != TDX_OK) return rc;
        TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    }
    ctx->phase = "set-up";
    uint32_t* info32 = nullptr;
    double2* d_P = nullptr;
    if (!use_walk) {
        info32 = static_cast<uint32_t*>(ctx->scratch(TDX_S_I, n * 4));
        d_P = static_cast<double2*>(ctx->scratch(TDX_S_A, n * 16));
        if (!info32 || !d_P) return TDX_ERR_NOMEM;
    }
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        if (use_walk)
            hipLaunchKernelGGL(dinf_setup_kernel, grid2d, dim3(256), 0, s, ang_use, inx, iny, st.y0, st.y1, ang_nodata, d_rows, info, cnt, d_out, out_nodata);
        else
        {
            uint8_t* code = static_cast<uint8_t*>(ctx->scratch(TDX_S_B, n));
            if (!code) return TDX_ERR_NOMEM;
            if (inx % 4 == 0 && ((reinterpret_cast<uintptr_t>(ang_use) | reinterpret_cast<uintptr_t>(d_out)) & 15u) == 0)
                hipLaunchKernelGGL(dsweep::setup_out4_kernel, dim3(tdx_blocks_for(n / 4, 256)), dim3(256), 0, s, ang_use, n, inx, st.y0, st.y1, ang_nodata, d_rows, code, d_P,
                                   d_out, out_nodata);
            else
                hipLaunchKernelGGL(dsweep::setup_out_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, ang_use, n, inx, st.y0, st.y1, ang_nodata, d_rows, code, d_P, d_out,
                                   out_nodata);
            const int sin_nbx = (inx + dsweep::SIN_COLS - 1) / dsweep::SIN_COLS;
            hipLaunchKernelGGL(dsweep::setup_in_kernel, dim3(tdx_xcd_grid_x(unsigned(sin_nbx)), (iny + 4 * dsweep::SIN_ROWS - 1) / (4 * dsweep::SIN_ROWS)), dim3(256), 0, s, code, inx, iny,
                               info32, sin_nbx, tdx_xcd_map() ? 1 : 0);
        }
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    // halo rows of the result (and of the walk's counters) start as "not evaluated"
    rc = strip_exchange<float>(ctx, st, d_out, out_nodata);
    if (rc != TDX_OK) return rc;
    int64_t rounds = 0, outer = 0;
    if (!use_walk) {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        // two tile geometries over the same arrays (the state of the sweep is the result raster alone): 32 x 32 tiles for the bulk
        // rounds, 64 x 64 tiles for the tail (dinf_sweep_tile.inc)
        tilek::TileGeom geom = tilek::make_geom(inx, iny, st.y0, st.y1);
        geom.max_sweeps = dsweep64::BULK_SWEEPS;
        // edge of the bulk rounds' tiles: 16 x 16 one-wave tiles gain 5 % at 32768^2 and nothing at 16384^2 (docs/experiments_r05.md), so they take the large
        // rasters (>= 2^29 cells per strip) of AreaDinf; DinfDecayAccum stays on 32 x 32 (its decay rows in LDS); TDX_DINF_BULK_TILE=16|32 decides otherwise
        static const int bulk_env = getenv("TDX_DINF_BULK_TILE") ? atoi(getenv("TDX_DINF_BULK_TILE")) : 0;
        const int bulk_ts = bulk_env == 16 ? 16 : (bulk_env == 32 ? 32 : ((std::is_same<Alg, AreaAlg>::value && n >= (size_t(1) << 29)) ? 16 : 32));
        const int bulk_sh = bulk_ts == 16 ? 2 : 1;
        tilek::TileGeom geom32 = geom;
        geom32.tiles_x = (inx + bulk_ts - 1) / bulk_ts; geom32.tiles_y = (iny + bulk_ts - 1) / bulk_ts;
        // lockstep sweeps of a fresh tile before the walks take over (dinf_sweep_tile.inc)
        static const int bulk_sweeps = getenv("TDX_DINF_BULK_SWEEPS") ? std::max(0, atoi(getenv("TDX_DINF_BULK_SWEEPS"))) : dsweep32::BULK_SWEEPS;
        geom32.max_sweeps = bulk_sweeps;
        const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y), ntiles32 = size_t(geom32.tiles_x) * size_t(geom32.tiles_y);
        uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));
        uint32_t* flags32 = static_cast<uint32_t*>(ctx->scratch(TDX_S_G, ntiles32 * 4 * (1 + tilek::SCHED_LIST_WORDS)));
        unsigned long long* counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
        if (!flags || !flags32 || !counts) return TDX_ERR_NOMEM;
        const tilek::Sched sched{flags, flags + ntiles, counts}, sched32{flags32, flags32 + ntiles32, counts};
        const float* d_w = alg.W;
        static const bool dbg_rounds = getenv("TDX_DEBUG_ROUNDS") != nullptr;   // active tiles per round on stderr
        static const bool dbg_cycles = dbg_rounds && atoi(getenv("TDX_DEBUG_ROUNDS")) == 1;
        // the bulk phase ends when a round has at most this many active 32 x 32 tiles (0: no bulk phase)
        static const unsigned long long bulk_until = getenv("TDX_DINF_BULK_UNTIL") ? strtoull(getenv("TDX_DINF_BULK_UNTIL"), nullptr, 10) : 400ull;
        unsigned long long* dbg = dbg_cycles ? reinterpret_cast<unsigned long long*>(ctx->d_mail) + TDX_MAIL_DBG_SWEEP : nullptr;
        auto launch = [&](bool small, const tilek::TileGeom& gg, unsigned grid, hipStream_t ls, const uint32_t* list, unsigned long long* count, uint32_t* fcur,
                          uint32_t* fnext, uint32_t* lnext, unsigned pull_max) {
#define TDX_DSWEEP_LAUNCH(NS)                                                                                                                                   \
    if constexpr (std::is_same<Alg, AreaAlg>::value) {                                                                                                          \
        if (d_w)                                                                                                                                                \
            hipLaunchKernelGGL((NS::sweep_kernel<dsweep::AreaEval, true, false>), dim3(grid), dim3(NS::NT), 0, ls, dsweep::AreaEval{}, gg, list, count, fcur,  \
                               fnext, lnext, pull_max, d_P, d_w, nullptr, info32, d_rows, d_out, out_nodata, contcheck, dbg);                                   \
        else                                                                                                                                                    \
            hipLaunchKernelGGL((NS::sweep_kernel<dsweep::AreaEval, false, false>), dim3(grid), dim3(NS::NT), 0, ls, dsweep::AreaEval{}, gg, list, count, fcur, \
                               fnext, lnext, pull_max, d_P, nullptr, nullptr, info32, d_rows, d_out, out_nodata, contcheck, dbg);                               \
    } else {                                                                                                                                                    \
        if (d_w)                                                                                                                                                \
            hipLaunchKernelGGL((NS::sweep_kernel<dsweep::DecayEval, true, true>), dim3(grid), dim3(NS::NT), 0, ls, dsweep::DecayEval{alg.dm_nodata}, gg, list, \
                               count, fcur, fnext, lnext, pull_max, d_P, d_w, alg.DM, info32, d_rows, d_out, out_nodata, contcheck, dbg);                       \
        else                                                                                                                                                    \
            hipLaunchKernelGGL((NS::sweep_kernel<dsweep::DecayEval, false, true>), dim3(grid), dim3(NS::NT), 0, ls, dsweep::DecayEval{alg.dm_nodata}, gg, list, \
                               count, fcur, fnext, lnext, pull_max, d_P, nullptr, alg.DM, info32, d_rows, d_out, out_nodata, contcheck, dbg);                   \
    }
            if (small && bulk_ts == 16) { TDX_DSWEEP_LAUNCH(dsweep16) } else if (small) { TDX_DSWEEP_LAUNCH(dsweep32) } else { TDX_DSWEEP_LAUNCH(dsweep64) }
#undef TDX_DSWEEP_LAUNCH
        };
        int64_t launches = 0;
        // runs one geometry until no tile is active, or (stop_at > 0) until a round has at most stop_at active tiles; returns whether
        // tiles are still active (their flags of the next round are then in run.flags_of(run.parity))
        // ... or (max_rounds > 0) until that many rounds have run: a multi-strip tail exchanges its boundary rows every few rounds instead of running
        // every strip to its local fixed point first
        auto run_rounds = [&](bool small, const tilek::TileGeom& gg, const tilek::Sched& sc, unsigned long long stop_at, bool* active_left, int* parity_out,
                              int max_rounds = 0) -> int {
            RoundRunner<flatk::LevelOp> run(ctx, s, flatk::LevelOp{nullptr, nullptr}, gg, sc, ctx->h_mail + TDX_MAIL_RUN_A, nullptr);
            if (max_rounds > 0) { run.batch = max_rounds; run.batch_max = max_rounds; }
            if (small) { run.grid_full = unsigned(std::min(run.ntiles, (bulk_ts == 16 ? 48 : 16) * ctx->num_cus)); run.grid_small = unsigned(std::min(run.ntiles, 4 * ctx->num_cus)); }
            run.custom_launch = [&](const tilek::TileGeom& rg, unsigned grid, hipStream_t ls, const uint32_t* list, unsigned long long* count, uint32_t* fcur, uint32_t* fnext,
                                    uint32_t* lnext, unsigned pull_max) { launch(small, rg, grid, ls, list, count, fcur, fnext, lnext, pull_max); };
            run.print_counts = dbg_rounds;
            if (dbg_rounds) fprintf(stderr, "\ndinf sweep rounds(%d tiles of %d):", run.ntiles, small ? bulk_ts : 64);
            if (dbg) TDX_HIP_CHECK(ctx, hipMemset(dbg, 0, 80));
            int last_printed = -1;
            int rcl = run.start();
            if (rcl != TDX_OK) return rcl;
            *active_left = false;
            while (!run.done) {
                rcl = run.enqueue();
                if (rcl != TDX_OK) return rcl;
                TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
                run.collect();
                if (dbg) {   // per batch of rounds: average phase times (us) and hops per activation
                    TDX_HIP_CHECK(ctx, hipMemcpy(ctx->h_mail + TDX_MAIL_DBG_SWEEP, dbg, 80, hipMemcpyDeviceToHost));
                    TDX_HIP_CHECK(ctx, hipMemset(dbg, 0, 80));
                    const double na = double(ctx->h_mail[TDX_MAIL_DBG_SWEEP + 7] ? ctx->h_mail[TDX_MAIL_DBG_SWEEP + 7] : 1);
                    fprintf(stderr, "\n  [rounds %d..%lld] activations %.0f: stage %.1f scan+bulk %.1f walks %.1f writeback %.1f us; hops/act %.1f, busiest lane %.1f, phases %.2f; of the records staged %.1f %% were pending, %.1f %% were evaluated\n",
                            last_printed + 1, (long long)run.rounds - 1, na, ctx->h_mail[TDX_MAIL_DBG_SWEEP + 0] / na / 100.0, ctx->h_mail[TDX_MAIL_DBG_SWEEP + 1] / na / 100.0, ctx->h_mail[TDX_MAIL_DBG_SWEEP + 2] / na / 100.0,
                            ctx->h_mail[TDX_MAIL_DBG_SWEEP + 3] / na / 100.0, ctx->h_mail[TDX_MAIL_DBG_SWEEP + 4] / na, ctx->h_mail[TDX_MAIL_DBG_SWEEP + 5] / na, ctx->h_mail[TDX_MAIL_DBG_SWEEP + 6] / na,
                            100.0 * double(ctx->h_mail[TDX_MAIL_DBG_SWEEP + 8]) / (na * double(small ? bulk_ts * bulk_ts : 4096)), 100.0 * double(ctx->h_mail[TDX_MAIL_DBG_SWEEP + 9]) / (na * double(small ? bulk_ts * bulk_ts : 4096)));
                    last_printed = int(run.rounds) - 1;
                }
                if (!run.done && stop_at > 0 && run.last_count <= stop_at) { *active_left = true; *parity_out = run.parity; break; }
                if (!run.done && max_rounds > 0 && run.rounds >= max_rounds) { *active_left = true; *parity_out = run.parity; break; }
            }
            rounds += run.rounds;
            launches += run.launches;
            return TDX_OK;
        };
        hipLaunchKernelGGL(tilek::fill_u32_kernel, dim3(tdx_blocks_for(ntiles, 256)), dim3(256), 0, s, flags, tilek::FLAG_FULL, ntiles);   // round 0: every tile
        bool bulk = bulk_until > 0 && ntiles32 > bulk_until;
        for (;;) {
            bool left = false, bulk_just_ran = false;
            int par = 0;
            if (bulk) {
                ctx->phase = "bulk rounds";
                // bulk phase on 32 x 32 tiles: every 32-tile under an active 64-tile starts active
                hipLaunchKernelGGL(dsweep::flags_down_kernel, dim3(tdx_blocks_for(ntiles32, 256)), dim3(256), 0, s, flags, geom.tiles_x, flags32, geom32.tiles_x,
                                   geom32.tiles_y, bulk_sh);
                TDX_HIP_CHECK(ctx, hipMemsetAsync(flags, 0, ntiles * 4, s));
                rc = run_rounds(true, geom32, sched32, bulk_until, &left, &par);
                if (rc != TDX_OK) return rc;
                if (left) {   // what is still active goes on in 64 x 64 tiles
                    uint32_t* f32next = par ? sched32.list + 2 * ntiles32 : sched32.flags;
                    hipLaunchKernelGGL(dsweep::flags_up_kernel, dim3(tdx_blocks_for(ntiles32, 256)), dim3(256), 0, s, f32next, geom32.tiles_x, geom32.tiles_y, flags,
                                       geom.tiles_x, bulk_sh);
                }
                bulk = false;   // (strip re-activations are few tiles: 64 x 64)
                bulk_just_ran = true;
            } else left = true;
            // (with neighbours the bulk phase is followed by an exchange at once: the tail's first rounds then see what the neighbours' bulk phases finished,
            // and the segment trace shows the two phases apart - the step time is the same either way: 311.4 against 311.2 ms projected at BASELINE.json configs[4])
            if (left && !(bulk_just_ran && st.multi())) {
                ctx->phase = "tail rounds";
                // Multi-strip tail: at most `eager` rounds between two exchanges.  A strip that runs to its local fixed point first makes every flow path
                // that crosses a strip boundary wait for the longest chain ANYWHERE in the strip it enters - at BASELINE.json configs[4] 47 crossings x
                // ~10 ms (profiles/r05b_projection_decay.txt); with frequent exchanges the paths advance side by side, as on one GPU.
                const int eager_env = getenv("TDX_SWEEP_EAGER_ROUNDS") ? std::max(0, atoi(getenv("TDX_SWEEP_EAGER_ROUNDS"))) : 8;   // (0: local fixed points)
                const int eager = st.multi() ? eager_env : 0;
                rc = run_rounds(false, geom, sched, 0, &left, &par, eager);
                if (rc != TDX_OK) return rc;
                if (left && par)   // stopped with tiles still active: the next schedule starts from the first flag half
                    hipLaunchKernelGGL(tilek::flags_fold_kernel, dim3(tdx_blocks_for(ntiles, 256)), dim3(256), 0, s, sched.flags, sched.list + 2 * ntiles, int(ntiles));
            }
            if (!st.multi()) break;
            // the neighbours' boundary rows: cells finished there release the owned cells they drain into (addBorders() + queue
            // refill of src/areadinf.cpp:241-262); tiles that see a changed halo cell run again
            int64_t changed = 0;
            rc = strip_exchange<float>(ctx, st, d_out, out_nodata, flags, geom.tiles_x, &changed, true, left ? 1 : 0);
            if (rc != TDX_OK) return rc;
            if (changed == 0) break;
            outer++;
        }
        if (stats) stats->launches[TDX_K_ACCUM] += launches;
        if (d8sweep::verify_enabled()) {
            unsigned long long* d_ver = reinterpret_cast<unsigned long long*>(ctx->d_mail) + TDX_MAIL_VERIFY;
            const unsigned long long init[3] = {0ull, 0ull, ~0ull};
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_ver, init, sizeof init, hipMemcpyHostToDevice, s));
            TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));   // `init` is a local
#define TDX_DSWEEP_VERIFY(EV, EVOBJ, HW, HD, DMP)                                                                                                             \
    hipLaunchKernelGGL((dsweep::verify_kernel<EV, HW, HD>), grid2d, dim3(256), 0, s, EVOBJ, ang_use, ang_nodata, inx, iny, st.y0, st.y1, d_P, d_w, DMP, info32, \
                       d_rows, d_out, out_nodata, contcheck, d_ver)
            if constexpr (std::is_same<Alg, AreaAlg>::value) {
                if (d_w) TDX_DSWEEP_VERIFY(dsweep::AreaEval, dsweep::AreaEval{}, true, false, nullptr);
                else TDX_DSWEEP_VERIFY(dsweep::AreaEval, dsweep::AreaEval{}, false, false, nullptr);
            } else {
                if (d_w) TDX_DSWEEP_VERIFY(dsweep::DecayEval, dsweep::DecayEval{alg.dm_nodata}, true, true, alg.DM);
                else TDX_DSWEEP_VERIFY(dsweep::DecayEval, dsweep::DecayEval{alg.dm_nodata}, false, true, alg.DM);
            }
#undef TDX_DSWEEP_VERIFY
            rc = d8sweep::verify_report(ctx, "D-infinity dependency sweep", d_ver, inx);
            if (rc != TDX_OK) return rc;
        }
    } else {
    rc = strip_exchange<int32_t>(ctx, st, cnt, CNT_NOT_PART);
    if (rc != TDX_OK) return rc;
    {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        uint32_t *lst = ovfa, *nxt = ovfb;
        auto drain_overflow = [&]() -> int {   // cells that did not fit a lane's private stack
            for (;;) {
                TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
                const unsigned long long novf = ctx->h_mail[0];
                if (novf == 0) return TDX_OK;
                if (novf > ovf_cap) return tdx_fail(ctx, TDX_ERR_NOMEM, "D-infinity accumulation overflow list exhausted");
                std::swap(lst, nxt);   // the list just filled becomes the input
                TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s));
                hipLaunchKernelGGL((dinf_walk_list_kernel<Alg>), dim3(tdx_blocks_for(novf, 256)), dim3(256), 0, s, alg, ang_use, d_rows, inx, st.y0, st.y1,
                                   info, contcheck, cnt, d_out, nxt, novf, lst, d_cnt, ovf_cap);
                rounds++;
            }
        };
        hipLaunchKernelGGL((dinf_walk_kernel<Alg>), grid2d, dim3(256), 0, s, alg, ang_use, d_rows, inx, st.y0, st.y1, info, contcheck, cnt, d_out, lst,
                           d_cnt, ovf_cap);
        rounds++;
        rc = drain_overflow();
        if (rc != TDX_OK) return rc;
        while (st.multi()) {
            // boundary rows of the result and of the counters (evaluated cells carry CNT_DONE) to both neighbours
            const size_t rowb = size_t(inx) * 4;
            float *r_out_up = recvbuf, *r_out_dn = recvbuf + inx;
            int32_t *r_cnt_up = reinterpret_cast<int32_t*>(recvbuf + 2 * size_t(inx)), *r_cnt_dn = reinterpret_cast<int32_t*>(recvbuf + 3 * size_t(inx));
            rc = strip_exchange_buffers(ctx, st, d_out + size_t(st.y0) * inx, d_out + size_t(st.y1 - 1) * inx, r_out_up, r_out_dn, rowb);
            if (rc != TDX_OK) return rc;
            rc = strip_exchange_buffers(ctx, st, cnt + size_t(st.y0) * inx, cnt + size_t(st.y1 - 1) * inx, r_cnt_up, r_cnt_dn, rowb);
            if (rc != TDX_OK) return rc;
            TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt + 1, 0, sizeof(unsigned long long), s));
            const unsigned gx = tdx_blocks_for(size_t(inx), 256);
            if (st.up)
                hipLaunchKernelGGL((dinf_halo_kernel<Alg>), dim3(gx), dim3(256), 0, s, alg, ang_use, d_rows, inx, st.y0, st.y1, info, contcheck, cnt, d_out,
                                   st.y0 - 1, r_out_up, r_cnt_up, lst, d_cnt, ovf_cap, d_cnt + 1);
            if (st.down)
                hipLaunchKernelGGL((dinf_halo_kernel<Alg>), dim3(gx), dim3(256), 0, s, alg, ang_use, d_rows, inx, st.y0, st.y1, info, contcheck, cnt, d_out,
                                   st.y1, r_out_dn, r_cnt_dn, lst, d_cnt, ovf_cap, d_cnt + 1);
            rc = drain_overflow();
            if (rc != TDX_OK) return rc;
            int64_t changed = 0;
            rc = strip_allreduce_device(ctx, st, d_cnt + 1, 1, TDX_OP_SUM, &changed);   // the vote: device counter -> all ranks -> host, one synchronisation
            if (rc != TDX_OK) return rc;
            if (changed == 0) break;
            outer++;
            rounds++;
        }
        if (stats) stats->launches[TDX_K_ACCUM] += rounds;
    }
    }
    {
        const size_t first = size_t(st.y0) * size_t(inx), nown = size_t(st.y1 - st.y0) * size_t(inx);
        hipLaunchKernelGGL(dinf_finish_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, d_out, first, nown, out_nodata);
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) { stt->rounds = rounds; stt->cells_evaluated = outer; }
    return TDX_OK;
}

}  // namespace

int dinf_outlet_recode(tdx_context* ctx, const Strip& st, const float* d_ang, float ang_nodata, const RowProp* d_rows, const int32_t* outlet_x,
                       const int32_t* outlet_y, int64_t n_outlets, float** ang_use, tdx_stats* stats) {
    hipStream_t s = ctx->stream;
    const int inx = st.nx, iny = st.ny_arr;
    const size_t n = size_t(inx) * size_t(iny);
    TdxSpan sp(ctx, TDX_K_BFS);
    const tilek::TileGeom geom = tilek::make_geom(inx, iny, st.y0, st.y1);
    const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y);
    int32_t* reach = static_cast<int32_t*>(ctx->scratch(TDX_S_N, n * 4));
    uint8_t* mask = static_cast<uint8_t*>(ctx->scratch(TDX_S_O, n));
    float* aprime = static_cast<float*>(ctx->scratch(TDX_S_P, n * 4));
    uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));
    unsigned long long* counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
    int32_t* d_oxy = static_cast<int32_t*>(ctx->scratch(TDX_S_R, size_t(n_outlets ? n_outlets : 1) * 8));
    if (!reach || !mask || !aprime || !flags || !counts || !d_oxy) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemsetAsync(reach, 0, n * 4, s));
    TDX_HIP_CHECK(ctx, hipMemsetAsync(flags, 0, ntiles * 4, s));
    hipLaunchKernelGGL(dinf_reach_mask_kernel, dim3((inx + 63) / 64, (iny + 3) / 4), dim3(256), 0, s, d_ang, inx, iny, ang_nodata, d_rows, mask);
    if (n_outlets > 0) {
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_oxy, outlet_x, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_oxy + n_outlets, outlet_y, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(dinf_reach_seed_kernel, dim3(tdx_blocks_for(size_t(n_outlets), 256)), dim3(256), 0, s, d_oxy, d_oxy + n_outlets, int(n_outlets), inx,
                           iny, st.y0, st.y1, geom.tiles_x, reach, flags);
    }
    int64_t rr = 0, ll = 0;
    int rc = reach_closure(ctx, st, reach, mask, flags, flags + ntiles, counts, &rr, &ll);
    if (rc != TDX_OK) return rc;
    hipLaunchKernelGGL(dinf_apply_reach_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_ang, reach, n, ang_nodata, aprime);
    *ang_use = aprime;
    if (stats) stats->launches[TDX_K_BFS] += ll;
    return TDX_OK;
}

extern "C" int tdx_areadinf_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                                const double* dxc, const double* dyc, const float* d_w, int contcheck,
                                const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_sca, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_sca || !dxc || !dyc || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_areadinf_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    AreaAlg alg{d_w};
    return run_dinf_accum(ctx, alg, strip_single(int(nx), int(ny)), const_cast<float*>(d_ang), ang_nodata, dxc, dyc, contcheck, outlet_x, outlet_y, n_outlets,
                          d_sca, TDX_AREA_NODATA, nullptr, 0.f, stats);
}

extern "C" int tdx_areadinf_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                                  const double* dxc, const double* dyc, const float* d_w, int contcheck, const int32_t* outlet_x,
                                  const int32_t* outlet_row, int64_t n_outlets, float* d_sca, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_sca || !dxc || !dyc || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_areadinf_strip: bad argument");
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    AreaAlg alg{d_w};
    return run_dinf_accum(ctx, alg, strip_from_comm(comm, int(nx), int(ny_local)), d_ang, ang_nodata, dxc, dyc, contcheck, outlet_x, outlet_row, n_outlets,
                          d_sca, TDX_AREA_NODATA, nullptr, 0.f, stats);
}

extern "C" int tdx_dinfdecayaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                                      const double* dxc, const double* dyc, const float* d_dm, float dm_nodata, const float* d_w, int contcheck,
                                      const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_dsca, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_dm || !d_dsca || !dxc || !dyc || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfdecayaccum_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    DecayAlg alg{d_w, d_dm, dm_nodata};
    return run_dinf_accum(ctx, alg, strip_single(int(nx), int(ny)), const_cast<float*>(d_ang), ang_nodata, dxc, dyc, contcheck, outlet_x, outlet_y, n_outlets,
                          d_dsca, TDX_ANG_NODATA, nullptr, dm_nodata, stats);
}

extern "C" int tdx_dinfdecayaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                                        const double* dxc, const double* dyc, float* d_dm, float dm_nodata, const float* d_w, int contcheck,
                                        const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets, float* d_dsca, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_dm || !d_dsca || !dxc || !dyc || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfdecayaccum_strip: bad argument");
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    DecayAlg alg{d_w, d_dm, dm_nodata};
    return run_dinf_accum(ctx, alg, strip_from_comm(comm, int(nx), int(ny_local)), d_ang, ang_nodata, dxc, dyc, contcheck, outlet_x, outlet_row, n_outlets,
                          d_dsca, TDX_ANG_NODATA, d_dm, dm_nodata, stats);
}

extern "C" int tdx_areadinf(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                            const double* dxc, const double* dyc, const float* w, int contcheck,
                            const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* sca, tdx_stats* stats) {
    if (!ctx || !ang || !sca || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_areadinf: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_s = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_w = w ? static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4)) : nullptr;
    if (!d_a || !d_s || (w && !d_w)) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_a, ang, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (w) TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_w, w, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = tdx_areadinf_dev(ctx, d_a, nx, ny, ang_nodata, dxc, dyc, d_w, contcheck, outlet_x, outlet_y, n_outlets, d_s, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(sca, d_s, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}

extern "C" int tdx_dinfdecayaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                                  const double* dxc, const double* dyc, const float* dm, float dm_nodata, const float* w, int contcheck,
                                  const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* dsca, tdx_stats* stats) {
    if (!ctx || !ang || !dm || !dsca || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfdecayaccum: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_s = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_d = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    float* d_w = w ? static_cast<float*>(ctx->scratch(TDX_S_IO3, n * 4)) : nullptr;
    if (!d_a || !d_s || !d_d || (w && !d_w)) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_a, ang, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_d, dm, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (w) TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_w, w, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = tdx_dinfdecayaccum_dev(ctx, d_a, nx, ny, ang_nodata, dxc, dyc, d_d, dm_nodata, d_w, contcheck, outlet_x, outlet_y, n_outlets, d_s, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(dsca, d_s, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
