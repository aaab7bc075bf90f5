// GridNet (longest / total upstream path length and Strahler order on the D8 grid, src/gridnet.cpp:54-514, the branch
// without outlets) and Threshold (src/Threshold.cpp:49-162) - SURVEY.md 8(f) rank 2, the step after AreaD8 in every
// stream-delineation workflow.
//
// GridNet is the D8 dependency sweep of AreaD8 with a different per-cell expression (three outputs instead of one), so it
// reuses the scheme of aread8.hip's exact pull walk: an in-degree per cell, one lane per ready cell, a lane evaluates its
// cell by PULLING its contributors in k order (the reference's order of float32 operations, src/gridnet.cpp:392-422),
// publishes the three values, decrements the in-degree of the downstream cell and continues there iff it was the last
// contributor.  The 3x3 window of directions / mask flags of the next cell is prefetched with the decrement, and all
// contributor values are requested before the first one is used (a hop is a chain of memory round trips).
//
// Default path: the tile dependency sweep of d8_sweep.hpp (LDS tiles on the round schedule; outlets through the upstream
// closure of flats.hpp: reach_closure, src/gridnet.cpp:269-369; row strips).  The pull walk below is kept as an A/B hook
// (TDX_GN_WALK=1: single raster, no outlets).
// Deliberate restriction (same as oracle/taudem_oracle.c: orc_gridnet): a neighbour whose COLUMN lies outside the raster
// is skipped; the reference reads it through linearpart::getData, which returns a stale temporary for an out-of-range x
// (src/linearpart.h:501-512).  It only concerns ring cells, which carry nodata in every D8FlowDir output.
#include "context.hpp"
#include "d8_sweep.hpp"
#include "device_common.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace {
using namespace tdxk;

constexpr int32_t GN_NOT_PART = 0x40000000;   // nodata direction: never counted down to 0
constexpr int32_t GN_SOURCE = -1;             // no contributor (a value no decrement produces, see aread8.hip)
constexpr int32_t GN_DONE = -2;

__device__ __forceinline__ bool mask_ok(const int32_t* __restrict__ mask, size_t idx, int thresh) { return !mask || mask[idx] >= thresh; }

// in-degree (src/gridnet.cpp:238-267), initial outputs (src/gridnet.cpp:176-178,228-235)
__global__ __launch_bounds__(256) void gn_setup_kernel(const int16_t* __restrict__ P, int nx, int ny, int16_t nodata, const int32_t* __restrict__ mask,
                                                       int thresh, int32_t* __restrict__ cnt, float* __restrict__ plen, float* __restrict__ tlen,
                                                       int32_t* __restrict__ gord) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    const bool valid = !is_nodata_s(P[idx], nodata);
    int32_t c = GN_NOT_PART;
    if (valid) {
        c = 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            const int xn = x + d1(k), yn = y + d2(k);
            if (xn < 0 || xn >= nx || yn < 0 || yn >= ny) continue;
            const int16_t pn = P[size_t(yn) * size_t(nx) + size_t(xn)];
            if (!is_nodata_s(pn, nodata) && (pn - k == 4 || pn - k == -4)) c++;
        }
        if (c == 0) c = GN_SOURCE;
    }
    cnt[idx] = c;
    plen[idx] = -1.0f;
    tlen[idx] = -1.0f;
    gord[idx] = (valid && mask_ok(mask, idx, thresh)) ? 1 : -1;
}

// The 3x3 window of a cell (read-only during the sweep): pk[0] = the cell's direction, pk[k] = neighbour k's;
// bit k of `in` = neighbour k is inside the raster, bit k of `mk` = its mask value passes (bit 0: the cell itself).
struct GnWindow { int16_t pk[9]; unsigned in, mk; };
__device__ __forceinline__ void gn_load_window(const int16_t* __restrict__ P, const int32_t* __restrict__ mask, int thresh, int nx, int ny, int x, int y,
                                               size_t idx, GnWindow& w) {
    w.pk[0] = P[idx];
    w.in = 1u;
    w.mk = mask_ok(mask, idx, thresh) ? 1u : 0u;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        const int xn = x + d1(k), yn = y + d2(k);
        const bool in = xn >= 0 && xn < nx && yn >= 0 && yn < ny;
        const size_t n = in ? size_t(yn) * size_t(nx) + size_t(xn) : idx;
        w.pk[k] = P[n];
        if (in) w.in |= 1u << k;
        if (in && mask_ok(mask, n, thresh)) w.mk |= 1u << k;
    }
}

// src/gridnet.cpp:380-426.  Returns false when the cell's own mask value does not pass (nothing is written then).
__device__ __forceinline__ bool gn_evaluate(const GnWindow& w, const float* __restrict__ dist, int nx, int x, int y, const float* __restrict__ plen,
                                            const float* __restrict__ tlen, const int32_t* __restrict__ gord, float& pl_out, float& tl_out, int& go_out) {
    if (!(w.mk & 1u)) return false;
    unsigned contrib = 0;
    float pk[9], tk[9];
    int gk[9];
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        pk[k] = 0.f; tk[k] = 0.f; gk[k] = 0;
        const int16_t sdir = w.pk[k];
        if (!((w.in >> k) & 1u) || sdir <= 0 || !((w.mk >> k) & 1u) || !(sdir - k == 4 || sdir - k == -4)) continue;
        contrib |= 1u << k;
        const size_t n = size_t(y + d2(k)) * size_t(nx) + size_t(x + d1(k));
        pk[k] = ld_agent(&plen[n]);
        tk[k] = ld_agent(&tlen[n]);
        gk[k] = __hip_atomic_load(&gord[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float tl = 0.0f, pl = 0.0f;
    int a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        if (!((contrib >> k) & 1u)) continue;
        const int g = gk[k];                       // Strahler order (src/gridnet.cpp:404-411)
        if (g >= a1) { a2 = a1; a1 = g; }
        else if (g > a2) a2 = g;
        const float dd = dist[size_t(y) * 9 + size_t(w.pk[k])];   // dist[j][sdir]: the row of the evaluated cell, the code of the neighbour
        const float ld = pk[k] + dd;
        tl = tl + (float)(tk[k] + dd);
        if (ld > pl) pl = ld;
    }
    pl_out = pl; tl_out = tl;
    go_out = (a2 + 1 > a1) ? a2 + 1 : a1;
    return true;
}

__device__ __forceinline__ void gn_walk_from(size_t idx, const int16_t* __restrict__ P, const int32_t* __restrict__ mask, int thresh,
                                             const float* __restrict__ dist, int nx, int ny, int16_t nodata, int32_t* __restrict__ cnt,
                                             float* __restrict__ plen, float* __restrict__ tlen, int32_t* __restrict__ gord) {
    int x = int(idx % size_t(nx)), y = int(idx / size_t(nx));
    GnWindow w;
    gn_load_window(P, mask, thresh, nx, ny, x, y, idx, w);
    for (;;) {
        float pl, tl;
        int go;
        if (gn_evaluate(w, dist, nx, x, y, plen, tlen, gord, pl, tl, go)) {
            st_agent(&plen[idx], pl);
            st_agent(&tlen[idx], tl);
            __hip_atomic_store(&gord[idx], go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cnt[idx] = GN_DONE;
        const int16_t k = w.pk[0];
        if (k < 1 || k > 8) return;                 // (k == 0 decrements the cell itself in the reference: no effect)
        if (!((w.in >> k) & 1u)) return;            // off the raster
        if (is_nodata_s(w.pk[k], nodata)) return;   // src/gridnet.cpp:435
        const int xn = x + d1(k), yn = y + d2(k);
        const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
        drain_stores();                             // the values must be at the coherence point before the counter moves
        const int32_t old = __hip_atomic_fetch_sub(&cnt[n], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        GnWindow wn;                                // the next hop's window travels with the atomic, not after it
        gn_load_window(P, mask, thresh, nx, ny, xn, yn, n, wn);
        if (old != 1) return;                       // somebody else is the last contributor
        x = xn; y = yn; idx = n; w = wn;
    }
}

__global__ __launch_bounds__(256) void gn_walk_kernel(const int16_t* __restrict__ P, const int32_t* __restrict__ mask, int thresh,
                                                      const float* __restrict__ dist, int nx, int ny, int16_t nodata, int32_t* __restrict__ cnt,
                                                      float* __restrict__ plen, float* __restrict__ tlen, int32_t* __restrict__ gord) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    if (cnt[idx] != GN_SOURCE) return;
    gn_walk_from(idx, P, mask, thresh, dist, nx, ny, nodata, cnt, plen, tlen, gord);
}

__global__ __launch_bounds__(256) void gn_narrow_kernel(const int32_t* __restrict__ g32, size_t n, int16_t* __restrict__ g16) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) g16[i] = int16_t(g32[i]);
}

// src/Threshold.cpp:116-137
__global__ __launch_bounds__(256) void threshold_kernel(const float* __restrict__ ssa, size_t n, float ssa_nodata, const float* __restrict__ mask,
                                                        float thresh, int16_t* __restrict__ src) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = ssa[i];
    int16_t s;
    if (is_nodata_f(v, ssa_nodata)) s = int16_t(-32768);
    else if (mask) s = ((v >= thresh) & (mask[i] >= 0)) ? 1 : 0;
    else s = (v >= thresh) ? 1 : 0;
    src[i] = s;
}

}  // namespace

namespace {
// initial records on the owned rows (src/gridnet.cpp:176-178, 228-235; with outlets :271-283): plen pending on participating cells
// (-1 elsewhere), tlen -1; gord 1 on cells that will be evaluated (non-nodata cells whose mask value passes / the outlets' upstream
// closure), 0 on non-nodata cells outside the closure, -1 on nodata
__global__ __launch_bounds__(256) void gn_init_kernel(const uint32_t* __restrict__ info, const int16_t* __restrict__ P, int16_t nodata, int use_outlets, size_t first,
                                                      size_t n, float4* __restrict__ rec) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= first + n) return;
    const unsigned inf = info[i];
    const bool valid = !is_nodata_s(P[i], nodata);
    int g;
    if (use_outlets) g = (inf & d8sweep::INFO_PART) ? 1 : (valid ? 0 : -1);
    else g = (valid && (inf & d8sweep::INFO_OWNMASK)) ? 1 : -1;
    rec[i] = make_float4((inf & d8sweep::INFO_PART) ? __uint_as_float(d8sweep::PENDING_BITS) : -1.0f, -1.0f, __int_as_float(g), 0.f);
}
// records -> the three rasters; a cell that never became ready (cycle, p == 0 quirk) keeps plen -1 like the reference's never-queued cells
__global__ __launch_bounds__(256) void gn_unpack_kernel(const float4* __restrict__ rec, size_t first, size_t n, float* __restrict__ plen, float* __restrict__ tlen,
                                                        int16_t* __restrict__ gord) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= first + n) return;
    const float4 r = rec[i];
    plen[i] = d8sweep::pending(r.x) ? -1.0f : r.x;
    tlen[i] = r.y;
    gord[i] = int16_t(__float_as_int(r.z));
}

// One strip of gridnet() (src/gridnet.cpp:54-514).  dxc / dyc: cell sizes of the rows of the strip array.
int gridnet_impl(tdx_context* ctx, const Strip& st, int16_t* d_p, int16_t p_nodata, const double* dxc, const double* dyc, const int32_t* d_mask, int32_t thresh,
                 const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_plen, float* d_tlen, int16_t* d_gord, tdx_stats* stats) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = st.nx, iny = st.ny_arr;
    const size_t n = size_t(inx) * size_t(iny);
    if (!d_mask) thresh = 0;   // src/gridnet.cpp:155-159
    // dist[row][k] = sqrt((dx d1)^2 + (dy d2)^2) in double, stored as float (src/gridnet.cpp:196-209)
    static const int hd1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1};
    static const int hd2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};
    std::vector<float> dist(size_t(iny) * 9, 0.f);
    for (int m = 0; m < iny; m++)
        for (int k = 1; k <= 8; k++)
            dist[size_t(m) * 9 + size_t(k)] = (float)sqrt(dxc[m] * dxc[m] * hd1[k] * hd1[k] + dyc[m] * dyc[m] * hd2[k] * hd2[k]);
    float* d_dist = static_cast<float*>(ctx->scratch(TDX_S_J, dist.size() * sizeof(float)));
    if (!d_dist) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_dist, dist.data(), dist.size() * sizeof(float), hipMemcpyHostToDevice, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));   // `dist` is a local
    const dim3 grid2d((inx + 63) / 64, (iny + 3) / 4);
    const size_t first = size_t(st.y0) * size_t(inx), nown = size_t(st.y1 - st.y0) * size_t(inx);

    if (getenv("TDX_GN_WALK") && !st.multi() && n_outlets < 0) {   // A/B hook: the atomic pull walk
        int32_t* cnt = static_cast<int32_t*>(ctx->scratch(TDX_S_A, n * 4));
        int32_t* gord32 = static_cast<int32_t*>(ctx->scratch(TDX_S_B, n * 4));
        if (!cnt || !gord32) return TDX_ERR_NOMEM;
        ctx->begin_call(stats);
        strip_mark(ctx, st, "gridnet");
        {
            TdxSpan sp(ctx, TDX_K_STENCIL);
            hipLaunchKernelGGL(gn_setup_kernel, grid2d, dim3(256), 0, s, d_p, inx, iny, p_nodata, d_mask, int(thresh), cnt, d_plen, d_tlen, gord32);
            if (stats) stats->launches[TDX_K_STENCIL]++;
        }
        {
            TdxSpan sp(ctx, TDX_K_ACCUM);
            hipLaunchKernelGGL(gn_walk_kernel, grid2d, dim3(256), 0, s, d_p, d_mask, int(thresh), d_dist, inx, iny, p_nodata, cnt, d_plen, d_tlen, gord32);
            if (stats) stats->launches[TDX_K_ACCUM]++;
        }
        {
            TdxSpan sp(ctx, TDX_K_MISC);
            hipLaunchKernelGGL(gn_narrow_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, gord32, n, d_gord);
            if (stats) stats->launches[TDX_K_MISC]++;
        }
        TDX_HIP_CHECK(ctx, hipGetLastError());
        ctx->end_call();
        return TDX_OK;
    }

    const tilek::TileGeom geom = tilek::make_geom(inx, iny, st.y0, st.y1);
    const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y);
    uint32_t* info = static_cast<uint32_t*>(ctx->scratch(TDX_S_A, n * 4));
    uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));
    unsigned long long* counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
    if (!info || !flags || !counts) return TDX_ERR_NOMEM;
    ctx->begin_call(stats);
    strip_mark(ctx, st, "gridnet");
    int rc = strip_exchange<int16_t>(ctx, st, d_p, p_nodata);   // directions of the neighbours' boundary rows
    if (rc != TDX_OK) return rc;
    if (d_mask) {   // ... and their mask values (the value-contributor rule looks at the neighbour's mask)
        rc = strip_exchange<int32_t>(ctx, st, const_cast<int32_t*>(d_mask), int32_t(thresh) - 1);
        if (rc != TDX_OK) return rc;
    }
    int32_t* reach = nullptr;
    if (n_outlets >= 0) {
        // upstream closure of the outlets (src/gridnet.cpp:285-340: level by level, one MPI round per level; here one more fixed
        // point of the tile engine, across strips)
        TdxSpan sp(ctx, TDX_K_BFS);
        reach = static_cast<int32_t*>(ctx->scratch(TDX_S_N, n * 4));
        uint8_t* rmask = static_cast<uint8_t*>(ctx->scratch(TDX_S_O, n));
        int32_t* d_oxy = static_cast<int32_t*>(ctx->scratch(TDX_S_R, size_t(n_outlets ? n_outlets : 1) * 8));
        if (!reach || !rmask || !d_oxy) return TDX_ERR_NOMEM;
        TDX_HIP_CHECK(ctx, hipMemsetAsync(reach, 0, n * 4, s));
        TDX_HIP_CHECK(ctx, hipMemsetAsync(flags, 0, ntiles * 4, s));
        hipLaunchKernelGGL(d8sweep::reach_mask_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_p, n, p_nodata, rmask);
        if (n_outlets > 0) {
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_oxy, outlet_x, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_oxy + n_outlets, outlet_y, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
            // skip_nodata = 1: an outlet on a cell without direction is IGNORED.  The reference seeds it (gord = 1) and then indexes d1[] / d2[] with the nodata code
            // (src/gridnet.cpp:285-340): undefined behaviour, no defined result to reproduce - documented in DESIGN.md 4.5, pinned by tests/test_gpu_gridnet.py
            hipLaunchKernelGGL(d8sweep::reach_seed_kernel, dim3(tdx_blocks_for(size_t(n_outlets), 256)), dim3(256), 0, s, d_oxy, d_oxy + n_outlets, int(n_outlets), inx,
                               iny, st.y0, st.y1, geom.tiles_x, d_p, p_nodata, 1, reach, flags);
        }
        int64_t rr = 0, ll = 0;
        rc = reach_closure(ctx, st, reach, rmask, flags, flags + ntiles, counts, &rr, &ll);
        if (rc != TDX_OK) return rc;
        if (stats) stats->launches[TDX_K_BFS] += ll;
    }
    float4* rec = static_cast<float4*>(ctx->scratch(TDX_S_C, n * 16));   // {plen, tlen, gord, -} per cell: one record, one store (d8_sweep.hpp)
    if (!rec) return TDX_ERR_NOMEM;
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        hipLaunchKernelGGL(d8sweep::setup_kernel, grid2d, dim3(256), 0, s, d_p, inx, iny, p_nodata, 1, d_mask, int(thresh), reach, info);
        hipLaunchKernelGGL(gn_init_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, info, d_p, p_nodata, n_outlets >= 0 ? 1 : 0, first, nown, rec);
        if (stats) stats->launches[TDX_K_STENCIL] += 2;
    }
    {
        const float4 oc = d8sweep::GridNetAlg::outside();
        uint4 ob;
        memcpy(&ob, &oc, sizeof(ob));
        rc = strip_exchange<uint4>(ctx, st, reinterpret_cast<uint4*>(rec), ob);
        if (rc != TDX_OK) return rc;
    }
    int64_t rounds = 0, launches = 0, outer = 1;
    {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        d8sweep::Arrays<d8sweep::GridNetAlg> A{rec, nullptr, d_dist, nullptr, info};
        rc = d8sweep::run(ctx, st, d8sweep::GridNetAlg{}, A, flags, counts, &rounds, &launches, &outer);
        if (rc != TDX_OK) return rc;
        if (stats) stats->launches[TDX_K_ACCUM] += launches;
    }
    {
        TdxSpan sp(ctx, TDX_K_MISC);
        hipLaunchKernelGGL(gn_unpack_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, rec, first, nown, d_plen, d_tlen, d_gord);
        if (stats) stats->launches[TDX_K_MISC]++;
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) { stt->rounds = outer; stt->cells_evaluated = rounds; }
    return TDX_OK;
}
}  // namespace

extern "C" int tdx_gridnet_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata, const double* dxc, const double* dyc,
                               const int32_t* d_mask, int32_t thresh, const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_plen,
                               float* d_tlen, int16_t* d_gord, tdx_stats* stats) {
    if (!ctx || !d_p || !dxc || !dyc || !d_plen || !d_tlen || !d_gord || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_gridnet_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells");
    if (n_outlets > 0 && (!outlet_x || !outlet_y)) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_gridnet_dev: outlets missing");
    return gridnet_impl(ctx, strip_single(int(nx), int(ny)), const_cast<int16_t*>(d_p), p_nodata, dxc, dyc, d_mask, thresh, outlet_x, outlet_y, n_outlets, d_plen,
                        d_tlen, d_gord, stats);
}

extern "C" int tdx_gridnet_strip(tdx_context* ctx, const tdx_comm* comm, int16_t* d_p, int64_t nx, int64_t ny_local, int16_t p_nodata, const double* dxc,
                                 const double* dyc, int32_t* d_mask, int32_t thresh, const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets,
                                 float* d_plen, float* d_tlen, int16_t* d_gord, tdx_stats* stats) {
    if (!ctx || !d_p || !dxc || !dyc || !d_plen || !d_tlen || !d_gord || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_gridnet_strip: bad argument");
    if (nx > 0x7fffffff || ny_local > 0x7ffffff0 || uint64_t(nx) * uint64_t(ny_local + 2) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    if (n_outlets > 0 && (!outlet_x || !outlet_row)) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_gridnet_strip: outlets missing");
    return gridnet_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_p, p_nodata, dxc, dyc, d_mask, thresh, outlet_x, outlet_row, n_outlets, d_plen, d_tlen,
                        d_gord, stats);
}

extern "C" int tdx_gridnet(tdx_context* ctx, const int16_t* p, int64_t nx, int64_t ny, int16_t p_nodata, const double* dxc, const double* dyc,
                           const int32_t* mask, int32_t thresh, const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* plen, float* tlen,
                           int16_t* gord, tdx_stats* stats) {
    if (!ctx || !p || !plen || !tlen || !gord || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_gridnet: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    int16_t* d_p = static_cast<int16_t*>(ctx->scratch(TDX_S_IO0, n * 2));
    float* d_pl = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_tl = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    int16_t* d_go = static_cast<int16_t*>(ctx->scratch(TDX_S_IO3, n * 2));
    int32_t* d_m = mask ? static_cast<int32_t*>(ctx->scratch(TDX_S_IO4, n * 4)) : nullptr;
    if (!d_p || !d_pl || !d_tl || !d_go || (mask && !d_m)) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_p, p, n * 2, hipMemcpyHostToDevice, ctx->stream));
    if (mask) TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_m, mask, n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int rc = tdx_gridnet_dev(ctx, d_p, nx, ny, p_nodata, dxc, dyc, d_m, thresh, outlet_x, outlet_y, n_outlets, d_pl, d_tl, d_go, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(plen, d_pl, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(tlen, d_tl, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(gord, d_go, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}

extern "C" int tdx_threshold_dev(tdx_context* ctx, const float* d_ssa, int64_t nx, int64_t ny, float ssa_nodata, const float* d_mask, float thresh,
                                 int16_t* d_src, tdx_stats* stats) {
    if (!ctx || !d_ssa || !d_src || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_threshold_dev: bad argument");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t n = size_t(nx) * size_t(ny);
    ctx->begin_call(stats);
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        hipLaunchKernelGGL(threshold_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_ssa, n, ssa_nodata, d_mask, thresh, d_src);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    ctx->end_call();
    return TDX_OK;
}

extern "C" int tdx_threshold(tdx_context* ctx, const float* ssa, int64_t nx, int64_t ny, float ssa_nodata, const float* mask, float thresh, int16_t* src,
                             tdx_stats* stats) {
    if (!ctx || !ssa || !src || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_threshold: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_m = mask ? static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4)) : nullptr;
    int16_t* d_s = static_cast<int16_t*>(ctx->scratch(TDX_S_IO2, n * 2));
    if (!d_a || !d_s || (mask && !d_m)) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_a, ssa, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (mask) TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_m, mask, n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int rc = tdx_threshold_dev(ctx, d_a, nx, ny, ssa_nodata, d_m, thresh, d_s, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(src, d_s, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
