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
// Not built yet: the outlets branch (src/gridnet.cpp:269-369) and row strips.
// Deliberate restriction (same as oracle/taudem_oracle.c: orc_gridnet): a neighbour whose COLUMN lies outside the raster
// is skipped; the reference reads it through linearpart::getData, which returns a stale temporary for an out-of-range x
// (src/linearpart.h:501-512).  It only concerns ring cells, which carry nodata in every D8FlowDir output.
#include "context.hpp"
#include "device_common.hpp"

#include <cmath>
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
        const int32_t old = __hip_atomic_fetch_sub(&cnt[n], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

extern "C" int tdx_gridnet_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata, const double* dxc, const double* dyc,
                               const int32_t* d_mask, int32_t thresh, float* d_plen, float* d_tlen, int16_t* d_gord, tdx_stats* stats) {
    if (!ctx || !d_p || !dxc || !dyc || !d_plen || !d_tlen || !d_gord || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_gridnet_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = int(nx), iny = int(ny);
    const size_t n = size_t(inx) * size_t(iny);
    if (!d_mask) thresh = 0;   // src/gridnet.cpp:155-159
    // dist[row][k] = sqrt((dx d1)^2 + (dy d2)^2) in double, stored as float (src/gridnet.cpp:196-209)
    static const int hd1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1};
    static const int hd2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};
    std::vector<float> dist(size_t(iny) * 9, 0.f);
    for (int m = 0; m < iny; m++)
        for (int k = 1; k <= 8; k++)
            dist[size_t(m) * 9 + size_t(k)] = (float)sqrt(dxc[m] * dxc[m] * hd1[k] * hd1[k] + dyc[m] * dyc[m] * hd2[k] * hd2[k]);
    int32_t* cnt = static_cast<int32_t*>(ctx->scratch(TDX_S_A, n * 4));
    int32_t* gord32 = static_cast<int32_t*>(ctx->scratch(TDX_S_B, n * 4));
    float* d_dist = static_cast<float*>(ctx->scratch(TDX_S_J, dist.size() * sizeof(float)));
    if (!cnt || !gord32 || !d_dist) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_dist, dist.data(), dist.size() * sizeof(float), hipMemcpyHostToDevice, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));   // `dist` is a local
    ctx->begin_call(stats);
    const dim3 grid2d((inx + 63) / 64, (iny + 3) / 4);
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

extern "C" int tdx_gridnet(tdx_context* ctx, const int16_t* p, int64_t nx, int64_t ny, int16_t p_nodata, const double* dxc, const double* dyc,
                           const int32_t* mask, int32_t thresh, float* plen, float* tlen, int16_t* gord, tdx_stats* stats) {
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
    const int rc = tdx_gridnet_dev(ctx, d_p, nx, ny, p_nodata, dxc, dyc, d_m, thresh, d_pl, d_tl, d_go, stats);
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
