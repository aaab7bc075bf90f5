// PitRemove on gfx950: replaces the compute part of flood() (src/flood.cpp:243-479).
//
// The reference iterates  W <- (Z >= m ? Z : min(W, m)),  m = min over accessible neighbours of W,
// on cells with W > Z, from the seed surface of src/flood.cpp:243-271, with a raster scan followed
// by alternating stacks of still-raised cells (src/flood.cpp:292-479).  The update uses only float
// comparisons, is monotone non-increasing from the FLT_MAX start and has a unique fixed point
// (the minimax-path surface), so ANY schedule that runs to convergence yields
// the same bits.  Schedule used here:
//
//   * pit_seed_kernel    streaming 3x3 stencil: Z -> W0                 (src/flood.cpp:243-271)
//   * pit_relax_kernel   one workgroup per ACTIVE 64x64 tile: W tile + 1-cell halo staged in LDS
//                        (66x66 f32 = 17 KB), Z kept in registers, chaotic in-LDS relaxation until the
//                        tile stops changing, then write-back; tiles whose rim changed re-activate
//                        their neighbours for the next round
//   * pit_compact_kernel active flags -> compact tile list (wave ballot + one atomic per wave)
//
// HBM traffic per round = 3 tile images per active tile; rounds ~ longest fill path / 64.
#include "context.hpp"
#include "device_common.hpp"

namespace {

constexpr int TILE = 64;
constexpr int LDS_W = TILE + 2;
constexpr int ROWS_PER_WAVE = 16;   // 4 waves x 16 rows

__global__ __launch_bounds__(256) void pit_seed_kernel(const float* __restrict__ Z, const int16_t* __restrict__ mask,
                                                       float* __restrict__ W, int nx, int ny, float nodata, int step) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    const float z = Z[idx];
    float w;
    if (tdxk::is_nodata_f(z, nodata)) w = TDX_FEL_NODATA;
    else if (mask && mask[idx] == 1) w = z;
    else if (x == 0 || y == 0 || x == nx - 1 || y == ny - 1) w = z;   // !hasAccess(i+-1,j+-1): global edge ring
    else {
        bool con = false;
        for (int k = 1; k <= 8; k += step) {
            const float zn = Z[size_t(y + tdxk::d2(k)) * size_t(nx) + size_t(x + tdxk::d1(k))];
            con = con || tdxk::is_nodata_f(zn, nodata);
        }
        w = con ? z : FLT_MAX;
    }
    W[idx] = w;
}

// flags_next bit layout: any non-zero = active
__global__ __launch_bounds__(256) void pit_relax_kernel(const float* __restrict__ Z, float* __restrict__ W, int nx, int ny,
                                                        int tiles_x, int tiles_y, const uint32_t* __restrict__ list,
                                                        uint32_t* __restrict__ flags_next, int fourway) {
    __shared__ float sW[LDS_W * LDS_W];
    __shared__ int sRim[8];   // N S W E NW NE SW SE rim-changed flags
    const int tile = int(list[blockIdx.x]);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x0 = tx * TILE, y0 = ty * TILE;
    const int tid = threadIdx.x;
    if (tid < 8) sRim[tid] = 0;
    // stage W tile + halo; off-grid cells read as FLT_MAX (neutral for the min, never updated)
    for (int e = tid; e < LDS_W * LDS_W; e += 256) {
        const int ly = e / LDS_W, lx = e - ly * LDS_W;
        const int gx = x0 + lx - 1, gy = y0 + ly - 1;
        float v = FLT_MAX;
        if (gx >= 0 && gx < nx && gy >= 0 && gy < ny) v = W[size_t(gy) * size_t(nx) + size_t(gx)];
        sW[e] = v;
    }
    const int lx = tid & 63;                 // column inside the tile
    const int ry0 = (tid >> 6) * ROWS_PER_WAVE;
    const int gx = x0 + lx;
    float z[ROWS_PER_WAVE], w0[ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; r++) {
        const int gy = y0 + ry0 + r;
        z[r] = (gx < nx && gy < ny) ? Z[size_t(gy) * size_t(nx) + size_t(gx)] : FLT_MAX;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; r++) w0[r] = sW[(ry0 + r + 1) * LDS_W + lx + 1];

    bool any_change = false;
    for (int iter = 0;; iter++) {
        int changed = 0;
        if ((iter & 1) == 0) {
#pragma unroll
            for (int r = 0; r < ROWS_PER_WAVE; r++) {
                const int c = (ry0 + r + 1) * LDS_W + lx + 1;
                const float w = sW[c];
                if (w > z[r]) {
                    float m = fminf(fminf(sW[c + 1], sW[c - 1]), fminf(sW[c - LDS_W], sW[c + LDS_W]));
                    if (!fourway) m = fminf(m, fminf(fminf(sW[c - LDS_W + 1], sW[c - LDS_W - 1]), fminf(sW[c + LDS_W - 1], sW[c + LDS_W + 1])));
                    const float wn = fmaxf(z[r], fminf(w, m));
                    if (wn != w) { sW[c] = wn; changed = 1; }
                }
            }
        } else {
#pragma unroll
            for (int r = ROWS_PER_WAVE - 1; r >= 0; r--) {
                const int c = (ry0 + r + 1) * LDS_W + lx + 1;
                const float w = sW[c];
                if (w > z[r]) {
                    float m = fminf(fminf(sW[c + 1], sW[c - 1]), fminf(sW[c - LDS_W], sW[c + LDS_W]));
                    if (!fourway) m = fminf(m, fminf(fminf(sW[c - LDS_W + 1], sW[c - LDS_W - 1]), fminf(sW[c + LDS_W - 1], sW[c + LDS_W + 1])));
                    const float wn = fmaxf(z[r], fminf(w, m));
                    if (wn != w) { sW[c] = wn; changed = 1; }
                }
            }
        }
        const int any = __syncthreads_or(changed);
        if (!any) break;
        any_change = true;
    }
    if (!any_change) return;   // uniform: __syncthreads_or returned the same value to every thread
    // write back and detect rim changes
    int rim = 0;
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; r++) {
        const int ly = ry0 + r;
        const float w = sW[(ly + 1) * LDS_W + lx + 1];
        if (w != w0[r]) {
            const int gy = y0 + ly;
            W[size_t(gy) * size_t(nx) + size_t(gx)] = w;   // changed cells are always in-grid
            const bool top = (ly == 0), bot = (ly == TILE - 1), lef = (lx == 0), rig = (lx == TILE - 1);
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
    if (rim) {
#pragma unroll
        for (int b = 0; b < 8; b++) if (rim & (1 << b)) sRim[b] = 1;
    }
    __syncthreads();
    if (tid < 8 && sRim[tid]) {
        const int ddx[8] = {0, 0, -1, 1, -1, 1, -1, 1};
        const int ddy[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
        const int ntx = tx + ddx[tid], nty = ty + ddy[tid];
        if (ntx >= 0 && ntx < tiles_x && nty >= 0 && nty < tiles_y) flags_next[nty * tiles_x + ntx] = 1u;
    }
}

__global__ __launch_bounds__(256) void pit_compact_kernel(uint32_t* __restrict__ flags, int ntiles, uint32_t* __restrict__ list,
                                                          unsigned long long* __restrict__ counter) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bool act = false;
    if (t < ntiles) { act = flags[t] != 0u; flags[t] = 0u; }
    tdxk::wave_append(act, uint32_t(t), list, counter);
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

extern "C" int tdx_pitremove_dev(tdx_context* ctx, const float* d_dem, int64_t nx, int64_t ny, float dem_nodata,
                                 const int16_t* d_mask, int fourway, float* d_fel, tdx_stats* stats) {
    if (!ctx || !d_dem || !d_fel || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_pitremove_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = int(nx), iny = int(ny);
    const int tiles_x = (inx + TILE - 1) / TILE, tiles_y = (iny + TILE - 1) / TILE;
    const int ntiles = tiles_x * tiles_y;
    uint32_t* flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_A, size_t(ntiles) * 4));
    uint32_t* list = static_cast<uint32_t*>(ctx->scratch(TDX_S_B, size_t(ntiles) * 4));
    if (!flags || !list) return TDX_ERR_NOMEM;
    unsigned long long* d_count = reinterpret_cast<unsigned long long*>(ctx->d_mail);

    ctx->begin_call(stats);
    {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        dim3 grid((inx + 63) / 64, (iny + 3) / 4);
        hipLaunchKernelGGL(pit_seed_kernel, grid, dim3(256), 0, s, d_dem, d_mask, d_fel, inx, iny, dem_nodata, fourway ? 2 : 1);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    }
    // round 0: every tile is active
    hipLaunchKernelGGL(fill_u32_kernel, dim3(tdx_blocks_for(size_t(ntiles), 256)), dim3(256), 0, s, flags, 1u, size_t(ntiles));
    int64_t rounds = 0;
    {
        TdxSpan sp(ctx, TDX_K_RELAX);
        for (;;) {
            TDX_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(pit_compact_kernel, dim3(tdx_blocks_for(size_t(ntiles), 256)), dim3(256), 0, s, flags, ntiles, list, d_count);
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
            const unsigned long long nact = ctx->h_mail[0];
            if (nact == 0) break;
            hipLaunchKernelGGL(pit_relax_kernel, dim3(unsigned(nact)), dim3(256), 0, s, d_dem, d_fel, inx, iny, tiles_x, tiles_y, list, flags, fourway);
            rounds++;
            if (stats) stats->launches[TDX_K_RELAX]++;
        }
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    if (stats) stats->rounds = rounds;
    ctx->end_call();
    return TDX_OK;
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
