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

namespace {
using namespace tdxk;

constexpr int32_t CNT_NOT_PART = 0x40000000;   // never reaches 0: the reference's int16 counter wraps instead (src/aread8.cpp:266-268)
// A cell with no contributor is a SOURCE.  It must stay distinguishable from a cell whose counter was
// driven to 0 by its contributors (that cell is evaluated by its last contributor's lane, and its own
// lane may start later and must not evaluate it again), so sources carry a value no decrement produces.
constexpr int32_t CNT_SOURCE = -1;

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

__global__ __launch_bounds__(256) void ad8_setup_kernel(const int16_t* __restrict__ P, int nx, int ny, int16_t nodata,
                                                        int32_t* __restrict__ cnt, float* __restrict__ A) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const size_t idx = size_t(y) * size_t(nx) + size_t(x);
    const int16_t p = P[idx];
    int32_t c = CNT_NOT_PART;
    if (!is_nodata_s(p, nodata) && p >= 0 && p <= 8) {
        c = d8_indegree(P, nx, ny, x, y, nodata);
        if (c == 0) c = CNT_SOURCE;
    }
    cnt[idx] = c;
    A[idx] = TDX_AREA_NODATA;
}

// outlets mode: cnt pre-filled with CNT_NOT_PART, A with -1; frontier cells get their in-degree and
// push their contributing neighbours (src/commonLib.cpp:312-359)
__global__ __launch_bounds__(256) void ad8_outlet_expand_kernel(const int16_t* __restrict__ P, int nx, int ny, int16_t nodata,
                                                                const uint32_t* __restrict__ fin, unsigned long long nin,
                                                                int32_t* __restrict__ cnt, int32_t* __restrict__ mark,
                                                                uint32_t* __restrict__ fout, unsigned long long* __restrict__ counter) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < nin;
    const size_t c = live ? size_t(fin[q]) : 0;
    const int x = int(c % size_t(nx)), y = int(c / size_t(nx));
    int indeg = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        bool push = false;
        size_t n = 0;
        if (live) {
            const int xn = x + d1(k), yn = y + d2(k);
            if (xn >= 0 && xn < nx && yn >= 0 && yn < ny) {
                n = size_t(yn) * size_t(nx) + size_t(xn);
                const int16_t pn = P[n];
                if (!is_nodata_s(pn, nodata) && pn >= 0 && pn <= 8 && (pn - k == 4 || pn - k == -4)) {
                    indeg++;
                    push = (atomicCAS(&mark[n], 0, 1) == 0);
                }
            }
        }
        wave_append(push, uint32_t(n), fout, counter);
    }
    if (live) cnt[c] = indeg ? indeg : CNT_SOURCE;
}

__global__ __launch_bounds__(256) void ad8_outlet_seed_kernel(const int32_t* __restrict__ ox, const int32_t* __restrict__ oy, int nout,
                                                              int nx, int ny, int32_t* __restrict__ mark, uint32_t* __restrict__ fout,
                                                              unsigned long long* __restrict__ counter) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    bool push = false;
    uint32_t c = 0;
    if (o < nout) {
        const int x = ox[o], y = oy[o];
        if (x >= 0 && x < nx && y >= 0 && y < ny) {   // globalToLocal + isInPartition (src/commonLib.cpp:289-291)
            c = uint32_t(size_t(y) * size_t(nx) + size_t(x));
            push = (atomicCAS(&mark[c], 0, 1) == 0);
        }
    }
    wave_append(push, c, fout, counter);
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
__device__ __forceinline__ float ad8_evaluate(const int16_t* __restrict__ P, const float* __restrict__ Wt, float w_nodata,
                                              float* __restrict__ A, int nx, int ny, int x, int y, size_t idx, int16_t nodata, int contcheck) {
    float a;
    if (Wt) { const float w = Wt[idx]; a = is_nodata_f(w, w_nodata) ? TDX_AREA_NODATA : w; }   // nodata weight: keeps the initial -1
    else a = 1.0f;
    bool con = false;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        const int xn = x + d1(k), yn = y + d2(k);
        if (xn < 0 || xn >= nx || yn < 0 || yn >= ny) { con = true; continue; }
        const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
        const int16_t pn = P[n];
        if (is_nodata_s(pn, nodata)) { con = true; continue; }
        if (pn - k == 4 || pn - k == -4) {
            const float an = ld_agent(&A[n]);
            if (is_nodata_f(an, TDX_AREA_NODATA)) con = true;
            else a = a + an;
        }
    }
    if (con && contcheck == 1) a = TDX_AREA_NODATA;
    return a;
}

__global__ __launch_bounds__(256) void ad8_walk_kernel(const int16_t* __restrict__ P, const float* __restrict__ Wt, float w_nodata,
                                                       int nx, int ny, int16_t nodata, int contcheck, int32_t* __restrict__ cnt,
                                                       float* __restrict__ A, unsigned long long* __restrict__ nevaluated) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    unsigned long long done = 0;
    bool go = false;
    size_t idx = 0;
    if (x < nx && y < ny) {
        idx = size_t(y) * size_t(nx) + size_t(x);
        go = (cnt[idx] == CNT_SOURCE);          // participates and has no contributor
    }
    while (go) {
        const float a = ad8_evaluate(P, Wt, w_nodata, A, nx, ny, x, y, idx, nodata, contcheck);
        st_agent(&A[idx], a);
        done++;
        go = false;
        const int16_t k = P[idx];
        if (k >= 1 && k <= 8) {
            const int xn = x + d1(k), yn = y + d2(k);
            if (xn >= 0 && xn < nx && yn >= 0 && yn < ny) {
                const size_t n = size_t(yn) * size_t(nx) + size_t(xn);
                drain_stores();                 // value must be at the coherence point before the counter moves
                const int32_t old = __hip_atomic_fetch_sub(&cnt[n], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == 1) { x = xn; y = yn; idx = n; go = true; }   // I was the last contributor
            }
        }
    }
    (void)done; (void)nevaluated;
}

}  // namespace

extern "C" int tdx_aread8_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata,
                              const float* d_w, float w_nodata, int contcheck,
                              const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                              float* d_ad8, tdx_stats* stats) {
    if (!ctx || !d_p || !d_ad8 || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_aread8_dev: bad argument");
    if (nx > 0x7fffffff || ny > 0x7fffffff || uint64_t(nx) * uint64_t(ny) > 0xffffffffull)
        return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    if (n_outlets >= 0 && n_outlets > 0 && (!outlet_x || !outlet_y)) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_aread8_dev: outlets missing");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int inx = int(nx), iny = int(ny);
    const size_t n = size_t(nx) * size_t(ny);
    int32_t* cnt = static_cast<int32_t*>(ctx->scratch(TDX_S_A, n * 4));
    if (!cnt) return TDX_ERR_NOMEM;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    const dim3 grid2d((inx + 63) / 64, (iny + 3) / 4);

    ctx->begin_call(stats);
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    if (n_outlets < 0) {
        TdxSpan sp(ctx, TDX_K_STENCIL);
        hipLaunchKernelGGL(ad8_setup_kernel, grid2d, dim3(256), 0, s, d_p, inx, iny, p_nodata, cnt, d_ad8);
        if (stats) stats->launches[TDX_K_STENCIL]++;
    } else {
        // upstream closure of the outlets
        TdxSpan sp(ctx, TDX_K_BFS);
        int32_t* mark = static_cast<int32_t*>(ctx->scratch(TDX_S_B, n * 4));
        uint32_t* fa = static_cast<uint32_t*>(ctx->scratch(TDX_S_C, n * 4));
        uint32_t* fb = static_cast<uint32_t*>(ctx->scratch(TDX_S_D, n * 4));
        int32_t* d_ox = static_cast<int32_t*>(ctx->scratch(TDX_S_E, size_t(n_outlets ? n_outlets : 1) * 4));
        int32_t* d_oy = static_cast<int32_t*>(ctx->scratch(TDX_S_F, size_t(n_outlets ? n_outlets : 1) * 4));
        if (!mark || !fa || !fb || !d_ox || !d_oy) return TDX_ERR_NOMEM;
        hipLaunchKernelGGL(fill_i32_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, cnt, CNT_NOT_PART, n);
        hipLaunchKernelGGL(fill_f32_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_ad8, TDX_AREA_NODATA, n);
        TDX_HIP_CHECK(ctx, hipMemsetAsync(mark, 0, n * 4, s));
        unsigned long long ncur = 0;
        if (n_outlets > 0) {
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_ox, outlet_x, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_oy, outlet_y, size_t(n_outlets) * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(ad8_outlet_seed_kernel, dim3(tdx_blocks_for(size_t(n_outlets), 256)), dim3(256), 0, s, d_ox, d_oy, int(n_outlets),
                               inx, iny, mark, fa, d_cnt);
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
            ncur = ctx->h_mail[0];
        }
        uint32_t *cur = fa, *nxt = fb;
        while (ncur > 0) {
            TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(ad8_outlet_expand_kernel, dim3(tdx_blocks_for(ncur, 256)), dim3(256), 0, s, d_p, inx, iny, p_nodata, cur, ncur,
                               cnt, mark, nxt, d_cnt);
            TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
            ncur = ctx->h_mail[0];
            std::swap(cur, nxt);
            if (stats) stats->launches[TDX_K_BFS]++;
        }
        TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s));
    }
    {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        hipLaunchKernelGGL(ad8_walk_kernel, grid2d, dim3(256), 0, s, d_p, d_w, w_nodata, inx, iny, p_nodata, contcheck, cnt, d_ad8,
                           stats ? d_cnt + 4 : nullptr);
        if (stats) stats->launches[TDX_K_ACCUM]++;
    }
    if (stats) TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_cnt, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* st = stats;
    ctx->end_call();
    if (st) { st->cells_evaluated = int64_t(ctx->h_mail[4]); st->rounds = 1; }
    return TDX_OK;
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
