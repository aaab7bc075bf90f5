// Flat resolution shared by D8FlowDir and DinfFlowDir: the incfall / incrise relaxations of
// resolveflats() (src/d8.cpp:509-646, src/dinf.cpp:650-787) as frontier breadth-first sweeps.
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
#include <vector>

#include "context.hpp"
#include "device_common.hpp"

struct FlatLevels { int T; int Tr; int has_pits; };
struct FlatBuffers { int32_t* lvl; int32_t* rq; uint32_t* fa; uint32_t* fb; uint32_t* s2; uint32_t* ra; };

// elev2 + s as the reference's int16 arithmetic leaves it (src/d8.cpp:545,640-645)
__host__ __device__ __forceinline__ int16_t flat_elev2(int lvl, int rq, FlatLevels fl) {
    const int e = (lvl < 0) ? 1 : (lvl > 0 ? lvl : 1 + fl.T);
    const int s = (rq > 0) ? (fl.Tr - rq + 1) : 0;
    return (int16_t)(e + s);
}

namespace flatk {
using namespace tdxk;

// per flat cell: level-1 ("low") test, level-2 quirk source test, "higher neighbour" test.
// 8 list entries per lane; each of the three output lists is reserved with one atomic per block.
constexpr int CLASSIFY_ITEMS = 8;
template <class Traits>
__global__ __launch_bounds__(256) void classify_kernel(Traits tr, const float* __restrict__ Z, int nx, const uint32_t* __restrict__ list,
                                                       unsigned long long nq, int32_t* __restrict__ lvl, int32_t* __restrict__ rq,
                                                       uint32_t* __restrict__ f1, uint32_t* __restrict__ s2, uint32_t* __restrict__ r1,
                                                       unsigned long long* __restrict__ counters) {
    const unsigned long long base = (unsigned long long)blockIdx.x * (256 * CLASSIFY_ITEMS) + threadIdx.x;
    unsigned mlow = 0, mquirk = 0, mhigh = 0;
#pragma unroll
    for (int i = 0; i < CLASSIFY_ITEMS; i++) {
        const unsigned long long q = base + (unsigned long long)i * 256;
        if (q < nq) {
            const size_t c = list[q];
            const float z0 = Z[c];
            bool low = false, quirk = false, higher = false;
#pragma unroll
            for (int k = 1; k <= 8; k++) {
                const size_t n = size_t(ptrdiff_t(c) + ptrdiff_t(d2(k)) * nx + d1(k));
                const float zd = z0 - Z[n];
                if (zd < 0) higher = true;
                if (!tr.dont_cross(c, nx, k)) {
                    if (zd >= 0 && tr.has_direction(n)) low = true;
                    else if (zd == 0 && lvl[n] < 0) quirk = true;
                }
            }
            if (low) { lvl[c] = 1; mlow |= 1u << i; }
            else if (quirk) mquirk |= 1u << i;
            if (higher) { rq[c] = 1; mhigh |= 1u << i; }
        }
    }
    unsigned long long p0 = block_reserve(unsigned(__popc(mlow)), counters + 0);
    unsigned long long p1 = block_reserve(unsigned(__popc(mquirk)), counters + 1);
    unsigned long long p2 = block_reserve(unsigned(__popc(mhigh)), counters + 2);
#pragma unroll
    for (int i = 0; i < CLASSIFY_ITEMS; i++) {
        const unsigned bit = 1u << i;
        if ((mlow | mquirk | mhigh) & bit) {
            const uint32_t c = list[base + (unsigned long long)i * 256];
            if (mlow & bit) f1[p0++] = c;
            if (mquirk & bit) s2[p1++] = c;
            if (mhigh & bit) r1[p2++] = c;
        }
    }
}

// ---- level expansion kernels -------------------------------------------------------------------
// All frontiers of one BFS live back to back in ONE buffer (every flat cell is claimed at most once):
// level t occupies buf[tails[t-1] .. tails[t]).  The kernel that produces level `st` reads level st-1,
// appends claimed cells at the global `tail`, and its LAST block to finish publishes tails[st] = tail.
// No host round trip per level: the host enqueues levels in batches and reads back a slice of tails[].
struct LevelCtl {
    unsigned long long* tails;   // [max levels]
    unsigned long long* tail;    // running append position
    unsigned int* done;          // block-completion ticket
};

__device__ __forceinline__ void publish_level(const LevelCtl& ctl, int st) {
    // every append of this block is an atomic on `tail` whose result was consumed, so it is complete
    // here; tails[] and the reset ticket are read by the NEXT launch (kernel boundary = visibility).
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = atomicAdd(ctl.done, 1u);
        if (t == gridDim.x - 1) {
            ctl.tails[st] = atomicAdd(ctl.tail, 0ull);
            atomicExch(ctl.done, 0u);
        }
    }
}

// incfall: cells stopping in sweep `st` are the unvisited flat cells with an equal, non-crossing
// neighbour that stopped in sweep st-1
template <class Traits>
__global__ __launch_bounds__(256) void fall_expand_kernel(Traits tr, const float* __restrict__ Z, int nx, uint32_t* __restrict__ buf,
                                                          LevelCtl ctl, int st, int32_t* __restrict__ lvl) {
    const unsigned long long begin = ctl.tails[st - 2], end = ctl.tails[st - 1];
    const unsigned long long nin = end - begin;
    for (unsigned long long base = (unsigned long long)blockIdx.x * 256; base < nin; base += (unsigned long long)gridDim.x * 256) {
        const unsigned long long q = base + threadIdx.x;
        const bool live = q < nin;
        const size_t n = live ? size_t(buf[begin + q]) : 0;
        const float zn = live ? Z[n] : 0.f;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            bool claim = false;
            size_t c = 0;
            if (live) {
                c = size_t(ptrdiff_t(n) + ptrdiff_t(d2(k)) * nx + d1(k));
                if (lvl[c] == 0 && (Z[c] - zn) == 0) {
                    const int kc = ((k + 3) & 7) + 1;            // direction from c back to n
                    if (!tr.dont_cross(c, nx, kc)) claim = (atomicCAS(&lvl[c], 0, st) == 0);
                }
            }
            wave_append(claim, uint32_t(c), buf, ctl.tail);
        }
    }
    publish_level(ctl, st);
}

// level-2 quirk sources (appended to level 2 before the level-2 expansion publishes)
static __global__ __launch_bounds__(256) void fall_s2_kernel(const uint32_t* __restrict__ s2, unsigned long long ns2, int32_t* __restrict__ lvl,
                                                             uint32_t* __restrict__ buf, unsigned long long* __restrict__ tail) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    bool claim = false;
    uint32_t c = 0;
    if (q < ns2) { c = s2[q]; claim = (atomicCAS(&lvl[c], 0, 2) == 0); }
    wave_append(claim, c, buf, tail);
}

// incrise: any unmarked flat 8-neighbour of a cell marked in the previous sweep
static __global__ __launch_bounds__(256) void rise_expand_kernel(int nx, uint32_t* __restrict__ buf, LevelCtl ctl, int q_level,
                                                                 int32_t* __restrict__ rq) {
    const unsigned long long begin = ctl.tails[q_level - 2], end = ctl.tails[q_level - 1];
    const unsigned long long nin = end - begin;
    for (unsigned long long base = (unsigned long long)blockIdx.x * 256; base < nin; base += (unsigned long long)gridDim.x * 256) {
        const unsigned long long q = base + threadIdx.x;
        const bool live = q < nin;
        const size_t n = live ? size_t(buf[begin + q]) : 0;
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            bool claim = false;
            size_t c = 0;
            if (live) {
                c = size_t(ptrdiff_t(n) + ptrdiff_t(d2(k)) * nx + d1(k));
                if (rq[c] == 0) claim = (atomicCAS(&rq[c], 0, q_level) == 0);
            }
            wave_append(claim, uint32_t(c), buf, ctl.tail);
        }
    }
    publish_level(ctl, q_level);
}

static __global__ void init_levels_kernel(LevelCtl ctl, unsigned long long n1) {
    ctl.tails[0] = 0ull;
    ctl.tails[1] = n1;
    *ctl.tail = n1;
    *ctl.done = 0u;
}

static __global__ __launch_bounds__(256) void reset_q_kernel(const uint32_t* __restrict__ list, unsigned long long nq, int32_t* __restrict__ lvl,
                                                      int32_t* __restrict__ rq) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    lvl[list[q]] = 0;
    rq[list[q]] = 0;
}

static __global__ __launch_bounds__(256) void overwrite_elev_kernel(size_t n, const int32_t* __restrict__ lvl, const int32_t* __restrict__ rq,
                                                             FlatLevels fl, float* __restrict__ Zout) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) Zout[i] = (float)flat_elev2(lvl[i], rq[i], fl);
}

}  // namespace flatk

static inline int flats_read_counters(tdx_context* ctx, int nwords) {
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail, size_t(nwords) * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}

static inline int flats_reset_markers(tdx_context* ctx, size_t n, const uint32_t* qlist, unsigned long long nq, int32_t* lvl, int32_t* rq) {
    TDX_HIP_CHECK(ctx, hipMemsetAsync(lvl, 0xFF, n * 4, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemsetAsync(rq, 0xFF, n * 4, ctx->stream));
    hipLaunchKernelGGL(flatk::reset_q_kernel, dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, ctx->stream, qlist, nq, lvl, rq);
    return TDX_OK;
}

static inline int flats_overwrite_elevation(tdx_context* ctx, size_t n, const int32_t* lvl, const int32_t* rq, FlatLevels fl, float* zout) {
    TdxSpan sp(ctx, TDX_K_MISC);
    hipLaunchKernelGGL(flatk::overwrite_elev_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, lvl, rq, fl, zout);
    return TDX_OK;
}

constexpr int TDX_MAX_FLAT_LEVELS = 1 << 17;

// Enqueues level expansions in growing batches until a level comes back empty.  `launch(st)` enqueues
// the kernel(s) that produce level st.  Returns the last non-empty level (>= first_level-1) in *last.
// The grid of a batch is sized from the largest frontier of the previous batch (frontier sizes change
// slowly from level to level; the kernels grid-stride, so an undersized grid is only slower).
template <class Launch>
static int flats_run_levels(tdx_context* ctx, unsigned long long* d_tails, int first_level, unsigned long long first_size, Launch launch,
                            int* last, int64_t* launches) {
    int st = first_level, batch = 32;
    unsigned long long prev_tail = 0, widest = first_size;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_tails + (first_level - 1), sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    prev_tail = ctx->h_mail[0];
    std::vector<unsigned long long> h;
    for (;;) {
        if (st + batch >= TDX_MAX_FLAT_LEVELS) return tdx_fail(ctx, TDX_ERR_ARG, "flat resolution deeper than the reference's int16 level counter allows");
        const unsigned grid = unsigned(std::min<unsigned long long>(2048ull, std::max<unsigned long long>(8ull, (2 * widest + 255) / 256)));
        for (int b = 0; b < batch; b++) launch(st + b, grid);
        *launches += batch;
        h.resize(size_t(batch));
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), d_tails + st, size_t(batch) * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        widest = 0;
        for (int b = 0; b < batch; b++) {
            if (h[size_t(b)] == prev_tail) { *last = st + b - 1; return TDX_OK; }   // level st+b is empty
            widest = std::max(widest, h[size_t(b)] - prev_tail);
            prev_tail = h[size_t(b)];
        }
        st += batch;
        if (batch < 512) batch *= 2;
    }
}

// Runs classify + both BFS sweeps for the flat queue `qlist`; on return lvl/rq hold the levels and
// *out the sweep counts.  d_mail words: 0 level-1 count, 1 s2 count, 2 rise level-1 count, 3 tail, 4 ticket.
template <class Traits>
static int flats_bfs(tdx_context* ctx, Traits tr, const float* Z, int nx, int /*ny*/, const uint32_t* qlist, unsigned long long nq,
                     FlatBuffers b, FlatLevels* out, tdx_stats* stats) {
    hipStream_t s = ctx->stream;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    unsigned long long* d_tails = static_cast<unsigned long long*>(ctx->scratch(TDX_S_K, size_t(TDX_MAX_FLAT_LEVELS) * sizeof(unsigned long long)));
    if (!d_tails) return TDX_ERR_NOMEM;
    TdxSpan sp(ctx, TDX_K_BFS);
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((flatk::classify_kernel<Traits>), dim3(tdx_blocks_for(nq, 256 * flatk::CLASSIFY_ITEMS)), dim3(256), 0, s, tr, Z, nx, qlist,
                       nq, b.lvl, b.rq, b.fa, b.s2, b.ra, d_cnt);
    int rc = flats_read_counters(ctx, 3);
    if (rc != TDX_OK) return rc;
    const unsigned long long n1 = ctx->h_mail[0], ns2 = ctx->h_mail[1], r1 = ctx->h_mail[2];
    int64_t launches = 1;
    flatk::LevelCtl ctl{d_tails, d_cnt + 3, reinterpret_cast<unsigned int*>(d_cnt + 4)};

    // ---- incfall ----  frontier buffer fa; level 1 = fa[0..n1)
    int L = (n1 > 0) ? 1 : 0;
    unsigned long long stopped = n1;
    if (n1 > 0 || ns2 > 0) {
        hipLaunchKernelGGL(flatk::init_levels_kernel, dim3(1), dim3(1), 0, s, ctl, n1);
        int lastlvl = 1;
        rc = flats_run_levels(ctx, d_tails, 2, std::max(n1, ns2), [&](int st, unsigned grid) {
            if (st == 2 && ns2 > 0)
                hipLaunchKernelGGL(flatk::fall_s2_kernel, dim3(tdx_blocks_for(ns2, 256)), dim3(256), 0, s, b.s2, ns2, b.lvl, b.fa, ctl.tail);
            hipLaunchKernelGGL((flatk::fall_expand_kernel<Traits>), dim3(grid), dim3(256), 0, s, tr, Z, nx, b.fa, ctl, st, b.lvl);
        }, &lastlvl, &launches);
        if (rc != TDX_OK) return rc;
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail, d_tails + lastlvl, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
        stopped = ctx->h_mail[0];
        L = (lastlvl >= 2) ? lastlvl : L;
    }
    out->T = (n1 == nq) ? 1 : ((L > 1 ? L : 1) + 1);
    out->has_pits = (stopped < nq) ? 1 : 0;
    if (stats) stats->levels_fall += L;

    // ---- incrise ----  frontier buffer ra; level 1 = ra[0..r1)
    int Qmax = (r1 > 0) ? 1 : 0;
    if (r1 > 0) {
        hipLaunchKernelGGL(flatk::init_levels_kernel, dim3(1), dim3(1), 0, s, ctl, r1);
        int lastlvl = 1;
        rc = flats_run_levels(ctx, d_tails, 2, r1, [&](int q, unsigned grid) {
            hipLaunchKernelGGL(flatk::rise_expand_kernel, dim3(grid), dim3(256), 0, s, nx, b.ra, ctl, q, b.rq);
        }, &lastlvl, &launches);
        if (rc != TDX_OK) return rc;
        Qmax = lastlvl;
    }
    out->Tr = Qmax + 1;
    if (stats) { stats->levels_rise += Qmax; stats->launches[TDX_K_BFS] += launches; }
    return TDX_OK;
}
