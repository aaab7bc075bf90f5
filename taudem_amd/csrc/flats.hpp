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

// per flat cell: level-1 ("low") test, level-2 quirk source test, "higher neighbour" test
template <class Traits>
__global__ __launch_bounds__(256) void classify_kernel(Traits tr, const float* __restrict__ Z, int nx, const uint32_t* __restrict__ list,
                                                       unsigned long long nq, int32_t* __restrict__ lvl, int32_t* __restrict__ rq,
                                                       uint32_t* __restrict__ f1, uint32_t* __restrict__ s2, uint32_t* __restrict__ r1,
                                                       unsigned long long* __restrict__ counters) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    bool low = false, quirk = false, higher = false;
    uint32_t ci = 0;
    if (q < nq) {
        ci = list[q];
        const size_t c = ci;
        const float z0 = Z[c];
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
        if (low) lvl[c] = 1;
        if (higher) rq[c] = 1;
    }
    wave_append(low, ci, f1, counters + 0);
    wave_append(quirk && !low, ci, s2, counters + 1);
    wave_append(higher, ci, r1, counters + 2);
}

// incfall level expansion: cells stopping in sweep `st` are the unvisited flat cells with an equal,
// non-crossing neighbour that stopped in sweep st-1
template <class Traits>
__global__ __launch_bounds__(256) void fall_expand_kernel(Traits tr, const float* __restrict__ Z, int nx, const uint32_t* __restrict__ fin,
                                                          unsigned long long nin, int st, int32_t* __restrict__ lvl,
                                                          uint32_t* __restrict__ fout, unsigned long long* __restrict__ counter) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < nin;
    const size_t n = live ? size_t(fin[q]) : 0;
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
        wave_append(claim, uint32_t(c), fout, counter);
    }
}

// level-2 quirk sources
__global__ __launch_bounds__(256) void fall_s2_kernel(const uint32_t* __restrict__ s2, unsigned long long ns2, int32_t* __restrict__ lvl,
                                                      uint32_t* __restrict__ fout, unsigned long long* __restrict__ counter) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    bool claim = false;
    uint32_t c = 0;
    if (q < ns2) { c = s2[q]; claim = (atomicCAS(&lvl[c], 0, 2) == 0); }
    wave_append(claim, c, fout, counter);
}

// incrise level expansion: any unmarked flat 8-neighbour of a cell marked in the previous sweep
__global__ __launch_bounds__(256) void rise_expand_kernel(int nx, const uint32_t* __restrict__ fin, unsigned long long nin, int q_level,
                                                          int32_t* __restrict__ rq, uint32_t* __restrict__ fout,
                                                          unsigned long long* __restrict__ counter) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < nin;
    const size_t n = live ? size_t(fin[q]) : 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        bool claim = false;
        size_t c = 0;
        if (live) {
            c = size_t(ptrdiff_t(n) + ptrdiff_t(d2(k)) * nx + d1(k));
            if (rq[c] == 0) claim = (atomicCAS(&rq[c], 0, q_level) == 0);
        }
        wave_append(claim, uint32_t(c), fout, counter);
    }
}

__global__ __launch_bounds__(256) void reset_q_kernel(const uint32_t* __restrict__ list, unsigned long long nq, int32_t* __restrict__ lvl,
                                                      int32_t* __restrict__ rq) {
    const unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    lvl[list[q]] = 0;
    rq[list[q]] = 0;
}

__global__ __launch_bounds__(256) void overwrite_elev_kernel(size_t n, const int32_t* __restrict__ lvl, const int32_t* __restrict__ rq,
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

// Runs classify + both BFS sweeps for the flat queue `qlist`; on return lvl/rq hold the levels and
// *out the sweep counts.  d_mail words: 0 fall level-1 count / running counter, 1 s2 count, 2 rise-1 count.
template <class Traits>
static int flats_bfs(tdx_context* ctx, Traits tr, const float* Z, int nx, int /*ny*/, const uint32_t* qlist, unsigned long long nq,
                     FlatBuffers b, FlatLevels* out, tdx_stats* stats) {
    hipStream_t s = ctx->stream;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(ctx->d_mail);
    TdxSpan sp(ctx, TDX_K_BFS);
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((flatk::classify_kernel<Traits>), dim3(tdx_blocks_for(nq, 256)), dim3(256), 0, s, tr, Z, nx, qlist, nq, b.lvl, b.rq,
                       b.fa, b.s2, b.ra, d_cnt);
    int rc = flats_read_counters(ctx, 3);
    if (rc != TDX_OK) return rc;
    const unsigned long long n1 = ctx->h_mail[0], ns2 = ctx->h_mail[1], r1 = ctx->h_mail[2];
    int64_t launches = 1;

    // ---- incfall ----
    unsigned long long stopped = n1, ncur = n1;
    int L = (n1 > 0) ? 1 : 0;
    uint32_t *cur = b.fa, *nxt = b.fb;
    for (int st = 2;; st++) {
        if (ncur == 0 && !(st == 2 && ns2 > 0)) break;
        TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s));
        if (ncur > 0) {
            hipLaunchKernelGGL((flatk::fall_expand_kernel<Traits>), dim3(tdx_blocks_for(ncur, 256)), dim3(256), 0, s, tr, Z, nx, cur, ncur, st,
                               b.lvl, nxt, d_cnt);
            launches++;
        }
        if (st == 2 && ns2 > 0) {
            hipLaunchKernelGGL(flatk::fall_s2_kernel, dim3(tdx_blocks_for(ns2, 256)), dim3(256), 0, s, b.s2, ns2, b.lvl, nxt, d_cnt);
            launches++;
        }
        rc = flats_read_counters(ctx, 1);
        if (rc != TDX_OK) return rc;
        const unsigned long long nout = ctx->h_mail[0];
        if (nout == 0) break;
        L = st;
        stopped += nout;
        ncur = nout;
        std::swap(cur, nxt);
    }
    out->T = (n1 == nq) ? 1 : ((L > 1 ? L : 1) + 1);
    out->has_pits = (stopped < nq) ? 1 : 0;
    if (stats) stats->levels_fall += L;

    // ---- incrise ----  (fa/fb are free again; level-1 frontier is in ra)
    int Qmax = (r1 > 0) ? 1 : 0;
    ncur = r1;
    cur = b.ra; nxt = b.fa;
    uint32_t* spare = b.fb;
    for (int q = 2; ncur > 0; q++) {
        TDX_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(flatk::rise_expand_kernel, dim3(tdx_blocks_for(ncur, 256)), dim3(256), 0, s, nx, cur, ncur, q, b.rq, nxt, d_cnt);
        launches++;
        rc = flats_read_counters(ctx, 1);
        if (rc != TDX_OK) return rc;
        const unsigned long long nout = ctx->h_mail[0];
        if (nout == 0) break;
        Qmax = q;
        ncur = nout;
        uint32_t* t = cur; cur = nxt; nxt = (t == b.ra) ? spare : t;
    }
    out->Tr = Qmax + 1;
    if (stats) { stats->levels_rise += Qmax; stats->launches[TDX_K_BFS] += launches; }
    return TDX_OK;
}
