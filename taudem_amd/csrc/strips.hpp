// Row strips: halo rows and the host-provided exchange (tdx_comm, include/taudem_amd.h).
// Replaces linearpart<T>::share() / passBorders() / ringTerm() (src/linearpart.h:194-360).
//
// A Strip describes the rows of a device array that the calling rank may write ("owned") and whether a
// halo row above / below belongs to a neighbouring rank.  For the single-strip entry points the array
// has no halo rows at all (own = [0, ny)): kernels read rows outside the array as nodata, which is
// what a halo row beyond the global raster holds in a strip array.
#pragma once
#include <chrono>
#include <thread>

#include "context.hpp"
#include "device_common.hpp"

struct Strip {
    int nx = 0;
    int ny_arr = 0;          // rows of the device arrays
    int y0 = 0, y1 = 0;      // owned rows [y0, y1)
    bool up = false, down = false;   // a neighbouring rank owns the halo row above / below
    const tdx_comm* comm = nullptr;
    bool multi() const { return comm && comm->size > 1; }
};

static inline Strip strip_single(int nx, int ny) {
    Strip s; s.nx = nx; s.ny_arr = ny; s.y0 = 0; s.y1 = ny; return s;
}
// marks the context with the rank and the stage of the call that is starting (time-out messages, TDX_COMM_TRACE=1)
static inline void strip_mark(tdx_context* ctx, const Strip& st, const char* stage) {
    ctx->stage = stage;
    ctx->comm_rank = st.comm ? st.comm->rank : 0;
    ctx->comm_size = st.comm ? st.comm->size : 1;
    ctx->comm_ordered = st.comm && (st.comm->flags & TDX_COMM_STREAM_ORDERED) != 0;
}
static inline Strip strip_from_comm(const tdx_comm* comm, int nx, int ny_local) {
    Strip s; s.nx = nx; s.ny_arr = ny_local + 2; s.y0 = 1; s.y1 = ny_local + 1; s.comm = comm;
    if (comm && comm->size > 1) { s.up = comm->rank > 0; s.down = comm->rank < comm->size - 1; }
    return s;
}

namespace stripk {
template <class T>
__device__ __forceinline__ bool same_bits(T a, T b) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(uint32_t, a) == __builtin_bit_cast(uint32_t, b);
    else if constexpr (sizeof(T) == 2) return __builtin_bit_cast(uint16_t, a) == __builtin_bit_cast(uint16_t, b);
    else return a == b;
}
template <class T>
__global__ void fill_row_kernel(T* row, T v, int nx) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x < nx) row[x] = v;
}
// halo <- recv where they differ; counts the differing cells and raises the activation flag of the
// tiles that see the changed cell (tile rows tr0 / tr1, columns x-1 .. x+1)
template <class T>
__global__ void merge_row_kernel(T* halo, const T* recv, int nx, unsigned long long* nchanged, uint32_t* tile_flags, int tiles_x, int tr0, int tr1) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    bool ch = false;
    if (x < nx) {
        const T a = recv[x];
        if (!same_bits(a, halo[x])) {   // bit patterns: a NaN boundary cell must not count as "changed" in every round
            halo[x] = a;
            ch = true;
            if (tile_flags) {
                const int t0 = (x > 0 ? x - 1 : 0) / 64, t1 = (x + 1 < nx ? x + 1 : nx - 1) / 64;
                for (int t = t0; t <= t1; t++) {
                    tile_flags[tr0 * tiles_x + t] = 2u;   // tilek::FLAG_FULL: a row INSIDE the tile's area changed
                    tile_flags[tr1 * tiles_x + t] = 2u;
                }
            }
        }
    }
    const unsigned long long b = __ballot(ch);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(nchanged, (unsigned long long)__popcll(b));
}
}  // namespace stripk

// Waits for the context's stream like hipStreamSynchronize, but BOUNDED when other ranks are involved: a collective whose partner never
// arrives (a rank that failed, took another branch or stalled) would hang the job with no hint; after TDX_COMM_TIMEOUT seconds (default
// 600) the call fails and says which rank waited for what, in which stage, after how many exchanges.
static inline int strip_wait(tdx_context* ctx, const Strip& st, const char* what) {
    if (!st.multi()) { TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); return TDX_OK; }
    static const double limit_s = getenv("TDX_COMM_TIMEOUT") ? atof(getenv("TDX_COMM_TIMEOUT")) : 600.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) return TDX_OK;
        if (e != hipErrorNotReady) { TDX_HIP_CHECK(ctx, e); }
        if (spins > 2000) {   // ~100 us of pure polling first: the common wait is one small kernel + one row copy
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            if ((spins & 1023u) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
                char msg[320];
                snprintf(msg, sizeof msg, "rank %d of %d stalled in %s: %s did not complete within %.0f s (exchange %lld, all-reduce %lld of this call); a "
                         "neighbouring rank has probably failed or left the loop - TDX_COMM_TRACE=1 prints every rank's counts per stage",
                         ctx->comm_rank, ctx->comm_size, ctx->stage, what, limit_s, (long long)ctx->comm_exchanges, (long long)ctx->comm_allreduces);
                return tdx_fail(ctx, TDX_ERR_HIP, msg);
            }
        }
    }
}

// stream-ordered transport (RCCL): exchange() enqueues on the context's stream, no synchronisation around it
static inline bool strip_ordered(const Strip& st) { return st.comm && (st.comm->flags & TDX_COMM_STREAM_ORDERED) != 0; }
static inline int strip_pre_exchange_sync(tdx_context* ctx, const Strip& st) {
    if (!strip_ordered(st)) return strip_wait(ctx, st, "the work before an exchange");
    return TDX_OK;
}

// Sum / max over the ranks of `count` int64 counters that live in DEVICE memory (d_v, written by work already enqueued on
// the stream); the result lands in host_out.  With a transport that reduces device values in stream order (RCCL) this is
// one collective and ONE synchronisation; otherwise device -> host, synchronise, host all-reduce.
static inline int strip_allreduce_device(tdx_context* ctx, const Strip& st, unsigned long long* d_v, int count, int op, int64_t* host_out) {
    hipStream_t s = ctx->stream;
    if (st.multi()) { ctx->comm_allreduces++; ctx->comm_allreduces_total++; }
    if (st.multi() && st.comm->allreduce_dev) {
        ctx->seg_end(1);   // (segment trace: the collective and the read-back of its result lie between two segments)
        if (st.comm->allreduce_dev(st.comm->user, reinterpret_cast<int64_t*>(d_v), count, op) != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm allreduce_dev failed");
        TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail + TDX_MAIL_STRIP_REDUCE, d_v, size_t(count) * 8, hipMemcpyDeviceToHost, s));
        if (int rcw = strip_wait(ctx, st, "a device all-reduce (termination vote)")) return rcw;
        for (int i = 0; i < count; i++) host_out[i] = int64_t(ctx->h_mail[TDX_MAIL_STRIP_REDUCE + i]);
        ctx->seg_begin();
        return TDX_OK;
    }
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail + TDX_MAIL_STRIP_REDUCE, d_v, size_t(count) * 8, hipMemcpyDeviceToHost, s));
    if (int rcw = strip_wait(ctx, st, "the counters of a vote")) return rcw;
    for (int i = 0; i < count; i++) host_out[i] = int64_t(ctx->h_mail[TDX_MAIL_STRIP_REDUCE + i]);
    if (!st.multi()) return TDX_OK;
    ctx->seg_end(1);
    if (st.comm->allreduce(st.comm->user, host_out, count, op) != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm allreduce failed" + (g_tdx_thread_error.empty() ? std::string() : ": " + g_tdx_thread_error));
    ctx->seg_begin();
    return TDX_OK;
}

static inline int strip_allreduce(tdx_context* ctx, const Strip& st, int64_t* v, int count, int op) {
    if (!st.multi()) return TDX_OK;
    ctx->comm_allreduces++; ctx->comm_allreduces_total++;
    ctx->seg_end(1);
    if (st.comm->allreduce(st.comm->user, v, count, op) != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm allreduce failed" + (g_tdx_thread_error.empty() ? std::string() : ": " + g_tdx_thread_error));
    ctx->seg_begin();
    return TDX_OK;
}

// Halo rows of `arr` <- the neighbours' boundary rows (or `outside` beyond the raster).  With
// tile_flags != nullptr only differing cells are written, the tiles that see them are flagged, and
// *nchanged (host) receives this rank's number of changed halo cells - or, with global_vote, the number summed over
// all ranks (the termination vote of the caller's loop, fused into this exchange: one collective, one synchronisation).
// extra_vote (with global_vote): this rank's own "not done yet" (0 / 1), counted into the vote next to its changed halo cells - for callers that
// exchange BEFORE their strip-local schedule has run dry (the dependency sweeps' bounded rounds between two exchanges).
template <class T>
static int strip_exchange(tdx_context* ctx, const Strip& st, T* arr, T outside, uint32_t* tile_flags = nullptr, int tiles_x = 0, int64_t* nchanged = nullptr,
                          bool global_vote = false, int extra_vote = 0) {
    if (nchanged) *nchanged = 0;
    if (st.ny_arr == st.y1 - st.y0) return TDX_OK;   // single-strip array without halo rows
    hipStream_t s = ctx->stream;
    const size_t nx = size_t(st.nx), bytes = nx * sizeof(T);
    const unsigned g = tdx_blocks_for(nx, 256);
    if (!st.up) hipLaunchKernelGGL(stripk::fill_row_kernel<T>, dim3(g), dim3(256), 0, s, arr + size_t(st.y0 - 1) * nx, outside, st.nx);
    if (!st.down) hipLaunchKernelGGL(stripk::fill_row_kernel<T>, dim3(g), dim3(256), 0, s, arr + size_t(st.y1) * nx, outside, st.nx);
    if (!st.multi()) return TDX_OK;
    const tdx_comm* c = st.comm;
    if (bytes > c->capacity) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_comm buffers smaller than one raster row");
    if (st.up) TDX_HIP_CHECK(ctx, hipMemcpyAsync(c->send_up, arr + size_t(st.y0) * nx, bytes, hipMemcpyDeviceToDevice, s));
    if (st.down) TDX_HIP_CHECK(ctx, hipMemcpyAsync(c->send_down, arr + size_t(st.y1 - 1) * nx, bytes, hipMemcpyDeviceToDevice, s));
    if (int rcs = strip_pre_exchange_sync(ctx, st)) return rcs;
    ctx->comm_exchanges++; ctx->comm_exchanges_total++;
    ctx->seg_end(0);
    const int rce = c->exchange(c->user, bytes);
    if (rce == 0) ctx->seg_begin();
    if (rce != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm exchange failed" + (g_tdx_thread_error.empty() ? std::string() : ": " + g_tdx_thread_error));
    if (!tile_flags && !nchanged) {
        if (st.up) TDX_HIP_CHECK(ctx, hipMemcpyAsync(arr + size_t(st.y0 - 1) * nx, c->recv_up, bytes, hipMemcpyDeviceToDevice, s));
        if (st.down) TDX_HIP_CHECK(ctx, hipMemcpyAsync(arr + size_t(st.y1) * nx, c->recv_down, bytes, hipMemcpyDeviceToDevice, s));
        return TDX_OK;
    }
    unsigned long long* d_n = reinterpret_cast<unsigned long long*>(ctx->d_mail) + TDX_MAIL_STRIP_CHANGED;
    TDX_HIP_CHECK(ctx, hipMemsetAsync(d_n, 0, sizeof(unsigned long long), s));
    if (extra_vote) TDX_HIP_CHECK(ctx, hipMemsetAsync(d_n, 1, 1, s));   // (little-endian: the counter starts at 1)
    if (st.up) {
        const int yh = st.y0 - 1;
        hipLaunchKernelGGL(stripk::merge_row_kernel<T>, dim3(g), dim3(256), 0, s, arr + size_t(yh) * nx, static_cast<const T*>(c->recv_up), st.nx, d_n,
                           tile_flags, tiles_x, yh / 64, (yh + 1) / 64);
    }
    if (st.down) {
        const int yh = st.y1;
        hipLaunchKernelGGL(stripk::merge_row_kernel<T>, dim3(g), dim3(256), 0, s, arr + size_t(yh) * nx, static_cast<const T*>(c->recv_down), st.nx, d_n,
                           tile_flags, tiles_x, (yh - 1) / 64, yh / 64);
    }
    if (global_vote) {
        int64_t total = 0;
        if (int rcv = strip_allreduce_device(ctx, st, d_n, 1, TDX_OP_SUM, &total)) return rcv;
        if (nchanged) *nchanged = total;
        return TDX_OK;
    }
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail + TDX_MAIL_STRIP_CHANGED, d_n, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    if (int rcw = strip_wait(ctx, st, "a halo exchange")) return rcw;
    if (nchanged) *nchanged = int64_t(ctx->h_mail[TDX_MAIL_STRIP_CHANGED]);
    return TDX_OK;
}

// Raw neighbour exchange of two caller-chosen device buffers: `up_src` goes to the rank above, `down_src`
// to the rank below; what they sent towards this rank lands in up_dst (from above) / down_dst (from
// below).  Missing neighbours are skipped (their destination is left untouched).
static inline int strip_exchange_buffers(tdx_context* ctx, const Strip& st, const void* up_src, const void* down_src, void* up_dst, void* down_dst,
                                         size_t bytes) {
    if (!st.multi()) return TDX_OK;
    hipStream_t s = ctx->stream;
    const tdx_comm* c = st.comm;
    if (bytes > c->capacity) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_comm buffers smaller than one exchange row");
    if (st.up) TDX_HIP_CHECK(ctx, hipMemcpyAsync(c->send_up, up_src, bytes, hipMemcpyDeviceToDevice, s));
    if (st.down) TDX_HIP_CHECK(ctx, hipMemcpyAsync(c->send_down, down_src, bytes, hipMemcpyDeviceToDevice, s));
    if (int rcs = strip_pre_exchange_sync(ctx, st)) return rcs;
    ctx->comm_exchanges++; ctx->comm_exchanges_total++;
    ctx->seg_end(0);
    const int rce = c->exchange(c->user, bytes);
    if (rce == 0) ctx->seg_begin();
    if (rce != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm exchange failed" + (g_tdx_thread_error.empty() ? std::string() : ": " + g_tdx_thread_error));
    if (st.up && up_dst != c->recv_up) TDX_HIP_CHECK(ctx, hipMemcpyAsync(up_dst, c->recv_up, bytes, hipMemcpyDeviceToDevice, s));
    if (st.down && down_dst != c->recv_down) TDX_HIP_CHECK(ctx, hipMemcpyAsync(down_dst, c->recv_down, bytes, hipMemcpyDeviceToDevice, s));
    return TDX_OK;
}
