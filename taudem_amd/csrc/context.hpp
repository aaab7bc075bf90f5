// Per-GPU execution context shared by all stages: device, stream, error text, a reusable scratch
// arena (so that steady-state calls do not hipMalloc) and HIP-event timing per kernel class.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/taudem_amd.h"

#define TDX_NUM_KCLASS 8

// Layout of the two mailboxes (8-byte words; the same offsets in the pinned host copy and in the device copy).  Every user has its
// own words, so that a round batch still in flight on either stream, a strip vote and a stage counter can never land on each other.
enum {
    TDX_MAIL_STAGE = 0,             // [0, 32)    counters of the running stage (flat counts, list lengths, overflow counts)
    TDX_MAIL_DBG_RELAX = 32,        // [32, 48)   TDX_DEBUG_ROUNDS counters of the relaxation kernels
    TDX_MAIL_DBG_SWEEP = 48,        // [48, 64)   TDX_DEBUG_ROUNDS counters of the D-infinity sweep
    TDX_MAIL_STRIP_CHANGED = 64,    // [64, 72)   strip_exchange: changed halo cells
    TDX_MAIL_STRIP_REDUCE = 72,     // [72, 96)   strip_allreduce_device: values reduced over the ranks (up to 24)
    TDX_MAIL_VERIFY = 96,           // [96, 104)  sweep verifier: cells checked, mismatches, first mismatch
    TDX_MAIL_RUN_A = 128,           // [128, 256) RoundRunner on ctx->stream: two slots of TDX_MAIL_RUN_SLOT per-round counts
    TDX_MAIL_RUN_B = 256,           // [256, 384) RoundRunner on ctx->stream2
    TDX_MAIL_RUN_SLOT = 64,
    TDX_MAIL_WORDS = 384
};
static_assert(TDX_MAIL_RUN_A + 2 * TDX_MAIL_RUN_SLOT <= TDX_MAIL_RUN_B && TDX_MAIL_RUN_B + 2 * TDX_MAIL_RUN_SLOT <= TDX_MAIL_WORDS, "mailbox layout");

struct tdx_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // second stream for side-by-side relaxations (created on first use)
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_batch[4] = {nullptr, nullptr, nullptr, nullptr};   // round batches in flight (tile_relax.hpp), created on first use
    std::string err;
    int num_cus = 256;

    // ---- scratch arena: named slots that only grow; freed with the context ----
    struct Slot { void* p = nullptr; size_t bytes = 0; };
    std::vector<Slot> slots;
    void* scratch(int slot, size_t bytes);   // returns nullptr + sets err on failure
    // the cell sizes the D8 distance table resident in slot TDX_S_FACT was built from (tdx_build_fact_table: a repeated call with the same
    // geometry - every step of a pipeline over one raster - finds its table in place instead of a host loop, an upload and a stream synchronisation)
    std::vector<double> fact_dxc, fact_dyc;

    // pinned host mailbox for small device->host readbacks (counters, flags)
    uint64_t* h_mail = nullptr;              // TDX_MAIL_WORDS words, hipHostMalloc (layout: the TDX_MAIL_* offsets below)
    uint64_t* d_mail = nullptr;              // TDX_MAIL_WORDS words of device memory
    uint32_t run_seq = 0;                    // sequence number of the last batch of rounds enqueued on this context (tile_relax.hpp: RoundRunner::enqueue)

    // ---- timing ----
    struct Span { hipEvent_t a, b; int kclass; };
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::vector<Span> spans;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool timing = false;
    bool kernel_timing = false;               // option "kernel_timing"
    tdx_stats* cur_stats = nullptr;

    // ---- multi-strip diagnostics (strips.hpp): what this rank is doing, for time-out messages and TDX_COMM_TRACE=1 ----
    const char* stage = "";                   // tool stage of the running call ("pitremove", "d8flowdir" ...)
    int comm_rank = 0, comm_size = 1;
    bool comm_ordered = false;   // the call's transport enqueues its collectives on the stream (TDX_COMM_STREAM_ORDERED)
    int64_t comm_exchanges = 0, comm_allreduces = 0;   // of the running call
    int64_t comm_exchanges_total = 0, comm_allreduces_total = 0;   // since the context was created (tdx_context_comm_counters)

    // ---- segment trace (option "segment_trace"; scripts/project_8gpu.py): a strip run is a sequence of SEGMENTS of rank-local work, each ended by a
    // collective (halo exchange / all-reduce) or by the end of the call.  Every rank passes through the same sequence (the protocol is rank-symmetric),
    // so the critical path of a real N-GPU run is  sum over segments of (max over ranks of the segment's time) + collectives x their latency  - which
    // can be measured on ONE GPU: mode 2 serialises the ranks' segments through a process-wide token, so that each segment is timed alone on the device.
    struct Segment { const char* stage; const char* phase; int kind; float device_ms; float wall_ms; };   // kind: 0 exchange, 1 all-reduce, 2 end of call
    int seg_mode = 0;                         // 0 off, 1 timed, 2 timed + one rank on the device at a time
    bool seg_open = false, seg_token = false;
    hipEvent_t seg_ev0 = nullptr, seg_ev1 = nullptr;
    double seg_t0 = 0.0;
    const char* phase = "";                   // sub-stage of the running call (free text: "forest", "big cells" ...)
    std::vector<Segment> segments;
    void seg_begin();                         // after a collective has returned / at the start of a call
    void seg_end(int kind);                   // before a collective is entered / at the end of a call

    hipEvent_t get_event();
    void begin_call(tdx_stats* st);
    void end_call();                          // synchronises, fills stats
    void abort_call();   // an entry point leaves early: nothing may point at the caller's stats any more
    int span_begin(int kclass);              // returns the span's index (-1 when timing is off); spans may nest
    void span_end(int index);
};

extern thread_local std::string g_tdx_thread_error;

#define TDX_HIP_CHECK(ctx, expr)                                                                   \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                        \
            g_tdx_thread_error = (ctx)->err;                                                       \
            (ctx)->abort_call();                                                                   \
            return TDX_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

// RAII helper: times everything enqueued in its scope under one kernel class
struct TdxSpan {
    tdx_context* c;
    int index;
    TdxSpan(tdx_context* ctx, int kclass) : c(ctx), index(ctx->span_begin(kclass)) {}
    ~TdxSpan() { c->span_end(index); }
};

// scratch slot ids (one namespace for all stages; stages never run concurrently on a context)
enum {
    TDX_S_A = 0, TDX_S_B, TDX_S_C, TDX_S_D, TDX_S_E, TDX_S_F, TDX_S_G, TDX_S_H, TDX_S_I, TDX_S_J, TDX_S_K,
    TDX_S_Q, TDX_S_L, TDX_S_M, TDX_S_N, TDX_S_O, TDX_S_P, TDX_S_R, TDX_S_IO0, TDX_S_IO1, TDX_S_IO2, TDX_S_IO3, TDX_S_IO4, TDX_S_FACT, TDX_S_MACRO, TDX_S_COUNT
};

static inline int tdx_fail(tdx_context* ctx, int code, const std::string& msg) {
    if (ctx) { ctx->err = msg; ctx->abort_call(); }
    g_tdx_thread_error = msg;
    return code;
}
