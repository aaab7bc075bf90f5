#include "context.hpp"

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>

thread_local std::string g_tdx_thread_error;

void* tdx_context::scratch(int slot, size_t bytes) {
    if (slots.size() < size_t(TDX_S_COUNT)) slots.resize(size_t(TDX_S_COUNT));
    Slot& s = slots[size_t(slot)];
    if (bytes == 0) bytes = 16;
    if (s.bytes >= bytes) return s.p;
    if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.bytes = 0; }
    size_t want = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&s.p, want);
    if (e != hipSuccess) {
        err = std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
        g_tdx_thread_error = err;
        s.p = nullptr;
        return nullptr;
    }
    s.bytes = want;
    return s.p;
}

// the device token of segment_trace mode 2 (a plain mutex would have to be unlocked by the thread that locked it; a rank thread always is, but keep it explicit)
namespace {
std::mutex g_tok_m;
std::condition_variable g_tok_cv;
bool g_tok_taken = false;
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

void tdx_context::seg_begin() {
    if (!seg_mode || seg_open) return;
    // Mode 2 hands ONE device token round: that is only safe when a collective BLOCKS until every rank has entered it (the peer transport's barriers).  A
    // stream-ordered transport (RCCL) merely enqueues: a rank could re-take the token behind a collective its peer - still waiting for the token - can
    // never post, and the next seg_end would wait on that stream for ever.  There mode 2 times like mode 1 (the ranks have a GPU each anyway).
    if (seg_mode == 2 && comm_size > 1 && !comm_ordered) {
        std::unique_lock<std::mutex> lk(g_tok_m);
        g_tok_cv.wait(lk, [] { return !g_tok_taken; });
        g_tok_taken = true;
        seg_token = true;
    }
    if (!seg_ev0) { (void)hipEventCreate(&seg_ev0); (void)hipEventCreate(&seg_ev1); }
    (void)hipEventRecord(seg_ev0, stream);
    seg_t0 = now_ms();
    seg_open = true;
}

void tdx_context::seg_end(int kind) {
    if (!seg_mode || !seg_open) return;
    (void)hipEventRecord(seg_ev1, stream);
    (void)hipEventSynchronize(seg_ev1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, seg_ev0, seg_ev1);
    segments.push_back(Segment{stage, phase, kind, ms, float(now_ms() - seg_t0)});
    seg_open = false;
    if (seg_token) {
        { std::lock_guard<std::mutex> lk(g_tok_m); g_tok_taken = false; }
        g_tok_cv.notify_one();
        seg_token = false;
    }
}

void tdx_context::abort_call() {
    timing = false; cur_stats = nullptr;
    if (seg_token) {   // a failing rank must not keep the others off the device
        { std::lock_guard<std::mutex> lk(g_tok_m); g_tok_taken = false; }
        g_tok_cv.notify_one();
        seg_token = false;
    }
    seg_open = false;
}

hipEvent_t tdx_context::get_event() {
    if (events_used == event_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        event_pool.push_back(e);
    }
    return event_pool[events_used++];
}

void tdx_context::begin_call(tdx_stats* st) {
    comm_exchanges = comm_allreduces = 0;
    cur_stats = st;
    timing = (st != nullptr);
    spans.clear();
    events_used = 0;
    phase = "";
    seg_begin();
    if (st) {
        memset(st, 0, sizeof(*st));
        ev_begin = get_event();
        (void)hipEventRecord(ev_begin, stream);
    }
}

int tdx_context::span_begin(int kclass) {
    if (!timing) return -1;
    Span s;
    s.a = get_event(); s.b = get_event(); s.kclass = kclass;
    (void)hipEventRecord(s.a, stream);
    spans.push_back(s);
    return int(spans.size()) - 1;
}

void tdx_context::span_end(int index) {
    if (!timing || index < 0 || size_t(index) >= spans.size()) return;
    (void)hipEventRecord(spans[size_t(index)].b, stream);
}

void tdx_context::end_call() {
    seg_end(2);
    if (timing) {
        ev_end = get_event();
        (void)hipEventRecord(ev_end, stream);
    }
    (void)hipStreamSynchronize(stream);
    if (timing && cur_stats) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev_begin, ev_end);
        cur_stats->ms_total = ms;
        for (const Span& s : spans) {
            float t = 0;
            if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess) cur_stats->ms_kernel[s.kclass] += t;
        }
    }
    if (comm_size > 1) {
        static const bool trace = getenv("TDX_COMM_TRACE") != nullptr && atoi(getenv("TDX_COMM_TRACE")) != 0;
        if (trace)
            fprintf(stderr, "taudem_amd[rank %d/%d] %s: %lld exchanges, %lld all-reduces%s\n", comm_rank, comm_size, stage, (long long)comm_exchanges,
                    (long long)comm_allreduces, timing && cur_stats ? (", " + std::to_string(cur_stats->ms_total) + " ms").c_str() : "");
    }
    timing = false;
    cur_stats = nullptr;
}

extern "C" {

void tdx_context_destroy(tdx_context* c);

const char* tdx_version(void) { return "taudem_amd 0.1.0 (TauDEM 5.4.0 hot path, gfx950)"; }

int tdx_context_set_option(tdx_context* c, const char* name, int64_t value) {
    if (!c || !name) return TDX_ERR_ARG;
    if (strcmp(name, "kernel_timing") == 0) { c->kernel_timing = value != 0; return TDX_OK; }
    if (strcmp(name, "segment_trace") == 0) {
        if (value < 0 || value > 2) return tdx_fail(c, TDX_ERR_ARG, "segment_trace: 0 (off), 1 (timed) or 2 (timed, one rank on the device at a time)");
        c->seg_mode = int(value);
        c->segments.clear();
        return TDX_OK;
    }
    return tdx_fail(c, TDX_ERR_ARG, std::string("unknown option ") + name);
}

void tdx_context_comm_counters(const tdx_context* c, int64_t* exchanges, int64_t* allreduces) {
    if (exchanges) *exchanges = c ? c->comm_exchanges_total : 0;
    if (allreduces) *allreduces = c ? c->comm_allreduces_total : 0;
}

int64_t tdx_context_segments(tdx_context* c, tdx_segment* out, int64_t capacity) {
    if (!c) return 0;
    const int64_t n = int64_t(c->segments.size());
    for (int64_t i = 0; out && i < n && i < capacity; i++) {
        const tdx_context::Segment& g = c->segments[size_t(i)];
        tdx_segment& o = out[i];
        memset(&o, 0, sizeof(o));
        strncpy(o.stage, g.stage ? g.stage : "", sizeof(o.stage) - 1);
        strncpy(o.phase, g.phase ? g.phase : "", sizeof(o.phase) - 1);
        o.kind = g.kind; o.device_ms = g.device_ms; o.wall_ms = g.wall_ms;
    }
    if (out) c->segments.clear();
    return n;
}

int tdx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int context_init(tdx_context* c, int device) {
    c->device = device;
    TDX_HIP_CHECK(c, hipSetDevice(device));
    TDX_HIP_CHECK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    TDX_HIP_CHECK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_mail), TDX_MAIL_WORDS * sizeof(uint64_t), hipHostMallocDefault));
    TDX_HIP_CHECK(c, hipMalloc(reinterpret_cast<void**>(&c->d_mail), TDX_MAIL_WORDS * sizeof(uint64_t)));
    TDX_HIP_CHECK(c, hipMemset(c->d_mail, 0, TDX_MAIL_WORDS * sizeof(uint64_t)));
    c->slots.resize(size_t(TDX_S_COUNT));
    return TDX_OK;
}

int tdx_context_create(int device, tdx_context** out) {
    if (!out) return TDX_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_tdx_thread_error = "no HIP device available: taudem_amd has no CPU fallback";
        return TDX_ERR_NOGPU;
    }
    if (device < 0 || device >= n) { g_tdx_thread_error = "bad device index"; return TDX_ERR_ARG; }
    tdx_context* c = new tdx_context;
    const int rc = context_init(c, device);
    if (rc != TDX_OK) {   // nothing of a half-built context survives (the error text stays in the calling thread's slot)
        const std::string keep = g_tdx_thread_error;
        tdx_context_destroy(c);
        g_tdx_thread_error = keep;
        return rc;
    }
    *out = c;
    return TDX_OK;
}

void tdx_context_destroy(tdx_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->slots) if (s.p) (void)hipFree(s.p);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    if (c->d_mail) (void)hipFree(c->d_mail);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->seg_ev0) { (void)hipEventDestroy(c->seg_ev0); (void)hipEventDestroy(c->seg_ev1); }
    for (hipEvent_t e : c->ev_batch) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Gives the scratch arena back to the device (the slots grow again with the next call that needs them): for a host that keeps a context alive across
// workloads of very different sizes - a 32768^2 raster, then eight contexts of other ranks on the same GPU.
int tdx_context_release_scratch(tdx_context* c) {
    if (!c) return TDX_ERR_ARG;
    TDX_HIP_CHECK(c, hipSetDevice(c->device));
    if (c->stream) TDX_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) TDX_HIP_CHECK(c, hipStreamSynchronize(c->stream2));
    for (auto& s : c->slots) if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.bytes = 0; }
    c->fact_dxc.clear(); c->fact_dyc.clear();   // (the distance table lived in a slot)
    return TDX_OK;
}

const char* tdx_last_error(const tdx_context* c) { return c ? c->err.c_str() : g_tdx_thread_error.c_str(); }

int tdx_synchronize(tdx_context* c) {
    if (!c) return TDX_ERR_ARG;
    TDX_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TDX_OK;
}

void* tdx_stream(tdx_context* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

int tdx_device_alloc(tdx_context* c, uint64_t bytes, void** dptr) {
    if (!c || !dptr) return TDX_ERR_ARG;
    TDX_HIP_CHECK(c, hipSetDevice(c->device));
    TDX_HIP_CHECK(c, hipMalloc(dptr, bytes ? bytes : 16));
    return TDX_OK;
}
int tdx_device_free(tdx_context* c, void* dptr) {
    if (!c) return TDX_ERR_ARG;
    TDX_HIP_CHECK(c, hipFree(dptr));
    return TDX_OK;
}
int tdx_copy_to_device(tdx_context* c, void* dptr, const void* host, uint64_t bytes) {
    if (!c) return TDX_ERR_ARG;
    TDX_HIP_CHECK(c, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, c->stream));
    TDX_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TDX_OK;
}
int tdx_copy_to_host(tdx_context* c, void* host, const void* dptr, uint64_t bytes) {
    if (!c) return TDX_ERR_ARG;
    TDX_HIP_CHECK(c, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
    TDX_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TDX_OK;
}

}  // extern "C"
