// Native transports behind tdx_comm (include/taudem_amd.h): what linearpart<T>::share() / passBorders() / ringTerm() /
// MPI_Allreduce are in the reference (src/linearpart.h:194-219, 301-384), for row strips that live in HBM.
//
//   RCCL (xGMI)   one communicator rank per GPU.  A boundary-row exchange is ONE grouped ncclSend/ncclRecv pair per strip
//                 neighbour, enqueued on the context's stream (stream-ordered: the library does not synchronise around it);
//                 a termination vote is an ncclAllReduce on DEVICE int64 values followed by one device->host copy.  Used
//                 by one-process-per-GPU launches (tdx_rccl_comm_create, ids distributed by the launcher) and by the
//                 one-process-N-threads group (ncclCommInitAll).
//   peer          one process, one thread per rank: a rank copies its neighbours' send buffers into its own receive
//                 buffers (hipMemcpyAsync between peer devices = one xGMI transfer, or a local copy when ranks share a
//                 GPU) between two host barriers.  Host-synchronous contract; needs no communication library, so it also
//                 runs N ranks on ONE GPU (how the multi-strip command-line path is tested on a 1-GPU box).
//
// librccl is opened lazily (dlopen) the first time a transport needs it: single-GPU tools never load its 570 MB.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "context.hpp"

namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string err;
};

RcclApi load_rccl() {
    RcclApi a;
    void* h = nullptr;
    // a copy that is already mapped (PyTorch-ROCm bundles one) wins: two RCCLs in one process would each own a topology
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { a.err = std::string("cannot open librccl: ") + dlerror(); return a; }
#define TDX_SYM(name)                                                                  \
    a.name = reinterpret_cast<decltype(a.name)>(dlsym(h, "nccl" #name));              \
    if (!a.name) { a.err = "librccl lacks nccl" #name; return a; }
    TDX_SYM(GetUniqueId) TDX_SYM(CommInitRank) TDX_SYM(CommInitAll) TDX_SYM(CommDestroy) TDX_SYM(CommAbort) TDX_SYM(Send) TDX_SYM(Recv) TDX_SYM(AllReduce)
    TDX_SYM(GroupStart) TDX_SYM(GroupEnd) TDX_SYM(GetErrorString)
#undef TDX_SYM
    a.ok = true;
    return a;
}
RcclApi& rccl() {
    static RcclApi api = load_rccl();
    return api;
}

#define TDX_NCCL(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) {                                                                               \
            g_tdx_thread_error = std::string(#expr) + ": " + rccl().GetErrorString(_r);                       \
            return 1;                                                                                          \
        }                                                                                                      \
    } while (0)

class HostBarrier {   // sense-reversing barrier of the rank threads of one process
public:
    explicit HostBarrier(int n) : n_(n) {}
    // false: the other ranks did not arrive within TDX_COMM_TIMEOUT seconds (default 600) - one of them has failed or left the loop
    bool wait() {
        static const double limit_s = getenv("TDX_COMM_TIMEOUT") ? atof(getenv("TDX_COMM_TIMEOUT")) : 600.0;
        std::unique_lock<std::mutex> lk(m_);
        if (aborted_) return false;
        const unsigned gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); return true; }
        if (cv_.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return gen_ != gen || aborted_; }) && !aborted_) return true;
        // timed out (or aborted): the generation can no longer complete with everybody on board - a rank that arrives late must not pass
        // with one participant missing, and whoever waits now or later fails at once instead of sitting out its own limit
        aborted_ = true;
        cv_.notify_all();
        return false;
    }
    // a rank has failed: everybody waiting here, now or later, gets `false` at once instead of waiting for a rank that will not come
    void abort() {
        std::lock_guard<std::mutex> lk(m_);
        aborted_ = true;
        cv_.notify_all();
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    unsigned gen_ = 0;
    bool aborted_ = false;
};

constexpr int RED_MAX = 16;

}  // namespace

struct tdx_rccl_comm {
    tdx_context* ctx = nullptr;
    ncclComm_t comm = nullptr;   // guarded by `use`: null after an abort
    std::mutex use;              // held around every host-side RCCL call on `comm` (they only enqueue) and, when it can be had, around its abort
    std::atomic<bool> dead{false};   // set by tdx_group_abort BEFORE it asks for `use`: no new RCCL call starts on this communicator
    bool abandoned = false;      // the communicator was aborted under a rank thread that was stuck inside an RCCL call: nobody may touch `comm` again
    tdx_comm c{};
    char* bufs = nullptr;        // 4 x capacity bytes of device memory: send_up send_down recv_up recv_down
    int64_t* d_red = nullptr;    // RED_MAX device words for host-value votes
    int64_t* h_red = nullptr;    // pinned twin
    int64_t exchanges = 0, allreduces = 0;
};

namespace {

int rccl_exchange(void* user, uint64_t bytes) {
    tdx_rccl_comm* r = static_cast<tdx_rccl_comm*>(user);
    RcclApi& a = rccl();
    const int rank = r->c.rank, size = r->c.size;
    hipStream_t s = r->ctx->stream;
    r->exchanges++;
    std::lock_guard<std::mutex> lk(r->use);   // tdx_group_abort (another rank's thread) must not free the communicator under these calls
    if (!r->comm || r->dead.load()) { g_tdx_thread_error = "rank " + std::to_string(rank) + " of " + std::to_string(size) + ": the rank group was aborted"; return 1; }
    TDX_NCCL(a.GroupStart());
    // inside the group the first failure is remembered and the group is ALWAYS closed: a thread that returned between GroupStart
    // and GroupEnd would queue every later RCCL call (the communicator's destruction included) into a group that never ends
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    auto note = [&](ncclResult_t e, const char* w) { if (first == ncclSuccess && e != ncclSuccess) { first = e; what = w; } };
    if (rank > 0) {
        note(a.Send(r->c.send_up, bytes, ncclChar, rank - 1, r->comm, s), "ncclSend(up)");
        note(a.Recv(r->c.recv_up, bytes, ncclChar, rank - 1, r->comm, s), "ncclRecv(up)");
    }
    if (rank < size - 1) {
        note(a.Send(r->c.send_down, bytes, ncclChar, rank + 1, r->comm, s), "ncclSend(down)");
        note(a.Recv(r->c.recv_down, bytes, ncclChar, rank + 1, r->comm, s), "ncclRecv(down)");
    }
    note(a.GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess) {
        g_tdx_thread_error = "rank " + std::to_string(rank) + " of " + std::to_string(size) + ", exchange " + std::to_string(r->exchanges) + ": " + what + ": " +
                             a.GetErrorString(first);
        return 1;
    }
    return 0;
}
int rccl_allreduce_dev(void* user, int64_t* d_values, int32_t count, int32_t op) {
    tdx_rccl_comm* r = static_cast<tdx_rccl_comm*>(user);
    r->allreduces++;
    std::lock_guard<std::mutex> lk(r->use);
    if (!r->comm || r->dead.load()) { g_tdx_thread_error = "rank " + std::to_string(r->c.rank) + " of " + std::to_string(r->c.size) + ": the rank group was aborted"; return 1; }
    TDX_NCCL(rccl().AllReduce(d_values, d_values, size_t(count), ncclInt64, op == TDX_OP_MAX ? ncclMax : ncclSum, r->comm, r->ctx->stream));
    return 0;
}
int rccl_allreduce(void* user, int64_t* values, int32_t count, int32_t op) {
    tdx_rccl_comm* r = static_cast<tdx_rccl_comm*>(user);
    if (count > RED_MAX) { g_tdx_thread_error = "tdx_comm allreduce: more than 16 values"; return 1; }
    hipStream_t s = r->ctx->stream;
    memcpy(r->h_red, values, size_t(count) * 8);
    if (hipMemcpyAsync(r->d_red, r->h_red, size_t(count) * 8, hipMemcpyHostToDevice, s) != hipSuccess) return 1;
    if (rccl_allreduce_dev(user, r->d_red, count, op) != 0) return 1;
    if (hipMemcpyAsync(r->h_red, r->d_red, size_t(count) * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    memcpy(values, r->h_red, size_t(count) * 8);
    return 0;
}

// buffers + callbacks around an existing communicator rank
int rccl_wrap(tdx_context* ctx, ncclComm_t comm, int rank, int size, int64_t nx, tdx_rccl_comm** out) {
    tdx_rccl_comm* r = new tdx_rccl_comm;
    r->ctx = ctx; r->comm = comm;
    const uint64_t cap = uint64_t(nx) * 16;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&r->bufs), size_t(cap) * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&r->d_red), RED_MAX * 8) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&r->h_red), RED_MAX * 8) != hipSuccess) {
        delete r;
        return tdx_fail(ctx, TDX_ERR_NOMEM, "tdx_rccl_comm: cannot allocate the exchange buffers");
    }
    r->c.rank = rank; r->c.size = size; r->c.user = r;
    r->c.exchange = rccl_exchange; r->c.allreduce = rccl_allreduce; r->c.allreduce_dev = rccl_allreduce_dev;
    r->c.send_up = r->bufs; r->c.send_down = r->bufs + cap; r->c.recv_up = r->bufs + 2 * cap; r->c.recv_down = r->bufs + 3 * cap;
    r->c.capacity = cap;
    r->c.flags = TDX_COMM_STREAM_ORDERED;
    *out = r;
    return TDX_OK;
}

}  // namespace

extern "C" int tdx_rccl_unique_id(void* id128) {
    if (!id128) return tdx_fail(nullptr, TDX_ERR_ARG, "tdx_rccl_unique_id: null");
    if (!rccl().ok) return tdx_fail(nullptr, TDX_ERR_HIP, rccl().err);
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return tdx_fail(nullptr, TDX_ERR_HIP, "ncclGetUniqueId failed");
    static_assert(sizeof(id) == TDX_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return TDX_OK;
}

extern "C" int tdx_rccl_comm_create(tdx_context* ctx, const void* id128, int32_t rank, int32_t size, int64_t nx, tdx_rccl_comm** out) {
    if (!ctx || !id128 || !out || size < 1 || rank < 0 || rank >= size || nx <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_rccl_comm_create: bad argument");
    if (!rccl().ok) return tdx_fail(ctx, TDX_ERR_HIP, rccl().err);
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl().CommInitRank(&comm, size, id, rank);
    if (r != ncclSuccess) return tdx_fail(ctx, TDX_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    const int rc = rccl_wrap(ctx, comm, rank, size, nx, out);
    if (rc != TDX_OK) rccl().CommDestroy(comm);
    return rc;
}
extern "C" const tdx_comm* tdx_rccl_comm_handle(tdx_rccl_comm* c) { return c ? &c->c : nullptr; }
extern "C" void tdx_rccl_comm_counters(const tdx_rccl_comm* c, int64_t* exchanges, int64_t* allreduces) {
    if (exchanges) *exchanges = c ? c->exchanges : 0;
    if (allreduces) *allreduces = c ? c->allreduces : 0;
}
extern "C" void tdx_rccl_comm_destroy(tdx_rccl_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm && !c->abandoned) rccl().CommDestroy(c->comm);
    (void)hipFree(c->bufs); (void)hipFree(c->d_red); (void)hipHostFree(c->h_red);
    delete c;
}

// One rank talking to itself: the send/recv pair, the device vote and the host vote, on the context's stream.
extern "C" int tdx_rccl_selftest(tdx_context* ctx) {
    if (!ctx) return tdx_fail(nullptr, TDX_ERR_ARG, "tdx_rccl_selftest: null context");
    unsigned char id[TDX_RCCL_ID_BYTES];
    int rc = tdx_rccl_unique_id(id);
    if (rc != TDX_OK) return rc;
    tdx_rccl_comm* r = nullptr;
    rc = tdx_rccl_comm_create(ctx, id, 0, 1, 1024, &r);
    if (rc != TDX_OK) return rc;
    RcclApi& a = rccl();
    hipStream_t s = ctx->stream;
    const size_t bytes = 4096;
    int fail = 0;
    std::vector<unsigned char> h(bytes), back(bytes, 0);
    for (size_t i = 0; i < bytes; i++) h[i] = (unsigned char)(i * 7 + 3);
    if (hipMemcpyAsync(r->c.send_up, h.data(), bytes, hipMemcpyHostToDevice, s) != hipSuccess) fail = 1;
    if (!fail && a.GroupStart() == ncclSuccess) {   // (the group is closed whatever happens inside it)
        const bool sent = a.Send(r->c.send_up, bytes, ncclChar, 0, r->comm, s) == ncclSuccess && a.Recv(r->c.recv_down, bytes, ncclChar, 0, r->comm, s) == ncclSuccess;
        if (a.GroupEnd() != ncclSuccess || !sent) fail = 2;
    } else if (!fail) fail = 2;
    if (!fail && hipMemcpyAsync(back.data(), r->c.recv_down, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) fail = 3;
    int64_t v[2] = {41, -7};
    if (!fail && r->c.allreduce(r, v, 2, TDX_OP_SUM) != 0) fail = 4;   // synchronises the stream
    if (!fail && (v[0] != 41 || v[1] != -7 || back != h)) fail = 5;
    if (!fail && r->c.exchange(r, 64) != 0) fail = 6;                  // size 1: no neighbour, an empty group
    if (!fail && hipStreamSynchronize(s) != hipSuccess) fail = 7;
    tdx_rccl_comm_destroy(r);
    if (fail) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_rccl_selftest failed at step " + std::to_string(fail) + (g_tdx_thread_error.empty() ? "" : ": " + g_tdx_thread_error));
    return TDX_OK;
}

// Latencies of the transport's two primitives as the strip protocol uses them (strips.hpp), `reps` repetitions each, microseconds per repetition:
//   out_us[0]  one boundary-row exchange (bytes per direction) + what the protocol does around it: the pre-exchange synchronisation of a
//              host-synchronous transport, none for a stream-ordered one - and ONE stream synchronisation at the very end (exchanges are followed by
//              device work, not by a host wait)
//   out_us[1]  one termination vote: all-reduce of one int64 that lives in device memory + the device-to-host read of the result + the wait for it
//              (strip_allreduce_device), i.e. the host-visible round trip every outer round ends with
// Collective: every rank of the communicator calls it with the same arguments.  comm == NULL: out_us = 0.
extern "C" int tdx_comm_latency(tdx_context* ctx, const tdx_comm* comm, int32_t reps, uint64_t bytes, double* out_us) {
    if (!ctx || !out_us || reps < 1) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_comm_latency: bad argument");
    out_us[0] = out_us[1] = 0.0;
    if (!comm) return TDX_OK;
    if (bytes > comm->capacity) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_comm_latency: more bytes than the exchange buffers hold");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const bool ordered = (comm->flags & TDX_COMM_STREAM_ORDERED) != 0;
    unsigned long long* d_v = reinterpret_cast<unsigned long long*>(ctx->d_mail) + TDX_MAIL_STRIP_CHANGED;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int pass = 0; pass < 2; pass++) {   // pass 0: warm-up (connections are set up by the first collective)
        const int n = pass ? reps : std::min(reps, 8);
        TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
        double t0 = now();
        for (int i = 0; i < n; i++) {
            if (!ordered) TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
            if (comm->exchange(comm->user, bytes) != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm_latency: exchange failed: " + g_tdx_thread_error);
        }
        TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
        out_us[0] = (now() - t0) / n;
        t0 = now();
        for (int i = 0; i < n; i++) {
            TDX_HIP_CHECK(ctx, hipMemsetAsync(d_v, 0, 8, s));
            int64_t v = 0;
            if (comm->allreduce_dev) {
                if (comm->allreduce_dev(comm->user, reinterpret_cast<int64_t*>(d_v), 1, TDX_OP_SUM) != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm_latency: allreduce_dev failed: " + g_tdx_thread_error);
                TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail + TDX_MAIL_STRIP_REDUCE, d_v, 8, hipMemcpyDeviceToHost, s));
                TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
            } else {
                TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_mail + TDX_MAIL_STRIP_REDUCE, d_v, 8, hipMemcpyDeviceToHost, s));
                TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));
                if (comm->allreduce(comm->user, &v, 1, TDX_OP_SUM) != 0) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_comm_latency: allreduce failed: " + g_tdx_thread_error);
            }
        }
        out_us[1] = (now() - t0) / n;
    }
    return TDX_OK;
}

// The same two figures for RCCL with ONE rank talking to itself (the only RCCL configuration a one-GPU box can run: a send / recv pair of `bytes`
// to rank 0 in one group, and the device all-reduce + read-back): the software path of a collective - enqueue, proxy, kernel launch, completion -
// without a link in it, i.e. a LOWER bound of the latencies between GPUs.  out_us[0] exchange, out_us[1] vote.
extern "C" int tdx_rccl_latency(tdx_context* ctx, int32_t reps, uint64_t bytes, double* out_us) {
    if (!ctx || !out_us || reps < 1 || bytes < 1) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_rccl_latency: bad argument");
    unsigned char id[TDX_RCCL_ID_BYTES];
    int rc = tdx_rccl_unique_id(id);
    if (rc != TDX_OK) return rc;
    tdx_rccl_comm* r = nullptr;
    rc = tdx_rccl_comm_create(ctx, id, 0, 1, int64_t((bytes + 15) / 16), &r);
    if (rc != TDX_OK) return rc;
    RcclApi& a = rccl();
    hipStream_t s = ctx->stream;
    int fail = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int pass = 0; pass < 2 && !fail; pass++) {
        const int n = pass ? reps : std::min(reps, 8);
        if (hipStreamSynchronize(s) != hipSuccess) fail = 1;
        double t0 = now();
        for (int i = 0; i < n && !fail; i++) {
            if (a.GroupStart() != ncclSuccess) { fail = 2; break; }
            const bool sent = a.Send(r->c.send_up, bytes, ncclChar, 0, r->comm, s) == ncclSuccess && a.Recv(r->c.recv_down, bytes, ncclChar, 0, r->comm, s) == ncclSuccess;
            if (a.GroupEnd() != ncclSuccess || !sent) fail = 2;
        }
        if (!fail && hipStreamSynchronize(s) != hipSuccess) fail = 3;
        out_us[0] = (now() - t0) / n;
        t0 = now();
        for (int i = 0; i < n && !fail; i++) {
            if (hipMemsetAsync(r->d_red, 0, 8, s) != hipSuccess) fail = 4;
            if (!fail && r->c.allreduce_dev(r, r->d_red, 1, TDX_OP_SUM) != 0) fail = 4;
            if (!fail && hipMemcpyAsync(r->h_red, r->d_red, 8, hipMemcpyDeviceToHost, s) != hipSuccess) fail = 4;
            if (!fail && hipStreamSynchronize(s) != hipSuccess) fail = 4;
        }
        out_us[1] = (now() - t0) / n;
    }
    tdx_rccl_comm_destroy(r);
    if (fail) return tdx_fail(ctx, TDX_ERR_HIP, "tdx_rccl_latency failed at step " + std::to_string(fail) + (g_tdx_thread_error.empty() ? "" : ": " + g_tdx_thread_error));
    return TDX_OK;
}

// ---- one process, N rank threads ---------------------------------------------------------------------------------------
struct tdx_group {
    int size = 0;
    std::string transport;
    std::vector<tdx_context*> ctxs;
    std::vector<tdx_rccl_comm*> rc;       // "rccl"
    struct PeerRank { tdx_group* g; int rank; tdx_comm c; char* bufs; };
    std::vector<PeerRank> pr;             // "peer"
    HostBarrier* bar = nullptr;
    std::vector<int64_t> red;             // size x RED_MAX vote slots
    std::atomic<bool> aborted{false};
};

namespace {

int peer_exchange(void* user, uint64_t bytes) {
    tdx_group::PeerRank* me = static_cast<tdx_group::PeerRank*>(user);
    tdx_group* g = me->g;
    tdx_context* ctx = g->ctxs[size_t(me->rank)];
    // every rank has synchronised its stream before calling (host-synchronous contract): after the barrier all send buffers are final
    auto late = [&](const char* where) {
        g_tdx_thread_error = "peer transport: rank " + std::to_string(me->rank) + " of " + std::to_string(g->size) + " waited in vain for the other ranks at " + where;
        return 1;
    };
    if (!g->bar->wait()) return late("the barrier before an exchange");
    int fail = 0;
    if (hipSetDevice(ctx->device) != hipSuccess) fail = 1;
    if (!fail && me->rank > 0 && hipMemcpyAsync(me->c.recv_up, g->pr[size_t(me->rank - 1)].c.send_down, bytes, hipMemcpyDefault, ctx->stream) != hipSuccess) fail = 1;
    if (!fail && me->rank < g->size - 1 && hipMemcpyAsync(me->c.recv_down, g->pr[size_t(me->rank + 1)].c.send_up, bytes, hipMemcpyDefault, ctx->stream) != hipSuccess) fail = 1;
    if (!fail && hipStreamSynchronize(ctx->stream) != hipSuccess) fail = 1;
    if (!g->bar->wait()) return late("the barrier after an exchange");   // nobody refills a send buffer before its neighbour has copied it
    return fail;
}
int peer_allreduce(void* user, int64_t* values, int32_t count, int32_t op) {
    tdx_group::PeerRank* me = static_cast<tdx_group::PeerRank*>(user);
    tdx_group* g = me->g;
    if (count > RED_MAX) return 1;
    memcpy(&g->red[size_t(me->rank) * RED_MAX], values, size_t(count) * 8);
    auto late = [&]() {
        g_tdx_thread_error = "peer transport: rank " + std::to_string(me->rank) + " of " + std::to_string(g->size) + " waited in vain for the other ranks at an all-reduce";
        return 1;
    };
    if (!g->bar->wait()) return late();
    int64_t acc[RED_MAX];
    for (int i = 0; i < count; i++) {
        int64_t a = g->red[size_t(i)];
        for (int r = 1; r < g->size; r++) {
            const int64_t b = g->red[size_t(r) * RED_MAX + size_t(i)];
            a = op == TDX_OP_MAX ? (b > a ? b : a) : a + b;
        }
        acc[i] = a;
    }
    if (!g->bar->wait()) return late();   // all ranks have read the slots
    memcpy(values, acc, size_t(count) * 8);
    return 0;
}

}  // namespace

extern "C" int tdx_group_create(int32_t size, const int32_t* devices, int64_t nx, tdx_group** out) {
    if (size < 1 || !devices || nx <= 0 || !out) return tdx_fail(nullptr, TDX_ERR_ARG, "tdx_group_create: bad argument");
    tdx_group* g = new tdx_group;
    g->size = size;
    std::set<int> distinct(devices, devices + size);
    const char* want = getenv("TAUDEM_AMD_COMM");
    g->transport = want ? want : (int(distinct.size()) == size ? "rccl" : "peer");
    if (g->transport != "rccl" && g->transport != "peer") { delete g; return tdx_fail(nullptr, TDX_ERR_ARG, "TAUDEM_AMD_COMM must be rccl or peer"); }
    if (g->transport == "rccl" && int(distinct.size()) != size) { delete g; return tdx_fail(nullptr, TDX_ERR_ARG, "the RCCL transport needs one distinct GPU per rank"); }
    int rc = TDX_OK;
    for (int r = 0; r < size && rc == TDX_OK; r++) {
        tdx_context* c = nullptr;
        rc = tdx_context_create(devices[r], &c);
        if (rc == TDX_OK) g->ctxs.push_back(c);
    }
    if (rc == TDX_OK && g->transport == "rccl") {
        if (!rccl().ok) rc = tdx_fail(nullptr, TDX_ERR_HIP, rccl().err);
        std::vector<ncclComm_t> comms(size_t(size), nullptr);
        if (rc == TDX_OK) {
            const ncclResult_t nr = rccl().CommInitAll(comms.data(), size, devices);
            if (nr != ncclSuccess) rc = tdx_fail(nullptr, TDX_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(nr));
        }
        for (int r = 0; r < size && rc == TDX_OK; r++) {
            tdx_rccl_comm* w = nullptr;
            rc = rccl_wrap(g->ctxs[size_t(r)], comms[size_t(r)], r, size, nx, &w);
            if (rc == TDX_OK) { g->rc.push_back(w); comms[size_t(r)] = nullptr; }
        }
        for (ncclComm_t c : comms) if (c) rccl().CommDestroy(c);
    } else if (rc == TDX_OK) {
        g->bar = new HostBarrier(size);
        g->red.assign(size_t(size) * RED_MAX, 0);
        g->pr.resize(size_t(size));
        const uint64_t cap = uint64_t(nx) * 16;
        for (int r = 0; r < size && rc == TDX_OK; r++) {
            tdx_group::PeerRank& p = g->pr[size_t(r)];
            p.g = g; p.rank = r; p.bufs = nullptr;
            if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&p.bufs), size_t(cap) * 4) != hipSuccess) {
                rc = tdx_fail(nullptr, TDX_ERR_NOMEM, "tdx_group_create: cannot allocate the exchange buffers");
                break;
            }
            for (int nb : {r - 1, r + 1})   // strip neighbours on other GPUs: direct xGMI access
                if (nb >= 0 && nb < size && devices[nb] != devices[r]) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, devices[r], devices[nb]) == hipSuccess && can) {
                        const hipError_t e = hipDeviceEnablePeerAccess(devices[nb], 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                    }
                }
            memset(&p.c, 0, sizeof(p.c));
            p.c.rank = r; p.c.size = size; p.c.user = &p;
            p.c.exchange = peer_exchange; p.c.allreduce = peer_allreduce;
            p.c.send_up = p.bufs; p.c.send_down = p.bufs + cap; p.c.recv_up = p.bufs + 2 * cap; p.c.recv_down = p.bufs + 3 * cap;
            p.c.capacity = cap;
        }
    }
    if (rc != TDX_OK) { const std::string keep = g_tdx_thread_error; tdx_group_destroy(g); g_tdx_thread_error = keep; return rc; }
    *out = g;
    return TDX_OK;
}
extern "C" tdx_context* tdx_group_context(tdx_group* g, int32_t rank) { return (g && rank >= 0 && rank < g->size) ? g->ctxs[size_t(rank)] : nullptr; }
extern "C" const tdx_comm* tdx_group_comm(tdx_group* g, int32_t rank) {
    if (!g || rank < 0 || rank >= g->size) return nullptr;
    return g->transport == "rccl" ? &g->rc[size_t(rank)]->c : &g->pr[size_t(rank)].c;
}
extern "C" const char* tdx_group_transport(const tdx_group* g) { return g ? g->transport.c_str() : ""; }
// A rank of the group has failed outside a collective (allocation, input ...): the other ranks must not wait for it.  Peer transport:
// every barrier wait, pending or future, fails at once; RCCL: the communicators are aborted, so that enqueued and future operations
// end in an error instead of waiting for a partner.  The group can only be destroyed afterwards.
extern "C" void tdx_group_abort(tdx_group* g) {
    if (!g) return;
    if (g->aborted.exchange(true)) return;   // one shot: every failing rank thread calls this, the first one does the work
    if (g->bar) g->bar->abort();
    // Every communicator is marked dead FIRST (no rank thread starts a new RCCL call), then ONE bounded wait of 2 s covers all of them: N stuck ranks
    // delay the abort by 2 s, not 2 N s.
    for (tdx_rccl_comm* r : g->rc) if (r) r->dead.store(true);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
    for (tdx_rccl_comm* r : g->rc) {
        if (!r) continue;
        // Take the communicator away under its lock: a rank thread is either before its calls (and will find `dead`) or past them.  A rank thread
        // that is STUCK inside an RCCL call (ncclGroupEnd setting up a connection to a peer that has already failed) holds the lock for good -
        // and the abort is the only thing that releases it: after the bounded wait the communicator is aborted under that thread's feet (what
        // ncclCommAbort is for); its call returns an error, every later call finds `dead`, and nobody touches the handle again.
        // ORDERING the unlocked accesses below rely on: this function runs at most once per group (the `aborted` guard above), and
        // tdx_group_destroy - the only other reader of r->comm / writer of r->abandoned outside a rank thread's locked section - is called
        // after the rank threads have been joined (taudem_amd.distributed.StripGroup.__exit__, tool_strips.hpp).
        std::unique_lock<std::mutex> lk(r->use, std::defer_lock);
        bool got = false;
        while (!(got = lk.try_lock()) && std::chrono::steady_clock::now() < deadline) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        ncclComm_t c = r->comm;
        if (got) { r->comm = nullptr; lk.unlock(); }
        else r->abandoned = true;
        if (c) rccl().CommAbort(c);          // ends what is enqueued on the rank's stream
    }
}
extern "C" void tdx_group_destroy(tdx_group* g) {
    if (!g) return;
    for (tdx_rccl_comm* r : g->rc) tdx_rccl_comm_destroy(r);
    for (auto& p : g->pr) if (p.bufs) { (void)hipSetDevice(g->ctxs[size_t(p.rank)]->device); (void)hipFree(p.bufs); }
    for (tdx_context* c : g->ctxs) tdx_context_destroy(c);
    delete g->bar;
    delete g;
}
