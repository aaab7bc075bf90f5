// Row strips behind the file-level tool functions: what `mpiexec -n P tool ...` is in the reference (every rank reads its
// strip, computes with halo exchange, writes its strip: src/linearpart.h:133-134, src/tiffIO.cpp:186-290) becomes ONE process
// that partitions the host raster over N GPUs, one thread per GPU (tdx_group: RCCL over xGMI between distinct GPUs, peer
// copies when ranks share a device), and gathers the owned rows back into the host raster that is written to the file.
#pragma once
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "context.hpp"

namespace toolstrips {

struct RankJob {
    tdx_context* ctx = nullptr;
    const tdx_comm* comm = nullptr;
    int rank = 0, size = 1;
    int64_t nx = 0, ny = 0, y0 = 0, y1 = 0, nyl = 0;   // global raster, owned global rows [y0, y1)
    std::vector<void*> owned;                           // device allocations of this job
    ~RankJob() { for (void* p : owned) (void)hipFree(p); }

    size_t strip_cells() const { return size_t(nx) * size_t(nyl + 2); }
    // a (nyl + 2) x nx strip array in HBM; host_full != nullptr: its owned rows are uploaded into rows 1..nyl (halo rows are the library's)
    template <class T>
    T* strip(const T* host_full) {
        void* p = nullptr;
        if (hipMalloc(&p, strip_cells() * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        owned.push_back(p);
        if (host_full && hipMemcpyAsync(static_cast<T*>(p) + size_t(nx), host_full + size_t(y0) * size_t(nx), size_t(nx) * size_t(nyl) * sizeof(T),
                                        hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
        return static_cast<T*>(p);
    }
    // owned rows of a strip array back into the full host raster
    template <class T>
    bool fetch(T* host_full, const T* dev) {
        return hipMemcpyAsync(host_full + size_t(y0) * size_t(nx), dev + size_t(nx), size_t(nx) * size_t(nyl) * sizeof(T), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
               hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    // per-row values (cell sizes) of the nyl + 2 strip rows: global rows y0 - 1 .. y1, clamped into the raster
    std::vector<double> rows_of(const std::vector<double>& per_row) const {
        std::vector<double> v(size_t(nyl + 2));
        for (int64_t j = 0; j < nyl + 2; j++) v[size_t(j)] = per_row[size_t(std::min<int64_t>(std::max<int64_t>(y0 - 1 + j, 0), ny - 1))];
        return v;
    }
    // global outlet rows -> strip-array rows (row 1 = first owned row); the library ignores outlets outside the owned rows
    std::vector<int32_t> local_rows(const std::vector<int32_t>& oy) const {
        std::vector<int32_t> v(oy.size());
        for (size_t i = 0; i < oy.size(); i++) {
            const int64_t r = int64_t(oy[i]) - y0 + 1;
            v[i] = (r < 1 || r > nyl) ? int32_t(-1) : int32_t(r);
        }
        return v;
    }
};

// Runs body(job) on every rank's thread.  Returns the first non-zero code (a failing rank ends the process with that code
// when others may be waiting for it in a collective - what MPI_Abort does in the reference).  stats0: rank 0's statistics
// with the integer counters the ranks agree on; device time = the slowest rank's.
template <class F>
int run(int ngpus, int base_device, int64_t nx, int64_t ny, tdx_stats* stats0, F body) {
    int ndev = tdx_device_count();
    if (ndev < 1) return tdx_fail(nullptr, TDX_ERR_NOGPU, "no HIP device");
    int size = int(std::min<int64_t>(ngpus, ny));   // at least one row per rank
    std::vector<int32_t> devs(size_t(size), 0);
    for (int r = 0; r < size; r++) devs[size_t(r)] = int32_t((base_device + r) % ndev);
    tdx_group* g = nullptr;
    int rc = tdx_group_create(size, devs.data(), nx, &g);
    if (rc != TDX_OK) return rc;
    if (getenv("TAUDEM_AMD_STATS")) fprintf(stderr, "taudem_amd: %d strips over %d device(s), transport %s\n", size, std::min(size, ndev), tdx_group_transport(g));
    std::vector<int> rcs(static_cast<size_t>(size), 0);
    std::atomic<bool> failed{false};
    std::vector<tdx_stats> sts;
    sts.resize(static_cast<size_t>(size));
    std::vector<std::thread> th;
    const int64_t base = ny / size;
    for (int r = 0; r < size; r++) {
        th.emplace_back([&, r] {
            RankJob job;
            job.ctx = tdx_group_context(g, r); job.comm = tdx_group_comm(g, r);
            job.rank = r; job.size = size; job.nx = nx; job.ny = ny;
            job.y0 = int64_t(r) * base; job.y1 = (r == size - 1) ? ny : int64_t(r + 1) * base;   // remainder to the last rank (src/linearpart.h:133-134)
            job.nyl = job.y1 - job.y0;
            memset(&sts[size_t(r)], 0, sizeof(tdx_stats));
            int e = hipSetDevice(job.ctx->device) == hipSuccess ? body(job, &sts[size_t(r)]) : TDX_ERR_HIP;
            rcs[size_t(r)] = e;
            if (e != TDX_OK && size > 1) {
                // The other ranks may be waiting for this one in a collective: the group is aborted, so that their waits fail at once and
                // every rank RETURNS its error (the caller may be a Python process or the reference's main on the shim - it must not be
                // killed under its feet, which is what _Exit() here used to do).  The first rank to fail reports.
                if (!failed.exchange(true)) fprintf(stderr, "taudem_amd: rank %d of %d failed (%d): %s\n", r, size, e, tdx_last_error(job.ctx));
                tdx_group_abort(g);
            }
        });
    }
    for (auto& t : th) t.join();
    // the error that started it, not the "collective failed" of a rank that was merely waiting for the failed one
    for (int r = 0; r < size; r++) if (rcs[size_t(r)] != TDX_OK && (rc == TDX_OK || rc == TDX_ERR_HIP)) rc = rcs[size_t(r)];
    if (stats0) {
        *stats0 = sts[0];
        for (int r = 1; r < size; r++) {
            stats0->ms_total = std::max(stats0->ms_total, sts[size_t(r)].ms_total);
            for (int k = 0; k < 8; k++) stats0->ms_kernel[k] = std::max(stats0->ms_kernel[k], sts[size_t(r)].ms_kernel[k]);
        }
    }
    tdx_group_destroy(g);
    return rc;
}

}  // namespace toolstrips
