// DinfUpDependence (depgrd, src/DinfUpDependence.cpp:52-272) and DinfRevAccum (dsaccum, src/DinfRevAccum.cpp:51-290) on gfx950 -
// the D-infinity flow-algebra tools that sweep the dependency graph of AreaDinf in REVERSE.
//
// Both evaluate a cell from the cells it sends flow TO (its at most two downslope receivers), so the sweep starts at the cells
// without receivers and runs upstream; the reference does it with one queue per MPI rank and a share() per outer round.  A
// cell's value depends only on its receivers' values (folded in k = 1..8 order, the reference's order of float32 / float64
// operations), so the tile dependency sweep of d8_sweep.hpp applies with the roles swapped:
//   dependency mask  = the cell's own receivers: prop(angle, k) > 0, inside the raster, with an angle (src/DinfRevAccum.cpp:141-150)
//   release mask     = the neighbours that send flow to the cell (src/DinfRevAccum.cpp:201-219)
// DinfRevAccum's two results travel as one 8-byte record (one store: see d8_sweep.hpp).  The two proportions of a cell are
// recomputed from its angle when it is evaluated (prop() of dinf_prop.hpp: fp64 divisions, per-row atan2 from the host libm).
#include <cmath>
#include <cstring>
#include <vector>

#include "context.hpp"
#include "d8_sweep.hpp"
#include "device_common.hpp"
#include "dinf_prop.hpp"

namespace {
using namespace tdxk;

constexpr unsigned RINFO_P1 = 1u << 12, RINFO_P2 = 1u << 15;

// Per cell: [0:8) receivers that count (dependency), [9:12) s1 - 1, [12] / [15] prop > 0 towards s1 / s1 % 8 + 1, [13] the cell has an
// angle, [16:24) neighbours that send flow to the cell (they wait for it)
__global__ __launch_bounds__(256) void rev_setup_kernel(const uint8_t* __restrict__ code, int nx, int ny, uint32_t* __restrict__ info) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    unsigned c[9];
    dinf_code_window(code, nx, ny, x, y, c);   // (codes: pass 1, dinf_prop.hpp - two fp64 divisions per cell instead of ten)
    unsigned inf = 0;
    if (c[0] != DINF_CODE_NODATA) {
        inf |= d8sweep::INFO_PART | ((c[0] & 7u) << 9);
        const int s1 = int(c[0] & 7u) + 1, s2 = s1 % 8 + 1;
        // receivers that count: prop > 0, inside the raster, with an angle (src/DinfRevAccum.cpp:141-150)
        unsigned has_angle = 0;   // bit k - 1: neighbour k lies in the raster and has an angle
#pragma unroll
        for (int k = 1; k <= 8; k++) has_angle |= (c[k] != DINF_CODE_NODATA) ? 1u << (k - 1) : 0u;
        if (c[0] & DINF_CODE_P1) inf |= RINFO_P1 | (has_angle & (1u << (s1 - 1)));
        if (c[0] & DINF_CODE_P2) inf |= RINFO_P2 | (has_angle & (1u << (s2 - 1)));
        // who waits for this cell: neighbours with an angle whose flow reaches it (only meaningful when the cell itself has an angle:
        // a receiver without one is not counted by its senders)
#pragma unroll
        for (int k = 1; k <= 8; k++) {
            const int kk = (k + 4) % 8;
            if (dinf_code_sends(c[k], kk == 0 ? 8 : kk)) inf |= 1u << (16 + k - 1);
        }
    }
    info[size_t(y) * size_t(nx) + size_t(x)] = inf;
}

// the two receiver directions in ascending k (the reference's loop order), with their proportion slots
struct Recv { int k[2]; bool on[2]; };
__device__ __forceinline__ Recv receivers(unsigned inf) {
    const int s1 = int((inf >> 9) & 7u) + 1, s2 = s1 % 8 + 1;
    Recv r;
    const bool p1 = (inf & RINFO_P1) != 0u, p2 = (inf & RINFO_P2) != 0u;
    if (s2 > s1) { r.k[0] = s1; r.on[0] = p1; r.k[1] = s2; r.on[1] = p2; }
    else { r.k[0] = s2; r.on[0] = p2; r.k[1] = s1; r.on[1] = p1; }   // s1 == 8: k = 1 is visited before k = 8
    return r;
}

// what a pending cell needs in the lockstep form of the reverse sweep (d8sweep::sweep_tile_rev), made once per activation: its receivers
// in ascending k, which of them count (prop > 0, inside the raster, with an angle) and their proportions
__device__ __forceinline__ void rev_row_dinf(unsigned inf, float angle, double a2, int (&k)[2], bool (&on)[2], double (&p)[2]) {
    const Recv r = receivers(inf);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        k[t] = r.k[t];
        on[t] = r.on[t] && ((inf >> (r.k[t] - 1)) & 1u) != 0u;
        p[t] = on[t] ? prop_dev(angle, r.k[t], a2) : 0.;
    }
}

struct UpDepAlg {   // src/DinfUpDependence.cpp:184-208
    using Cell = float;
    using Aux = float2;                          // {angle, disturbance grid value (int bits)}
    static constexpr bool HAS_AUX = true, HAS_DIST = false, HAS_ROWS = true;
    static constexpr int kBulkSweeps = 0;            // (not used: the reverse sweeps run d8sweep::sweep_tile_rev, lockstep sweeps to the end of every activation)
    static constexpr int kMinWaves32 = 5;
    static constexpr int kMaxRelease = 8;
    static constexpr unsigned kBulkUntil = 64;       // a wide front of thousands of tiles to the very end: several small tiles per CU beat one large one (489 -> 380 ms at 16384^2)
    static __device__ __forceinline__ float head(float c) { return c; }
    static __host__ __device__ __forceinline__ float outside() { return -1.0f; }
    static __device__ __forceinline__ unsigned rel_mask(unsigned inf) { return (inf >> 16) & 0xFFu; }
    static __device__ __forceinline__ void rev_row(unsigned inf, const Aux& a, double a2, int (&k)[2], bool (&on)[2], double (&p)[2]) { rev_row_dinf(inf, a.x, a2, k, on, p); }
    __device__ __forceinline__ Cell eval2(const Aux& a, const bool (&on)[2], const double (&p)[2], const Cell (&n)[2]) const {
        if (__float_as_int(a.y) >= 1) return 1.0f;
        float dep = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; t++)
            if (on[t]) dep = dep + (float)(n[t] * p[t]);
        return dep;
    }
    template <class L>
    __device__ __forceinline__ void eval(L& S, int c, int cl, int ly, unsigned inf, const Cell (&nb)[9]) const {
        const float2 a = S.aux[c];
        float dep;
        if (__float_as_int(a.y) >= 1) dep = 1.0f;
        else {
            dep = 0.0f;
            const Recv r = receivers(inf);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (!r.on[t] || !((inf >> (r.k[t] - 1)) & 1u)) continue;   // prop > 0, inside the raster, with an angle
                const double p = prop_dev(a.x, r.k[t], S.rows[ly + 1]);
                float depp = 0.f;
#pragma unroll
                for (int k = 1; k <= 8; k++) if (k == r.k[t]) depp = nb[k];
                dep = dep + (float)(depp * p);
            }
        }
        S.v[cl] = dep;
    }
};

struct RevAccAlg {   // src/DinfRevAccum.cpp:176-199; record = {racc, dmax}
    using Cell = float2;
    using Aux = float2;                          // {angle, weight}
    static constexpr bool HAS_AUX = true, HAS_DIST = false, HAS_ROWS = true;
    static constexpr int kBulkSweeps = 0;            // (not used: the reverse sweeps run d8sweep::sweep_tile_rev, lockstep sweeps to the end of every activation)
    static constexpr unsigned kBulkUntil = 64;
    static constexpr int kMinWaves32 = 4;
    static constexpr int kMaxRelease = 8;
    float w_nodata;
    static __device__ __forceinline__ float head(const float2& c) { return c.x; }
    static __host__ __device__ __forceinline__ float2 outside() { return make_float2(TDX_ANG_NODATA, TDX_ANG_NODATA); }
    static __device__ __forceinline__ unsigned rel_mask(unsigned inf) { return (inf >> 16) & 0xFFu; }
    static __device__ __forceinline__ void rev_row(unsigned inf, const Aux& a, double a2, int (&k)[2], bool (&on)[2], double (&p)[2]) { rev_row_dinf(inf, a.x, a2, k, on, p); }
    __device__ __forceinline__ Cell eval2(const Aux& a, const bool (&on)[2], const double (&p)[2], const Cell (&n)[2]) const {
        if (is_nodata_f(a.y, w_nodata)) return make_float2(TDX_ANG_NODATA, TDX_ANG_NODATA);
        float racc = a.y, dmax = a.y;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (!on[t] || is_nodata_f(n[t].x, TDX_ANG_NODATA)) continue;   // (a receiver whose weight was nodata)
            const float valn = (float)(p[t] * n[t].x);
            racc = racc + valn;
            if (n[t].y > dmax) dmax = n[t].y;
        }
        return make_float2(racc, dmax);
    }
    template <class L>
    __device__ __forceinline__ void eval(L& S, int c, int cl, int ly, unsigned inf, const Cell (&nb)[9]) const {
        const float2 a = S.aux[c];
        float racc, dmax;
        if (is_nodata_f(a.y, w_nodata)) { racc = TDX_ANG_NODATA; dmax = TDX_ANG_NODATA; }
        else {
            racc = a.y; dmax = a.y;
            const Recv r = receivers(inf);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (!r.on[t] || !((inf >> (r.k[t] - 1)) & 1u)) continue;
                float2 n = make_float2(0.f, 0.f);
#pragma unroll
                for (int k = 1; k <= 8; k++) if (k == r.k[t]) n = nb[k];
                if (is_nodata_f(n.x, TDX_ANG_NODATA)) continue;        // a receiver whose weight was nodata
                const double p = prop_dev(a.x, r.k[t], S.rows[ly + 1]);
                const float valn = (float)(p * n.x);
                racc = racc + valn;
                if (n.y > dmax) dmax = n.y;
            }
        }
        S.v[cl] = make_float2(racc, dmax);
    }
};

__global__ __launch_bounds__(256) void pack_aux_i_kernel(const float* __restrict__ ANG, const int32_t* __restrict__ DG, size_t n, float2* __restrict__ aux) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) aux[i] = make_float2(ANG[i], __int_as_float(DG[i]));
}
__global__ __launch_bounds__(256) void pack_aux_f_kernel(const float* __restrict__ ANG, const float* __restrict__ W, size_t n, float2* __restrict__ aux) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) aux[i] = make_float2(ANG[i], W[i]);
}
__global__ __launch_bounds__(256) void rev_init2_kernel(const uint32_t* __restrict__ info, float2* __restrict__ rec, size_t first, size_t n) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < first + n) rec[i] = (info[i] & d8sweep::INFO_PART) ? make_float2(__uint_as_float(d8sweep::PENDING_BITS), 0.f) : make_float2(TDX_ANG_NODATA, TDX_ANG_NODATA);
}
__global__ __launch_bounds__(256) void rev_unpack2_kernel(const float2* __restrict__ rec, size_t first, size_t n, float* __restrict__ racc, float* __restrict__ dmax) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= first + n) return;
    const float2 r = rec[i];
    const bool pend = d8sweep::pending(r.x);   // on or above a cycle: never queued by the reference either
    racc[i] = pend ? TDX_ANG_NODATA : r.x;
    dmax[i] = pend ? TDX_ANG_NODATA : r.y;
}

struct RevSetup {
    double* d_a2 = nullptr;
    uint32_t* info = nullptr;
    float2* aux = nullptr;
    uint32_t* flags = nullptr;
    unsigned long long* counts = nullptr;
};
// common front part: halo rows of the angle grid, per-row atan2 table, info words
int rev_prepare(tdx_context* ctx, const Strip& st, float* d_ang, float ang_nodata, const double* dxc, const double* dyc, RevSetup& R, tdx_stats* stats) {
    hipStream_t s = ctx->stream;
    const int inx = st.nx, iny = st.ny_arr;
    const size_t n = size_t(inx) * size_t(iny);
    std::vector<double> a2(size_t(iny), 0.);
    for (int j = 0; j < iny; j++) a2[size_t(j)] = atan2(dyc[j], dxc[j]);
    const tilek::TileGeom geom = tilek::make_geom(inx, iny, st.y0, st.y1);
    const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y);
    R.d_a2 = static_cast<double*>(ctx->scratch(TDX_S_J, a2.size() * 8));
    R.info = static_cast<uint32_t*>(ctx->scratch(TDX_S_A, n * 4));
    R.aux = static_cast<float2*>(ctx->scratch(TDX_S_B, n * 8));
    R.flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));
    R.counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
    if (!R.d_a2 || !R.info || !R.aux || !R.flags || !R.counts) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(R.d_a2, a2.data(), a2.size() * 8, hipMemcpyHostToDevice, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));   // `a2` is a local
    ctx->begin_call(stats);
    strip_mark(ctx, st, "dinfupdependence / dinfrevaccum");
    int rc = strip_exchange<float>(ctx, st, d_ang, ang_nodata);   // flowData->share()
    if (rc != TDX_OK) return rc;
    TdxSpan sp(ctx, TDX_K_STENCIL);
    uint8_t* code = static_cast<uint8_t*>(ctx->scratch(TDX_S_D, n));
    if (!code) return TDX_ERR_NOMEM;
    hipLaunchKernelGGL(dinf_code_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_ang, n, inx, ang_nodata, -1.0e30f, R.d_a2, code);   // (no outlets mode here)
    hipLaunchKernelGGL(rev_setup_kernel, dim3((inx + 63) / 64, (iny + 3) / 4), dim3(256), 0, s, code, inx, iny, R.info);
    if (stats) stats->launches[TDX_K_STENCIL]++;
    return TDX_OK;
}

int updep_impl(tdx_context* ctx, const Strip& st, float* d_ang, float ang_nodata, const double* dxc, const double* dyc, int32_t* d_dg, float* d_dep, tdx_stats* stats) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    const size_t first = size_t(st.y0) * size_t(st.nx), nown = size_t(st.y1 - st.y0) * size_t(st.nx);
    RevSetup R;
    int rc = rev_prepare(ctx, st, d_ang, ang_nodata, dxc, dyc, R, stats);
    if (rc != TDX_OK) return rc;
    hipLaunchKernelGGL(pack_aux_i_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_ang, d_dg, n, R.aux);
    hipLaunchKernelGGL(d8sweep::init_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, R.info, d_dep, first, nown, -1.0f);   // depNodata = -1 (src/DinfUpDependence.cpp:113)
    rc = strip_exchange<float>(ctx, st, d_dep, -1.0f);
    if (rc != TDX_OK) return rc;
    int64_t rounds = 0, launches = 0, outer = 1;
    {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        d8sweep::Arrays<UpDepAlg> A{d_dep, R.aux, nullptr, R.d_a2, R.info};
        rc = d8sweep::run(ctx, st, UpDepAlg{}, A, R.flags, R.counts, &rounds, &launches, &outer);
        if (rc != TDX_OK) return rc;
        hipLaunchKernelGGL(d8sweep::finish_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, d_dep, first, nown, -1.0f);
        if (stats) stats->launches[TDX_K_ACCUM] += launches;
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) { stt->rounds = outer; stt->cells_evaluated = rounds; }
    return TDX_OK;
}

int revacc_impl(tdx_context* ctx, const Strip& st, float* d_ang, float ang_nodata, const double* dxc, const double* dyc, float* d_w, float w_nodata, float* d_racc,
                float* d_dmax, tdx_stats* stats) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    const size_t first = size_t(st.y0) * size_t(st.nx), nown = size_t(st.y1 - st.y0) * size_t(st.nx);
    RevSetup R;
    int rc = rev_prepare(ctx, st, d_ang, ang_nodata, dxc, dyc, R, stats);
    if (rc != TDX_OK) return rc;
    float2* rec = static_cast<float2*>(ctx->scratch(TDX_S_C, n * 8));
    if (!rec) return TDX_ERR_NOMEM;
    hipLaunchKernelGGL(pack_aux_f_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_ang, d_w, n, R.aux);
    hipLaunchKernelGGL(rev_init2_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, R.info, rec, first, nown);
    {
        const float2 oc = RevAccAlg::outside();
        uint2 ob;
        memcpy(&ob, &oc, sizeof(ob));
        rc = strip_exchange<uint2>(ctx, st, reinterpret_cast<uint2*>(rec), ob);
        if (rc != TDX_OK) return rc;
    }
    int64_t rounds = 0, launches = 0, outer = 1;
    {
        TdxSpan sp(ctx, TDX_K_ACCUM);
        d8sweep::Arrays<RevAccAlg> A{rec, R.aux, nullptr, R.d_a2, R.info};
        rc = d8sweep::run(ctx, st, RevAccAlg{w_nodata}, A, R.flags, R.counts, &rounds, &launches, &outer);
        if (rc != TDX_OK) return rc;
        hipLaunchKernelGGL(rev_unpack2_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, rec, first, nown, d_racc, d_dmax);
        if (stats) stats->launches[TDX_K_ACCUM] += launches;
    }
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) { stt->rounds = outer; stt->cells_evaluated = rounds; }
    return TDX_OK;
}

bool too_big(int64_t nx, int64_t rows) { return nx > 0x7fffffff || rows > 0x7ffffff0 || uint64_t(nx) * uint64_t(rows) > 0xffffffffull; }

}  // namespace

extern "C" int tdx_dinfupdependence_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc,
                                        const int32_t* d_dg, float* d_dep, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_dg || !d_dep || !dxc || !dyc || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfupdependence_dev: bad argument");
    if (too_big(nx, ny)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return updep_impl(ctx, strip_single(int(nx), int(ny)), const_cast<float*>(d_ang), ang_nodata, dxc, dyc, const_cast<int32_t*>(d_dg), d_dep, stats);
}
extern "C" int tdx_dinfupdependence_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata, const double* dxc,
                                          const double* dyc, int32_t* d_dg, float* d_dep, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_dg || !d_dep || !dxc || !dyc || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfupdependence_strip: bad argument");
    if (too_big(nx, ny_local + 2)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return updep_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_ang, ang_nodata, dxc, dyc, d_dg, d_dep, stats);
}
extern "C" int tdx_dinfupdependence(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc,
                                    const int32_t* dg, float* dep, tdx_stats* stats) {
    if (!ctx || !ang || !dg || !dep || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfupdependence: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    int32_t* d_g = static_cast<int32_t*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_o = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    if (!d_a || !d_g || !d_o) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_a, ang, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_g, dg, n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int rc = tdx_dinfupdependence_dev(ctx, d_a, nx, ny, ang_nodata, dxc, dyc, d_g, d_o, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(dep, d_o, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}

extern "C" int tdx_dinfrevaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc,
                                    const float* d_w, float w_nodata, float* d_racc, float* d_dmax, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_w || !d_racc || !d_dmax || !dxc || !dyc || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfrevaccum_dev: bad argument");
    if (too_big(nx, ny)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return revacc_impl(ctx, strip_single(int(nx), int(ny)), const_cast<float*>(d_ang), ang_nodata, dxc, dyc, const_cast<float*>(d_w), w_nodata, d_racc, d_dmax, stats);
}
extern "C" int tdx_dinfrevaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata, const double* dxc,
                                      const double* dyc, float* d_w, float w_nodata, float* d_racc, float* d_dmax, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_w || !d_racc || !d_dmax || !dxc || !dyc || nx <= 0 || ny_local <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfrevaccum_strip: bad argument");
    if (too_big(nx, ny_local + 2)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return revacc_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_ang, ang_nodata, dxc, dyc, d_w, w_nodata, d_racc, d_dmax, stats);
}
extern "C" int tdx_dinfrevaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc, const float* w,
                                float w_nodata, float* racc, float* dmax, tdx_stats* stats) {
    if (!ctx || !ang || !w || !racc || !dmax || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfrevaccum: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_w = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_r = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    float* d_m = static_cast<float*>(ctx->scratch(TDX_S_IO3, n * 4));
    if (!d_a || !d_w || !d_r || !d_m) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_a, ang, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_w, w, n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int rc = tdx_dinfrevaccum_dev(ctx, d_a, nx, ny, ang_nodata, dxc, dyc, d_w, w_nodata, d_r, d_m, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(racc, d_r, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(dmax, d_m, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
