// DinfConcLimAccum (dsllArea, src/DinfConcLimAccum.cpp:61-326) and DinfTransLimAccum (tlaccum, src/DinfTransLimAccum.cpp:61-372) on
// gfx950 - SURVEY.md 8(f) rank 4: the two D-infinity flow-algebra tools that sweep the dependency graph of AreaDinf FORWARD with a
// different per-cell expression.
//
// Same dependency-driven evaluation as AreaDinf: a cell is evaluated when all cells that send flow to it are done, by pulling them
// in k = 1..8 order with the reference's mixed float / double arithmetic; a cell's value depends only on its contributors' values,
// never on the schedule.  The tile dependency sweep of d8_sweep.hpp applies with
//   dependency mask = neighbours whose flow reaches the cell (initNeighborDinfup, src/commonLib.cpp:99-131)
//   release mask    = the (at most two) neighbours the cell sends flow to
// Everything a receiver needs from a contributor travels in the contributor's 16-byte record - its result(s) AND its static
// inputs (angle, q, dm) - so that one LDS read per neighbour serves the evaluation and one store publishes a finished cell:
//   DinfConcLimAccum   {ctpt, q, dm, angle}          own indicator dg as the cell's aux word
//   DinfTransLimAccum  {tla, csout, angle, cin->dep} own {tsup, tc} as aux; the 4th slot holds the cell's input concentration until
//                      the cell is evaluated and its deposition afterwards (neither is read by other cells)
// The proportion of a contributor is recomputed from its angle (prop() of dinf_prop.hpp: fp64, per-row atan2 from the host libm).
// Outlets (-o): the sweep runs on the re-coded angles of dinf_outlets.hpp.
#include <cmath>
#include <cstring>
#include <vector>

#include "context.hpp"
#include "d8_sweep.hpp"
#include "device_common.hpp"
#include "dinf_outlets.hpp"
#include "dinf_prop.hpp"

namespace {
using namespace tdxk;

constexpr unsigned FINFO_P1 = 1u << 12, FINFO_P2 = 1u << 15;

// Per cell: [0:8) contributors (dependency and value), [8] a neighbour is missing (off the raster or without angle: edge
// contamination), [9:12) s1 - 1, [12] / [15] prop > 0 towards s1 / s1 % 8 + 1, [13] the cell participates
__global__ __launch_bounds__(256) void fwd_setup_kernel(const uint8_t* __restrict__ code, int nx, int ny, uint32_t* __restrict__ info) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    unsigned c[9];
    dinf_code_window(code, nx, ny, x, y, c);   // (codes: pass 1, dinf_prop.hpp - two fp64 divisions per cell instead of ten)
    unsigned inf = 0;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
        if (dinf_code_missing(c[k])) inf |= d8sweep::INFO_CON;   // (a sink on a cell without angle: missing for the contamination test, sends nothing)
        if (c[k] == DINF_CODE_NODATA) continue;
        const int kk = (k + 4) % 8;
        if (dinf_code_sends(c[k], kk == 0 ? 8 : kk)) inf |= 1u << (k - 1);   // `float p > 0` of src/commonLib.cpp:99 == the sender's own proportion > 0
    }
    if (c[0] != DINF_CODE_NODATA && (c[0] & DINF_CODE_PART)) {
        inf |= d8sweep::INFO_PART | ((c[0] & 7u) << 9);
        if (c[0] & DINF_CODE_P1) inf |= FINFO_P1;
        if (c[0] & DINF_CODE_P2) inf |= FINFO_P2;
    }
    info[size_t(y) * size_t(nx) + size_t(x)] = inf;
}
__device__ __forceinline__ unsigned fwd_rel_mask(unsigned inf) {
    const int s1 = int((inf >> 9) & 7u) + 1, s2 = s1 % 8 + 1;
    return ((inf & FINFO_P1) ? 1u << (s1 - 1) : 0u) | ((inf & FINFO_P2) ? 1u << (s2 - 1) : 0u);
}

struct ConcLimAlg {   // src/DinfConcLimAccum.cpp:226-262; record = {ctpt, q, dm, angle}
    using Cell = float4;
    using Aux = float;                           // indicator grid value (int bits)
    static constexpr bool HAS_AUX = true, HAS_DIST = false, HAS_ROWS = true;
    static constexpr int kBulkSweeps = 6;   // (measured at 16384^2, ConcLim / TransLim: 3: 130.6 / 147.6 ms, 6: 129.3 / 144.9, 12: 131.6 / 146.0)
    static constexpr unsigned kBulkUntil = 16;
    static constexpr int kMinWaves32 = 4;
    static constexpr int kMaxRelease = 2;
    float dm_nodata, q_nodata, csol;
    int contcheck;
    static __device__ __forceinline__ float head(const float4& c) { return c.x; }
    static __host__ __device__ __forceinline__ float4 outside() { return make_float4(TDX_ANG_NODATA, 0.f, 0.f, TDX_ANG_NODATA); }
    static __device__ __forceinline__ unsigned rel_mask(unsigned inf) { return fwd_rel_mask(inf); }
    template <class L>
    __device__ __forceinline__ void eval(L& S, int c, int cl, int ly, unsigned inf, const Cell (&nb)[9]) const {
        float4 me = S.v[cl];
        float res = TDX_ANG_NODATA;
        if (me.y > 0.) {
            bool con = false;
            if (__float_as_int(S.aux[c]) > 0) res = csol;
            else {
                con = (inf & d8sweep::INFO_CON) != 0u;
                float conc = 0.0f;
#pragma unroll
                for (int k = 1; k <= 8; k++) {
                    if (!((inf >> (k - 1)) & 1u)) continue;
                    const float4 n = nb[k];
                    const double p = prop_dev(n.w, (k + 4) % 8, S.rows[ly + 1 + d2(k)]);
                    if (is_nodata_f(n.x, TDX_ANG_NODATA) || is_nodata_f(n.z, dm_nodata) || is_nodata_f(n.y, q_nodata)) con = true;
                    else conc = (float)(conc + p * n.x * n.y * n.z);   // double product, float accumulator
                }
                conc = conc / me.y;
                res = conc;
            }
            if (con && contcheck == 1) res = TDX_ANG_NODATA;
        }
        me.x = res;
        S.v[cl] = me;
    }
};

struct TransLimAlg {   // src/DinfTransLimAccum.cpp:236-307; record = {tla, csout, angle, cin -> dep}
    using Cell = float4;
    using Aux = float2;                          // {tsup, tc}
    static constexpr bool HAS_AUX = true, HAS_DIST = false, HAS_ROWS = true;
    static constexpr int kBulkSweeps = 6;   // (measured at 16384^2, ConcLim / TransLim: 3: 130.6 / 147.6 ms, 6: 129.3 / 144.9, 12: 131.6 / 146.0)
    static constexpr unsigned kBulkUntil = 16;
    static constexpr int kMinWaves32 = 4;
    static constexpr int kMaxRelease = 2;
    float tsup_nodata, tc_nodata, cin_nodata;
    int usec, contcheck;
    const float* cin_src;   // the input concentration raster (null without -cs): what slot w held before the evaluation turned it into the deposition
    __device__ __forceinline__ void unevaluate(float4& me, size_t idx) const { if (usec) me.w = cin_src[idx]; }   // (sweep verifier, d8_sweep.hpp)
    static __device__ __forceinline__ float head(const float4& c) { return c.x; }
    static __host__ __device__ __forceinline__ float4 outside() { return make_float4(TDX_ANG_NODATA, TDX_ANG_NODATA, TDX_ANG_NODATA, TDX_ANG_NODATA); }
    static __device__ __forceinline__ unsigned rel_mask(unsigned inf) { return fwd_rel_mask(inf); }
    template <class L>
    __device__ __forceinline__ void eval(L& S, int c, int cl, int ly, unsigned inf, const Cell (&nb)[9]) const {
        float4 me = S.v[cl];
        const float2 a = S.aux[c];
        const float cin = me.w;
        float tla = TDX_ANG_NODATA, dep = TDX_ANG_NODATA, cso = TDX_ANG_NODATA;
        if (!is_nodata_f(a.x, tsup_nodata) && !is_nodata_f(a.y, tc_nodata) && (usec == 0 || !is_nodata_f(cin, cin_nodata))) {
            float transin = 0.f, loadin = 0.f;
            bool con = (inf & d8sweep::INFO_CON) != 0u;
#pragma unroll
            for (int k = 1; k <= 8; k++) {
                if (!((inf >> (k - 1)) & 1u)) continue;
                const float4 n = nb[k];
                const double p = prop_dev(n.z, (k + 4) % 8, S.rows[ly + 1 + d2(k)]);
                float nt = 0.0f;
                if (is_nodata_f(n.x, TDX_ANG_NODATA)) con = true;
                else { nt = n.x; transin = (float)(transin + p * nt); }
                if (usec == 1) {
                    if (is_nodata_f(n.y, TDX_ANG_NODATA)) con = true;
                    else loadin = (float)(loadin + p * nt * n.y);
                }
            }
            const float tsupp = a.x, tcc = a.y;
            float transout;
            if ((transin + tsupp) > tcc) { transout = tcc; dep = transin + tsupp - transout; }
            else { transout = transin + tsupp; dep = 0.f; }
            tla = transout;
            if (usec == 1) {
                float loadout;
                if (transout < transin) loadout = (transin > 0) ? loadin * transout / transin : 0.f;
                else loadout = loadin + cin * (transout - transin);
                cso = (transout > 0.) ? (float)(loadout / transout) : 0.f;
            }
            if (con && contcheck == 1) { dep = TDX_ANG_NODATA; tla = TDX_ANG_NODATA; cso = TDX_ANG_NODATA; }
        }
        me.x = tla; me.y = cso; me.w = dep;
        S.v[cl] = me;
    }
};

// records of the owned rows: pending where the cell participates, "no value" elsewhere (never read: such a cell is nobody's contributor)
__global__ __launch_bounds__(256) void conc_pack_kernel(const uint32_t* __restrict__ info, const float* __restrict__ ANG, const float* __restrict__ Q,
                                                        const float* __restrict__ DM, size_t first, size_t n, float4* __restrict__ rec) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= first + n) return;
    rec[i] = make_float4((info[i] & d8sweep::INFO_PART) ? __uint_as_float(d8sweep::PENDING_BITS) : TDX_ANG_NODATA, Q[i], DM[i], ANG[i]);
}
__global__ __launch_bounds__(256) void conc_aux_kernel(const int16_t* __restrict__ DG, size_t n, float* __restrict__ aux) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) aux[i] = __int_as_float(int(DG[i]));
}
__global__ __launch_bounds__(256) void conc_unpack_kernel(const float4* __restrict__ rec, size_t first, size_t n, float* __restrict__ ctpt) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= first + n) return;
    const float4 r = rec[i];
    ctpt[i] = d8sweep::pending(r.x) ? TDX_ANG_NODATA : r.x;   // pending: on or below a cycle - never queued by the reference either
}
__global__ __launch_bounds__(256) void trans_pack_kernel(const uint32_t* __restrict__ info, const float* __restrict__ ANG, const float* __restrict__ CIN,
                                                         size_t first, size_t n, float4* __restrict__ rec) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= first + n) return;
    rec[i] = make_float4((info[i] & d8sweep::INFO_PART) ? __uint_as_float(d8sweep::PENDING_BITS) : TDX_ANG_NODATA, TDX_ANG_NODATA, ANG[i], CIN ? CIN[i] : 0.f);
}
__global__ __launch_bounds__(256) void trans_aux_kernel(const float* __restrict__ TSUP, const float* __restrict__ TC, size_t n, float2* __restrict__ aux) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) aux[i] = make_float2(TSUP[i], TC[i]);
}
__global__ __launch_bounds__(256) void trans_unpack_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ info, size_t first, size_t n,
                                                           float* __restrict__ tla, float* __restrict__ dep, float* __restrict__ cso) {
    const size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= first + n) return;
    const float4 r = rec[i];
    const bool done = (info[i] & d8sweep::INFO_PART) && !d8sweep::pending(r.x);   // (a cell that was never evaluated still holds cin in the 4th slot)
    tla[i] = done ? r.x : TDX_ANG_NODATA;
    dep[i] = done ? r.w : TDX_ANG_NODATA;
    if (cso) cso[i] = done ? r.y : TDX_ANG_NODATA;
}

struct FwdSetup {
    RowProp* d_rows = nullptr;
    double* d_a2 = nullptr;
    uint32_t* info = nullptr;
    float4* rec = nullptr;
    uint32_t* flags = nullptr;
    unsigned long long* counts = nullptr;
    float* ang_use = nullptr;
};
// common front part: halo rows of the angle grid, per-row tables, outlets, info words
int fwd_prepare(tdx_context* ctx, const Strip& st, float* d_ang, float ang_nodata, const double* dxc, const double* dyc, const int32_t* outlet_x,
                const int32_t* outlet_y, int64_t n_outlets, FwdSetup& R, tdx_stats* stats) {
    if (n_outlets > 0 && (!outlet_x || !outlet_y)) return tdx_fail(ctx, TDX_ERR_ARG, "outlets missing");
    hipStream_t s = ctx->stream;
    const int inx = st.nx, iny = st.ny_arr;
    const size_t n = size_t(inx) * size_t(iny);
    std::vector<RowProp> rows(static_cast<size_t>(iny));
    std::vector<double> a2(static_cast<size_t>(iny));
    for (int j = 0; j < iny; j++) { a2[size_t(j)] = atan2(dyc[j], dxc[j]); rows[size_t(j)].a2 = a2[size_t(j)]; rows[size_t(j)].dx = dxc[j]; }
    const tilek::TileGeom geom = tilek::make_geom(inx, iny, st.y0, st.y1);
    const size_t ntiles = size_t(geom.tiles_x) * size_t(geom.tiles_y);
    R.d_rows = static_cast<RowProp*>(ctx->scratch(TDX_S_J, rows.size() * sizeof(RowProp)));
    R.d_a2 = static_cast<double*>(ctx->scratch(TDX_S_K, a2.size() * 8));
    R.info = static_cast<uint32_t*>(ctx->scratch(TDX_S_A, n * 4));
    R.rec = static_cast<float4*>(ctx->scratch(TDX_S_C, n * 16));
    R.counts = static_cast<unsigned long long*>(ctx->scratch(TDX_S_M, size_t(tilek::COUNT_RING) * 16));
    if (!R.d_rows || !R.d_a2 || !R.info || !R.rec || !R.counts) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(R.d_rows, rows.data(), rows.size() * sizeof(RowProp), hipMemcpyHostToDevice, s));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(R.d_a2, a2.data(), a2.size() * 8, hipMemcpyHostToDevice, s));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(s));   // the tables are locals
    ctx->begin_call(stats);
    strip_mark(ctx, st, "dinfconclimaccum / dinftranslimaccum");
    int rc = strip_exchange<float>(ctx, st, d_ang, ang_nodata);   // flowData->share()
    if (rc != TDX_OK) return rc;
    R.ang_use = d_ang;
    if (n_outlets >= 0) {
        rc = dinf_outlet_recode(ctx, st, d_ang, ang_nodata, R.d_rows, outlet_x, outlet_y, n_outlets, &R.ang_use, stats);
        if (rc != TDX_OK) return rc;
    }
    R.flags = static_cast<uint32_t*>(ctx->scratch(TDX_S_L, ntiles * 4 * (1 + tilek::SCHED_LIST_WORDS)));   // (after the closure, which uses the same slot)
    if (!R.flags) return TDX_ERR_NOMEM;
    TdxSpan sp(ctx, TDX_K_STENCIL);
    uint8_t* code = static_cast<uint8_t*>(ctx->scratch(TDX_S_D, n));
    if (!code) return TDX_ERR_NOMEM;
    hipLaunchKernelGGL(dinf_code_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, R.ang_use, n, inx, ang_nodata, TDX_ANG_OUTSIDE, R.d_a2, code);
    hipLaunchKernelGGL(fwd_setup_kernel, dim3((inx + 63) / 64, (iny + 3) / 4), dim3(256), 0, s, code, inx, iny, R.info);
    if (stats) stats->launches[TDX_K_STENCIL]++;
    return TDX_OK;
}

template <class Alg>
int fwd_sweep(tdx_context* ctx, const Strip& st, Alg alg, FwdSetup& R, const typename Alg::Aux* aux, tdx_stats* stats, int64_t* rounds, int64_t* outer) {
    {   // records of the neighbours' boundary rows
        const float4 oc = Alg::outside();
        uint4 ob;
        memcpy(&ob, &oc, sizeof(ob));
        int rc = strip_exchange<uint4>(ctx, st, reinterpret_cast<uint4*>(R.rec), ob);
        if (rc != TDX_OK) return rc;
    }
    int64_t launches = 0;
    TdxSpan sp(ctx, TDX_K_ACCUM);
    d8sweep::Arrays<Alg> A{R.rec, aux, nullptr, R.d_a2, R.info};
    int rc = d8sweep::run(ctx, st, alg, A, R.flags, R.counts, rounds, &launches, outer);
    if (rc != TDX_OK) return rc;
    if (stats) stats->launches[TDX_K_ACCUM] += launches;
    return TDX_OK;
}

int conclim_impl(tdx_context* ctx, const Strip& st, float* d_ang, float ang_nodata, const double* dxc, const double* dyc, const float* d_dm, float dm_nodata,
                 const int16_t* d_dg, const float* d_q, float q_nodata, float csol, int contcheck, const int32_t* outlet_x, const int32_t* outlet_y,
                 int64_t n_outlets, float* d_ctpt, tdx_stats* stats) {
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    const size_t first = size_t(st.y0) * size_t(st.nx), nown = size_t(st.y1 - st.y0) * size_t(st.nx);
    FwdSetup R;
    int rc = fwd_prepare(ctx, st, d_ang, ang_nodata, dxc, dyc, outlet_x, outlet_y, n_outlets, R, stats);
    if (rc != TDX_OK) return rc;
    float* aux = static_cast<float*>(ctx->scratch(TDX_S_B, n * 4));
    if (!aux) return TDX_ERR_NOMEM;
    hipLaunchKernelGGL(conc_aux_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_dg, n, aux);
    hipLaunchKernelGGL(conc_pack_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, R.info, R.ang_use, d_q, d_dm, first, nown, R.rec);
    int64_t rounds = 0, outer = 1;
    rc = fwd_sweep(ctx, st, ConcLimAlg{dm_nodata, q_nodata, csol, contcheck}, R, aux, stats, &rounds, &outer);
    if (rc != TDX_OK) return rc;
    hipLaunchKernelGGL(conc_unpack_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, R.rec, first, nown, d_ctpt);
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) { stt->rounds = outer; stt->cells_evaluated = rounds; }
    return TDX_OK;
}

int translim_impl(tdx_context* ctx, const Strip& st, float* d_ang, float ang_nodata, const double* dxc, const double* dyc, const float* d_tsup, float tsup_nodata,
                  const float* d_tc, float tc_nodata, const float* d_cin, float cin_nodata, int contcheck, const int32_t* outlet_x, const int32_t* outlet_y,
                  int64_t n_outlets, float* d_tla, float* d_dep, float* d_cso, tdx_stats* stats) {
    if ((d_cin == nullptr) != (d_cso == nullptr)) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinftranslimaccum: the concentration input and output go together");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t n = size_t(st.nx) * size_t(st.ny_arr);
    const size_t first = size_t(st.y0) * size_t(st.nx), nown = size_t(st.y1 - st.y0) * size_t(st.nx);
    FwdSetup R;
    int rc = fwd_prepare(ctx, st, d_ang, ang_nodata, dxc, dyc, outlet_x, outlet_y, n_outlets, R, stats);
    if (rc != TDX_OK) return rc;
    float2* aux = static_cast<float2*>(ctx->scratch(TDX_S_B, n * 8));
    if (!aux) return TDX_ERR_NOMEM;
    hipLaunchKernelGGL(trans_aux_kernel, dim3(tdx_blocks_for(n, 256)), dim3(256), 0, s, d_tsup, d_tc, n, aux);
    hipLaunchKernelGGL(trans_pack_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, R.info, R.ang_use, d_cin, first, nown, R.rec);
    int64_t rounds = 0, outer = 1;
    rc = fwd_sweep(ctx, st, TransLimAlg{tsup_nodata, tc_nodata, cin_nodata, d_cin ? 1 : 0, contcheck, d_cin}, R, aux, stats, &rounds, &outer);
    if (rc != TDX_OK) return rc;
    hipLaunchKernelGGL(trans_unpack_kernel, dim3(tdx_blocks_for(nown, 256)), dim3(256), 0, s, R.rec, R.info, first, nown, d_tla, d_dep, d_cso);
    TDX_HIP_CHECK(ctx, hipGetLastError());
    tdx_stats* stt = stats;
    ctx->end_call();
    if (stt) { stt->rounds = outer; stt->cells_evaluated = rounds; }
    return TDX_OK;
}

bool too_big(int64_t nx, int64_t rows) { return nx > 0x7fffffff || rows > 0x7ffffff0 || uint64_t(nx) * uint64_t(rows) > 0xffffffffull; }

}  // namespace

extern "C" int tdx_dinfconclimaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc,
                                        const float* d_dm, float dm_nodata, const int16_t* d_dg, const float* d_q, float q_nodata, float csol, int contcheck,
                                        const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_ctpt, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_dm || !d_dg || !d_q || !d_ctpt || !dxc || !dyc || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfconclimaccum_dev: bad argument");
    if (too_big(nx, ny)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return conclim_impl(ctx, strip_single(int(nx), int(ny)), const_cast<float*>(d_ang), ang_nodata, dxc, dyc, d_dm, dm_nodata, d_dg, d_q, q_nodata, csol, contcheck,
                        outlet_x, outlet_y, n_outlets, d_ctpt, stats);
}
extern "C" int tdx_dinfconclimaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata, const double* dxc,
                                          const double* dyc, const float* d_dm, float dm_nodata, const int16_t* d_dg, const float* d_q, float q_nodata, float csol,
                                          int contcheck, const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets, float* d_ctpt, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_dm || !d_dg || !d_q || !d_ctpt || !dxc || !dyc || nx <= 0 || ny_local <= 0)
        return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfconclimaccum_strip: bad argument");
    if (too_big(nx, ny_local + 2)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return conclim_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_ang, ang_nodata, dxc, dyc, d_dm, dm_nodata, d_dg, d_q, q_nodata, csol, contcheck, outlet_x,
                        outlet_row, n_outlets, d_ctpt, stats);
}
extern "C" int tdx_dinfconclimaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc, const float* dm,
                                    float dm_nodata, const int16_t* dg, const float* q, float q_nodata, float csol, int contcheck, const int32_t* outlet_x,
                                    const int32_t* outlet_y, int64_t n_outlets, float* ctpt, tdx_stats* stats) {
    if (!ctx || !ang || !dm || !dg || !q || !ctpt || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfconclimaccum: bad argument");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_m = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_q = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    int16_t* d_g = static_cast<int16_t*>(ctx->scratch(TDX_S_IO3, n * 2));
    float* d_o = static_cast<float*>(ctx->scratch(TDX_S_IO4, n * 4));
    if (!d_a || !d_m || !d_q || !d_g || !d_o) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_a, ang, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_m, dm, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_q, q, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_g, dg, n * 2, hipMemcpyHostToDevice, ctx->stream));
    const int rc = tdx_dinfconclimaccum_dev(ctx, d_a, nx, ny, ang_nodata, dxc, dyc, d_m, dm_nodata, d_g, d_q, q_nodata, csol, contcheck, outlet_x, outlet_y, n_outlets, d_o,
                                            stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctpt, d_o, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}

extern "C" int tdx_dinftranslimaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc,
                                         const float* d_tsup, float tsup_nodata, const float* d_tc, float tc_nodata, const float* d_cs, float cs_nodata, int contcheck,
                                         const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_tla, float* d_tdep, float* d_ctpt,
                                         tdx_stats* stats) {
    if (!ctx || !d_ang || !d_tsup || !d_tc || !d_tla || !d_tdep || !dxc || !dyc || nx <= 0 || ny <= 0)
        return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinftranslimaccum_dev: bad argument");
    if (too_big(nx, ny)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return translim_impl(ctx, strip_single(int(nx), int(ny)), const_cast<float*>(d_ang), ang_nodata, dxc, dyc, d_tsup, tsup_nodata, d_tc, tc_nodata, d_cs, cs_nodata,
                         contcheck, outlet_x, outlet_y, n_outlets, d_tla, d_tdep, d_ctpt, stats);
}
extern "C" int tdx_dinftranslimaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata, const double* dxc,
                                           const double* dyc, const float* d_tsup, float tsup_nodata, const float* d_tc, float tc_nodata, const float* d_cs,
                                           float cs_nodata, int contcheck, const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets, float* d_tla,
                                           float* d_tdep, float* d_ctpt, tdx_stats* stats) {
    if (!ctx || !d_ang || !d_tsup || !d_tc || !d_tla || !d_tdep || !dxc || !dyc || nx <= 0 || ny_local <= 0)
        return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinftranslimaccum_strip: bad argument");
    if (too_big(nx, ny_local + 2)) return tdx_fail(ctx, TDX_ERR_ARG, "raster larger than 2^32 cells per device strip");
    return translim_impl(ctx, strip_from_comm(comm, int(nx), int(ny_local)), d_ang, ang_nodata, dxc, dyc, d_tsup, tsup_nodata, d_tc, tc_nodata, d_cs, cs_nodata, contcheck,
                         outlet_x, outlet_row, n_outlets, d_tla, d_tdep, d_ctpt, stats);
}
extern "C" int tdx_dinftranslimaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata, const double* dxc, const double* dyc,
                                     const float* tsup, float tsup_nodata, const float* tc, float tc_nodata, const float* cs, float cs_nodata, int contcheck,
                                     const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* tla, float* tdep, float* ctpt, tdx_stats* stats) {
    if (!ctx || !ang || !tsup || !tc || !tla || !tdep || nx <= 0 || ny <= 0) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinftranslimaccum: bad argument");
    if ((cs == nullptr) != (ctpt == nullptr)) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinftranslimaccum: the concentration input and output go together");
    const size_t n = size_t(nx) * size_t(ny);
    float* d_a = static_cast<float*>(ctx->scratch(TDX_S_IO0, n * 4));
    float* d_s = static_cast<float*>(ctx->scratch(TDX_S_IO1, n * 4));
    float* d_c = static_cast<float*>(ctx->scratch(TDX_S_IO2, n * 4));
    float* d_t = static_cast<float*>(ctx->scratch(TDX_S_IO3, n * 4));
    float* d_d = static_cast<float*>(ctx->scratch(TDX_S_IO4, n * 4));
    float* d_ci = cs ? static_cast<float*>(ctx->scratch(TDX_S_E, n * 4)) : nullptr;
    float* d_co = cs ? static_cast<float*>(ctx->scratch(TDX_S_F, n * 4)) : nullptr;
    if (!d_a || !d_s || !d_c || !d_t || !d_d || (cs && (!d_ci || !d_co))) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_a, ang, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_s, tsup, n * 4, hipMemcpyHostToDevice, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_c, tc, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (cs) TDX_HIP_CHECK(ctx, hipMemcpyAsync(d_ci, cs, n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int rc = tdx_dinftranslimaccum_dev(ctx, d_a, nx, ny, ang_nodata, dxc, dyc, d_s, tsup_nodata, d_c, tc_nodata, d_ci, cs_nodata, contcheck, outlet_x, outlet_y,
                                             n_outlets, d_t, d_d, d_co, stats);
    if (rc != TDX_OK) return rc;
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(tla, d_t, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipMemcpyAsync(tdep, d_d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (cs) TDX_HIP_CHECK(ctx, hipMemcpyAsync(ctpt, d_co, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
