// tdxbench - developer harness over the C ABI (not one of the reference's tools): times the stages of the hot path on a
// synthetic DEM that is generated on the device, with no Python start-up cost, and prints one JSON object with the stage
// times, the library's per-kernel-class times and a CRC-32 per output raster (so that A/B runs of a kernel change can be
// compared bit for bit without an oracle).  bench.py stays the measurement of record.
//
//   tdxbench d8   [-n 16384] [-steps 3] [-warmup 1] [-crc]        PitRemove -> D8FlowDir -> AreaD8
//   tdxbench dinf [-n 16384] [-steps 2] [-warmup 1] [-crc]        (PitRemove once) DinfFlowDir -> AreaDinf
//   tdxbench decay [-nx 65536 -ny 8192] [-steps 1] [-crc]         DinfDecayAccum with weights, decay multipliers and 64 outlets
#include <zlib.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "taudem_amd.h"

static tdx_context* g_ctx = nullptr;
#define CK(expr)                                                                                         \
    do {                                                                                                 \
        const int _rc = (expr);                                                                          \
        if (_rc != TDX_OK) {                                                                             \
            fprintf(stderr, "tdxbench: %s -> %d (%s)\n", #expr, _rc, tdx_last_error(g_ctx));             \
            exit(2);                                                                                     \
        }                                                                                                \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T>
static T* dalloc(size_t n) {
    void* p = nullptr;
    CK(tdx_device_alloc(g_ctx, uint64_t(n) * sizeof(T), &p));
    return static_cast<T*>(p);
}
static unsigned long crc_dev(const void* d, size_t bytes) {
    std::vector<unsigned char> h(bytes);
    CK(tdx_copy_to_host(g_ctx, h.data(), d, bytes));
    unsigned long c = crc32(0L, Z_NULL, 0);
    for (size_t off = 0; off < bytes; off += size_t(1) << 30) c = crc32(c, h.data() + off, uInt(std::min(bytes - off, size_t(1) << 30)));
    return c;
}
static void print_stats(const char* name, const tdx_stats& s) {
    printf("\"%s\": {\"ms_total\": %.3f, \"ms_class\": [%.3f, %.3f, %.3f, %.3f, %.3f, %.3f, %.3f], \"rounds\": %lld, \"flats_initial\": %lld, \"flat_iterations\": %lld, "
           "\"levels_fall\": %lld, \"levels_rise\": %lld, \"cells_evaluated\": %lld}",
           name, s.ms_total, s.ms_kernel[0], s.ms_kernel[1], s.ms_kernel[2], s.ms_kernel[3], s.ms_kernel[4], s.ms_kernel[5], s.ms_kernel[6], (long long)s.rounds,
           (long long)s.flats_initial, (long long)s.flat_iterations, (long long)s.levels_fall, (long long)s.levels_rise, (long long)s.cells_evaluated);
}
static int64_t base_wl(int64_t n) { int64_t wl = 2; while (wl * 2 < n) wl *= 2; return wl; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: tdxbench d8|dinf|decay [-n N | -nx X -ny Y] [-steps K] [-warmup W] [-seed S] [-crc]\n"); return 1; }
    const std::string mode = argv[1];
    int64_t nx = 0, ny = 0;
    int steps = 3, warmup = 1, want_crc = 0;
    uint64_t seed = 1234;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "-n") && i + 1 < argc) nx = ny = atoll(argv[++i]);
        else if (!strcmp(argv[i], "-nx") && i + 1 < argc) nx = atoll(argv[++i]);
        else if (!strcmp(argv[i], "-ny") && i + 1 < argc) ny = atoll(argv[++i]);
        else if (!strcmp(argv[i], "-steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-seed") && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "-crc")) want_crc = 1;
        else { fprintf(stderr, "tdxbench: unknown argument %s\n", argv[i]); return 1; }
    }
    if (nx == 0) { nx = mode == "decay" ? 65536 : 16384; ny = mode == "decay" ? 8192 : 16384; }
    if (ny == 0) ny = nx;
    CK(tdx_context_create(0, &g_ctx));
    const size_t n = size_t(nx) * size_t(ny);
    std::vector<double> dxc(size_t(ny), 30.0), dyc(size_t(ny), 30.0);
    float* dem = dalloc<float>(n);
    float* fel = dalloc<float>(n);
    CK(tdx_synth_dem_dev(g_ctx, seed, nx, ny, 0, 0, base_wl(std::max(nx, ny)), dem));
    tdx_stats s1{}, s2{}, s3{};
    printf("{\"mode\": \"%s\", \"nx\": %lld, \"ny\": %lld, \"steps\": %d, ", mode.c_str(), (long long)nx, (long long)ny, steps);
    if (mode == "d8") {
        int16_t* p = dalloc<int16_t>(n);
        float* sd8 = dalloc<float>(n);
        float* ad8 = dalloc<float>(n);
        auto step = [&]() {
            CK(tdx_pitremove_dev(g_ctx, dem, nx, ny, -9999.0f, nullptr, 0, fel, &s1));
            CK(tdx_d8flowdir_dev(g_ctx, fel, nx, ny, TDX_FEL_NODATA, dxc.data(), dyc.data(), p, sd8, &s2));
            CK(tdx_aread8_dev(g_ctx, p, nx, ny, TDX_P_NODATA, nullptr, -9999.0f, 1, nullptr, nullptr, -1, ad8, &s3));
        };
        if (getenv("TDXBENCH_ADDR"))   // where the rasters landed (process-to-process timing differences of the streaming kernels: DESIGN.md 4.1)
            fprintf(stderr, "tdxbench: dem %p fel %p p %p sd8 %p ad8 %p\n", (void*)dem, (void*)fel, (void*)p, (void*)sd8, (void*)ad8);
        for (int i = 0; i < warmup; i++) step();
        CK(tdx_synchronize(g_ctx));
        const double t0 = now_ms();
        double a1 = 0, a2 = 0, a3 = 0;
        for (int i = 0; i < steps; i++) { step(); a1 += s1.ms_total; a2 += s2.ms_total; a3 += s3.ms_total; }
        CK(tdx_synchronize(g_ctx));
        const double ms = (now_ms() - t0) / steps;
        printf("\"ms_per_step\": %.3f, \"mcells_per_s\": %.1f, \"pitremove_ms\": %.3f, \"d8flowdir_ms\": %.3f, \"aread8_ms\": %.3f, ", ms, double(n) / ms / 1e3, a1 / steps,
               a2 / steps, a3 / steps);
        print_stats("pitremove", s1); printf(", "); print_stats("d8flowdir", s2); printf(", "); print_stats("aread8", s3);
        if (want_crc)
            printf(", \"crc\": {\"fel\": %lu, \"p\": %lu, \"sd8\": %lu, \"ad8\": %lu}", crc_dev(fel, n * 4), crc_dev(p, n * 2), crc_dev(sd8, n * 4), crc_dev(ad8, n * 4));
    } else if (mode == "dinf" || mode == "decay") {
        CK(tdx_pitremove_dev(g_ctx, dem, nx, ny, -9999.0f, nullptr, 0, fel, &s1));
        float* ang = dalloc<float>(n);
        float* slp = dem;   // the raw surface is not needed any more
        float* sca = dalloc<float>(n);
        if (mode == "dinf") {
            auto step = [&]() {
                CK(tdx_dinfflowdir_dev(g_ctx, fel, nx, ny, TDX_FEL_NODATA, dxc.data(), dyc.data(), ang, slp, &s2));
                CK(tdx_areadinf_dev(g_ctx, ang, nx, ny, TDX_ANG_NODATA, dxc.data(), dyc.data(), nullptr, 1, nullptr, nullptr, -1, sca, &s3));
            };
            for (int i = 0; i < warmup; i++) step();
            CK(tdx_synchronize(g_ctx));
            const double t0 = now_ms();
            double a2 = 0, a3 = 0;
            for (int i = 0; i < steps; i++) { step(); a2 += s2.ms_total; a3 += s3.ms_total; }
            CK(tdx_synchronize(g_ctx));
            const double ms = (now_ms() - t0) / steps;
            printf("\"ms_per_step\": %.3f, \"mcells_per_s\": %.1f, \"dinfflowdir_ms\": %.3f, \"areadinf_ms\": %.3f, ", ms, double(n) / ms / 1e3, a2 / steps, a3 / steps);
            print_stats("dinfflowdir", s2); printf(", "); print_stats("areadinf", s3);
            if (want_crc) printf(", \"crc\": {\"ang\": %lu, \"slp\": %lu, \"sca\": %lu}", crc_dev(ang, n * 4), crc_dev(slp, n * 4), crc_dev(sca, n * 4));
        } else {
            // BASELINE.json configs[4] on one strip-sized raster: weights ~ U[0,1), decay multipliers ~ U[0.9,1) (SURVEY.md 8d iii), 64 outlets
            CK(tdx_dinfflowdir_dev(g_ctx, fel, nx, ny, TDX_FEL_NODATA, dxc.data(), dyc.data(), ang, slp, &s2));
            float* w = fel;     // the filled surface is not needed any more
            float* dm = slp;
            {
                std::vector<float> h(n);
                uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
                auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return float((x >> 40) * (1.0 / 16777216.0)); };
                for (size_t i = 0; i < n; i++) h[i] = rnd();
                CK(tdx_copy_to_device(g_ctx, w, h.data(), n * 4));
                for (size_t i = 0; i < n; i++) h[i] = 0.9f + 0.1f * rnd();
                CK(tdx_copy_to_device(g_ctx, dm, h.data(), n * 4));
            }
            std::vector<int32_t> ox(64), oy(64);
            for (int i = 0; i < 64; i++) { ox[size_t(i)] = int32_t((int64_t(i % 8) * 2 + 1) * nx / 16); oy[size_t(i)] = int32_t((int64_t(i / 8) * 2 + 1) * ny / 16); }
            auto step = [&]() {
                CK(tdx_dinfdecayaccum_dev(g_ctx, ang, nx, ny, TDX_ANG_NODATA, dxc.data(), dyc.data(), dm, -9999.0f, w, 1, ox.data(), oy.data(), 64, sca, &s3));
            };
            for (int i = 0; i < warmup; i++) step();
            CK(tdx_synchronize(g_ctx));
            const double t0 = now_ms();
            for (int i = 0; i < steps; i++) step();
            CK(tdx_synchronize(g_ctx));
            const double ms = (now_ms() - t0) / steps;
            printf("\"ms_per_step\": %.3f, \"mcells_per_s\": %.1f, ", ms, double(n) / ms / 1e3);
            print_stats("dinfdecayaccum", s3);
            if (want_crc) printf(", \"crc\": {\"dsca\": %lu}", crc_dev(sca, n * 4));
        }
    } else {
        fprintf(stderr, "tdxbench: unknown mode %s\n", mode.c_str());
        return 1;
    }
    printf("}\n");
    tdx_context_destroy(g_ctx);
    return 0;
}
