// Micro-benchmark: effective bandwidth of the tile engine's LOAD phase in isolation - every 64x64 tile of two n x n
// float rasters is read once by a 256-thread workgroup (4 bands of 16 rows, lane = column, one dword per lane and row:
// 16 + 16 loads per lane, what relax_tile_reg does), at the occupancy of the real kernel (4 waves per SIMD) unless
// noted.  Variants add one ingredient of the real kernel at a time:
//   V0 loads only            V1 + the 132 scattered halo-column cells   V2 + write-back of one array (all cells)
//   V3 V0 with tiles in a scattered order     V4 V0 at 8 waves per SIMD     V5 V2 + V1 + a spin of ~8000 cycles (the sweeps)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__device__ __forceinline__ void body(const float* __restrict__ W, float* __restrict__ Wout, const float* __restrict__ Z, int nx, int tiles_x, int ntiles, float* sink) {
    const int tid = threadIdx.x, lx = tid & 63, wv = tid >> 6;
    float acc = 0.f;
    for (int it = blockIdx.x; it < ntiles; it += gridDim.x) {
        const int tile = (V == 3) ? int((unsigned(it) * 40503u) % unsigned(ntiles)) : it;
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int x0 = tx * 64, y0 = ty * 64 + wv * 16;
        float a[16], b[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { const size_t i = size_t(y0 + r) * nx + x0 + lx; a[r] = W[i]; b[r] = Z[i]; }
        float side = 0.f;
        if (V == 1 || V == 5) {
            int sy = ty * 64 - 1 + (tid >> 1), sx = (tid & 1) ? x0 + 64 : x0 - 1;
            sy = sy < 0 ? 0 : (sy >= nx ? nx - 1 : sy); sx = sx < 0 ? 0 : (sx >= nx ? nx - 1 : sx);
            if (tid < 132) side = W[size_t(sy) * nx + sx];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) acc += a[r] * b[r];
        acc += side;
        if (V == 5) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < 8000ull) acc += 1e-9f;
        }
        if (V == 2 || V == 5) {
#pragma unroll
            for (int r = 0; r < 16; r++) Wout[size_t(y0 + r) * nx + x0 + lx] = a[r] + acc;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k4(const float* W, float* Wout, const float* Z, int nx, int tiles_x, int ntiles, float* sink) {
    body<V>(W, Wout, Z, nx, tiles_x, ntiles, sink);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k8(const float* W, float* Wout, const float* Z, int nx, int tiles_x, int ntiles, float* sink) {
    body<0>(W, Wout, Z, nx, tiles_x, ntiles, sink);
}

// V6: the level-field relaxation's access width - one int16 (level) and one uint8 (mask) per lane and row, 128- and 64-byte row segments.
// Known bytes per launch: 3 per cell (what a FETCH_SIZE pass over this kernel is compared with).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k16(const short* __restrict__ L, const unsigned char* __restrict__ M, int nx, int tiles_x,
                                                                                       int ntiles, float* sink) {
    const int tid = threadIdx.x, lx = tid & 63, wv = tid >> 6;
    int acc = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int x0 = tx * 64, y0 = ty * 64 + wv * 16;
        short a[16]; unsigned char b[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { const size_t i = size_t(y0 + r) * nx + x0 + lx; a[r] = L[i]; b[r] = M[i]; }
#pragma unroll
        for (int r = 0; r < 16; r++) acc += a[r] * b[r];
    }
    if (acc == 12345678) sink[0] = float(acc);
}

template <class K>
static void run(const char* name, K kern, const float* W, float* Wout, const float* Z, int n, int blocks, float* sink) {
    const int tiles_x = n / 64, ntiles = tiles_x * tiles_x;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, W, Wout, Z, n, tiles_x, ntiles, sink);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("%s blocks %5d: %.3f ms  %.0f GB/s of tile reads\n", name, blocks, best, double(n) * n * 8 / best * 1e-6);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const size_t cells = size_t(n) * n;
    float *W, *W2, *Z, *sink;
    CK(hipMalloc(&W, cells * 4)); CK(hipMalloc(&W2, cells * 4)); CK(hipMalloc(&Z, cells * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(W, 0, cells * 4)); CK(hipMemset(W2, 0, cells * 4)); CK(hipMemset(Z, 0, cells * 4));
    for (int blocks : {1024, 2048}) {
        run("V0 loads only          ", k4<0>, W, W2, Z, n, blocks, sink);
        run("V1 + halo columns      ", k4<1>, W, W2, Z, n, blocks, sink);
        run("V2 + write-back        ", k4<2>, W, W2, Z, n, blocks, sink);
        run("V2' write-back in place", k4<2>, W, W, Z, n, blocks, sink);
        run("V3 scattered tile order", k4<3>, W, W2, Z, n, blocks, sink);
        run("V4 8 waves per SIMD    ", k8, W, W2, Z, n, blocks, sink);
        run("V5 halo+spin+write-back", k4<5>, W, W, Z, n, blocks, sink);
        {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const short*>(W), reinterpret_cast<const unsigned char*>(Z), n, n / 64, (n / 64) * (n / 64), sink);
                CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
                float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
                best = ms < best ? ms : best;
            }
            printf("V6 int16 + uint8 loads   blocks %5d: %.3f ms  %.0f GB/s of tile reads (3 B per cell)\n", blocks, best, double(n) * n * 3 / best * 1e-6);
        }
    }
    return 0;
}
