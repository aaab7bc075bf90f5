// Micro-benchmark: how long does ONE workgroup need to pull a 66x66 float tile (+ a 64x64 tile of a second array)
// out of a large row-major raster?  Prints average cycles (s_memtime) per tile for a few variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ W, const float* __restrict__ Z, int nx, int ny, int tiles_x, int ntiles,
                                         int tiles_per_block, unsigned long long* cyc, float* sink) {
    __shared__ float sV[66 * 67];
    const int tid = threadIdx.x, lx = tid & 63, wv = tid >> 6;
    float acc = 0.f;
    unsigned long long total = 0;
    for (int t = 0; t < tiles_per_block; t++) {
        const int tile = (blockIdx.x * tiles_per_block + t) * 97 % ntiles;   // scattered tiles
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int x0 = tx * 64, y0 = ty * 64;
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (VARIANT == 0) {   // the relax engine's pattern: 17 row loads + 1 side load of W, 16 row loads of Z
            float st[17], zz[16];
#pragma unroll
            for (int i = 0; i < 17; i++) {
                int hy = y0 - 1 + wv + 4 * i; hy = hy < 0 ? 0 : (hy >= ny ? ny - 1 : hy);
                st[i] = W[(long long)hy * nx + x0 + lx];
            }
            int sy = y0 - 1 + (tid >> 1); sy = sy < 0 ? 0 : (sy >= ny ? ny - 1 : sy);
            int sx = (tid & 1) ? x0 + 64 : x0 - 1; sx = sx < 0 ? 0 : (sx >= nx ? nx - 1 : sx);
            const float side = W[(long long)sy * nx + sx];
#pragma unroll
            for (int r = 0; r < 16; r++) { int gy = y0 + wv * 16 + r; gy = gy >= ny ? ny - 1 : gy; zz[r] = Z[(long long)gy * nx + x0 + lx]; }
#pragma unroll
            for (int i = 0; i < 17; i++) if (wv + 4 * i < 66) sV[(wv + 4 * i) * 67 + lx + 1] = st[i];
            if (tid < 132) sV[(tid >> 1) * 67 + ((tid & 1) ? 65 : 0)] = side;
#pragma unroll
            for (int r = 0; r < 16; r++) acc += zz[r];
        } else if (VARIANT == 1) {   // W only
            float st[17];
#pragma unroll
            for (int i = 0; i < 17; i++) {
                int hy = y0 - 1 + wv + 4 * i; hy = hy < 0 ? 0 : (hy >= ny ? ny - 1 : hy);
                st[i] = W[(long long)hy * nx + x0 + lx];
            }
#pragma unroll
            for (int i = 0; i < 17; i++) if (wv + 4 * i < 66) sV[(wv + 4 * i) * 67 + lx + 1] = st[i];
        } else if (VARIANT == 2) {   // 4 rows only (one load per thread)
            int hy = y0 + wv; 
            sV[wv * 67 + lx + 1] = W[(long long)hy * nx + x0 + lx];
        }
        __syncthreads();
        const unsigned long long t1 = __builtin_readcyclecounter();
        total += t1 - t0;
        acc += sV[(tid * 7) % (66 * 67)];
    }
    if (tid == 0) atomicAdd(cyc, total);
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const size_t cells = size_t(n) * n;
    float *W, *Z, *sink; unsigned long long* cyc;
    CK(hipMalloc(&W, cells * 4)); CK(hipMalloc(&Z, cells * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&cyc, 8));
    CK(hipMemset(W, 0, cells * 4)); CK(hipMemset(Z, 0, cells * 4));
    const int tiles_x = n / 64, ntiles = tiles_x * tiles_x;
    for (int blocks : {64, 256, 1024}) {
        for (int variant = 0; variant < 3; variant++) {
            const int tpb = 16;
            CK(hipMemset(cyc, 0, 8));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a));
            if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, W, Z, n, n, tiles_x, ntiles, tpb, cyc, sink);
            if (variant == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, W, Z, n, n, tiles_x, ntiles, tpb, cyc, sink);
            if (variant == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, W, Z, n, n, tiles_x, ntiles, tpb, cyc, sink);
            CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            unsigned long long h = 0; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("n=%d blocks=%4d variant=%d: %.0f cycles per tile load (s_memtime), kernel %.3f ms => %.2f us per tile per block\n", n, blocks, variant,
                   double(h) / (double(blocks) * tpb), ms, ms * 1e3 / tpb);
        }
    }
    return 0;
}
