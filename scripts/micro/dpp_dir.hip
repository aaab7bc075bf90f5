// Checks the direction of the DPP wavefront shifts the register-resident tile kernel relies on (tile_relax.hpp):
// lane_left(x, edge)[i] == x[i-1] (lane 0: edge), lane_right(x, edge)[i] == x[i+1] (lane 63: edge).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int x = 100 + int(threadIdx.x), edge = -7;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(edge, x, 0x138, 0xf, 0xf, false);        // wave_shr:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(edge, x, 0x130, 0xf, 0xf, false);   // wave_shl:1
}
int main() {
    int* d; int h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        bad += h[i] != (i ? 100 + i - 1 : -7);
        bad += h[64 + i] != (i < 63 ? 100 + i + 1 : -7);
    }
    printf("dpp wave shifts: %s (shr: %d %d %d ... %d | shl: %d %d ... %d %d)\n", bad ? "UNEXPECTED" : "ok", h[0], h[1], h[2], h[63], h[64], h[65], h[126], h[127]);
    return bad != 0;
}
