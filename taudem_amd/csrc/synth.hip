// Synthetic fractal DEM on the device (benchmark input; see synth_dem.h).
#include "context.hpp"
#include "device_common.hpp"
#include "synth_dem.h"

namespace {
__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, int nx, int ny, int64_t x0, int64_t y0, int64_t base_wl, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    out[size_t(y) * size_t(nx) + size_t(x)] = tdx_synth_elev(seed, x0 + x, y0 + y, base_wl);
}
}  // namespace

extern "C" int tdx_synth_dem_dev(tdx_context* ctx, uint64_t seed, int64_t nx, int64_t ny, int64_t x0, int64_t y0,
                                 int64_t base_wavelength, float* d_out) {
    if (!ctx || !d_out || nx <= 0 || ny <= 0 || base_wavelength < 2) return tdx_fail(ctx, TDX_ERR_ARG, "tdx_synth_dem_dev: bad argument");
    TDX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    dim3 grid(unsigned((nx + 63) / 64), unsigned((ny + 3) / 4));
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, ctx->stream, seed, int(nx), int(ny), x0, y0, base_wavelength, d_out);
    TDX_HIP_CHECK(ctx, hipGetLastError());
    TDX_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return TDX_OK;
}
