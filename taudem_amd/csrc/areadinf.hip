// placeholder - implemented after the D8 path is parity-green
#include "context.hpp"
extern "C" int tdx_areadinf_dev(tdx_context* ctx, const float*, int64_t, int64_t, float, const double*, const double*, const float*, int,
                                const int32_t*, const int32_t*, int64_t, float*, tdx_stats*) { return tdx_fail(ctx, TDX_ERR_ARG, "not implemented yet"); }
extern "C" int tdx_areadinf(tdx_context* ctx, const float*, int64_t, int64_t, float, const double*, const double*, const float*, int,
                            const int32_t*, const int32_t*, int64_t, float*, tdx_stats*) { return tdx_fail(ctx, TDX_ERR_ARG, "not implemented yet"); }
extern "C" int tdx_dinfdecayaccum_dev(tdx_context* ctx, const float*, int64_t, int64_t, float, const double*, const double*, const float*, float,
                                      const float*, int, const int32_t*, const int32_t*, int64_t, float*, tdx_stats*) { return tdx_fail(ctx, TDX_ERR_ARG, "not implemented yet"); }
extern "C" int tdx_dinfdecayaccum(tdx_context* ctx, const float*, int64_t, int64_t, float, const double*, const double*, const float*, float,
                                  const float*, int, const int32_t*, const int32_t*, int64_t, float*, tdx_stats*) { return tdx_fail(ctx, TDX_ERR_ARG, "not implemented yet"); }
