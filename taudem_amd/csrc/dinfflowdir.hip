// placeholder - implemented after the D8 path is parity-green
#include "context.hpp"
extern "C" int tdx_dinfflowdir_dev(tdx_context* ctx, const float*, int64_t, int64_t, float, const double*, const double*, float*, float*, tdx_stats*) {
    return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfflowdir: not implemented yet");
}
extern "C" int tdx_dinfflowdir(tdx_context* ctx, const float*, int64_t, int64_t, float, const double*, const double*, float*, float*, tdx_stats*) {
    return tdx_fail(ctx, TDX_ERR_ARG, "tdx_dinfflowdir: not implemented yet");
}
