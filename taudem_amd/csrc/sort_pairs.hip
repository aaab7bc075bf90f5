// Device radix sort of (uint32 key, uint32 value) pairs - rocPRIM through hipCUB.  Used once per AreaD8 call to put
// the few cells whose count exceeds 2^24 into dependency order (ascending count); not a hot operation.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include "context.hpp"

int tdx_sort_pairs_u32(tdx_context* ctx, int scratch_slot, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                       size_t n) {
    if (n == 0) return TDX_OK;
    size_t bytes = 0;
    TDX_HIP_CHECK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, int(n), 0, 32, ctx->stream));
    void* tmp = ctx->scratch(scratch_slot, bytes);
    if (!tmp) return TDX_ERR_NOMEM;
    TDX_HIP_CHECK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, int(n), 0, 32, ctx->stream));
    return TDX_OK;
}
