// Device-side helpers shared by the stage kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace tdxk {

// neighbour tables d1 (dx) / d2 (dy), index 1..8: E NE N NW W SW S SE   (src/commonLib.h:83-84)
__device__ __constant__ const int kD1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1};
__device__ __constant__ const int kD2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};

__device__ __forceinline__ int d1(int k) { return kD1[k]; }
__device__ __forceinline__ int d2(int k) { return kD2[k]; }

#define TDX_MINEPS 1E-5f   /* src/commonLib.h:81 */

// linearpart<float>::isNodata for an in-grid value (src/linearpart.h:476)
__device__ __forceinline__ bool is_nodata_f(float v, float nodata) { return fabsf((float)(v - nodata)) < TDX_MINEPS; }
// linearpart<short>::isNodata: (float)(short-short) is an exact integer, so the test is equality
__device__ __forceinline__ bool is_nodata_s(int16_t v, int16_t nodata) { return v == nodata; }

__device__ __forceinline__ int lane_id() { return int(threadIdx.x & 63); }

// Wave-aggregated append: every lane with `pred` gets a unique slot in `list` (order within a
// wave = lane order; across waves = arrival order).  One atomic per wave.
__device__ __forceinline__ void wave_append(bool pred, uint32_t value, uint32_t* __restrict__ list,
                                            unsigned long long* __restrict__ counter) {
    const unsigned long long ballot = __ballot(pred);
    if (ballot == 0) return;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)ballot) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(ballot));
    base = __shfl(base, leader, 64);
    if (pred) {
        const unsigned long long below = ballot & ((1ull << lane) - 1ull);
        list[base + (unsigned long long)__popcll(below)] = value;
    }
}

// agent-scope relaxed accessors: sc1 loads/stores that bypass the per-CU L1 and are coherent
// across the 8 XCD L2s (MI355X_MICROARCH.md, inter-workgroup visibility)
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace tdxk

static inline unsigned tdx_blocks_for(uint64_t n, unsigned threads) { return unsigned((n + threads - 1) / threads); }
