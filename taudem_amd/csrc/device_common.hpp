// Device-side helpers shared by the stage kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdlib>

namespace tdxk {

// neighbour tables d1 (dx) / d2 (dy), index 1..8: E NE N NW W SW S SE   (src/commonLib.h:83-84)
__device__ __constant__ const int kD1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1};
__device__ __constant__ const int kD2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};

// As arithmetic on packed 2-bit fields, not table reads: a lookup in a __constant__ array with a RUN-TIME index compiles to a
// global_load (0.3-1 us) - in the dependency walks that was two or three dependent loads on every hop of the critical path.
// (With a compile-time k both forms fold to the constant.)
__device__ __forceinline__ int d1(int k) { return int((0x24069u >> (2 * k)) & 3u) - 1; }
__device__ __forceinline__ int d2(int k) { return int((0x2a405u >> (2 * k)) & 3u) - 1; }

#define TDX_MINEPS 1E-5f   /* src/commonLib.h:81 */

// linearpart<float>::isNodata for an in-grid value (src/linearpart.h:476)
__device__ __forceinline__ bool is_nodata_f(float v, float nodata) { return fabsf((float)(v - nodata)) < TDX_MINEPS; }
// linearpart<short>::isNodata: (float)(short-short) is an exact integer, so the test is equality
__device__ __forceinline__ bool is_nodata_s(int16_t v, int16_t nodata) { return v == nodata; }

// workgroup barrier that orders LDS traffic only: waits for this wave's outstanding LDS operations, NOT for its global loads / stores (which
// __syncthreads(), a workgroup-scope release + acquire, does)
__device__ __forceinline__ void tdx_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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


// Block-wide (256 threads, all must call) reservation of list slots: thread contributes `c` items and
// gets the index of its first item; ONE global atomic per block.  A single hot counter sustains only
// ~90 M atomics/s on MI355X, so full-grid passes must reserve per block, not per wave.
__device__ __forceinline__ unsigned long long block_reserve(unsigned c, unsigned long long* __restrict__ counter) {
    __shared__ unsigned s_wave[4];
    __shared__ unsigned long long s_base;
    const int lane = int(threadIdx.x & 63), w = int(threadIdx.x >> 6);
    unsigned v = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    __syncthreads();                       // previous use of s_wave / s_base is over
    if (lane == 63) s_wave[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned wave_off = 0;
    for (int i = 0; i < w; i++) wave_off += s_wave[i];
    return s_base + wave_off + (v - c);
}

// Sharded event counter: 32 counters one cache line apart; the host (or a reader kernel) sums them.
constexpr int kCounterShards = 32;
constexpr int kCounterStride = 16;   // in unsigned long long = 128 bytes
__device__ __forceinline__ void sharded_add(unsigned long long* __restrict__ shards, unsigned long long v) {
    if (v) atomicAdd(shards + size_t(blockIdx.x + blockIdx.y * 7u) % kCounterShards * kCounterStride, v);
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

// Column block of this workgroup in a streaming pass whose grid is (column blocks, row bands).  Block b of a launch runs on XCD b % 8 (observed -
// MI355X_MICROARCH.md, workgroup dispatch; a matter of speed only), so in launch order two neighbouring column blocks sit on DIFFERENT L2s: the halo
// columns both read are fetched twice and the cache lines at their seam are written in two halves.  With the grid's x extent padded to a multiple
// of 8 (tdx_xcd_grid_x) the blocks of one XCD take one contiguous eighth of the columns instead; a padding block gets -1.  xmap == 0: launch order.
__device__ __forceinline__ int xcd_block_x(int nblocks, int xmap) {
    if (!xmap) return int(blockIdx.x);
    const int b = int(blockIdx.x & 7u) * int(gridDim.x >> 3) + int(blockIdx.x >> 3);
    return b < nblocks ? b : -1;
}

}  // namespace tdxk

static inline unsigned tdx_blocks_for(uint64_t n, unsigned threads) { return unsigned((n + threads - 1) / threads); }

// host side of tdxk::xcd_block_x: the x extent of the grid and the kernel's xmap argument (TDX_XCD_MAP_OFF=1: launch order - A/B hook)
static inline bool tdx_xcd_map() { static const bool on = getenv("TDX_XCD_MAP_OFF") == nullptr; return on; }
static inline unsigned tdx_xcd_grid_x(unsigned nblocks) { return tdx_xcd_map() ? ((nblocks + 7u) & ~7u) : nblocks; }

