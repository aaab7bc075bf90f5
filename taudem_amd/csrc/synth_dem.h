/* Deterministic synthetic fractal DEM (benchmark / test input; not part of the reference).
 *
 * the benchmark input is a seeded fractional-Brownian surface, elevations
 * 0-1000 m.  FFT synthesis does not scale to 65536^2, so this is a procedural multi-octave
 * lattice value-noise: an integer hash of (seed, octave, ix, iy), smoothstep interpolation,
 * fp64 accumulation, one cast to float32.  Every operation is an exactly-rounded IEEE +,-,*,/
 * (compile with -ffp-contract=off), so host (gcc) and device (hipcc) produce identical bits and
 * any strip of any size can be generated independently.
 *
 * Plain C99 subset; TDX_HD expands to __host__ __device__ under hipcc.
 */
#ifndef TDX_SYNTH_DEM_H
#define TDX_SYNTH_DEM_H
#include <stdint.h>

#ifndef TDX_HD
#if defined(__HIPCC__)
#define TDX_HD __host__ __device__
#else
#define TDX_HD
#endif
#endif

#define TDX_SYNTH_GAIN 0.574349177498517   /* 2^-0.8 : Hurst exponent 0.8 (spectral beta = 3.6) */

TDX_HD static inline uint64_t tdx_mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

/* lattice value in [0,1) */
TDX_HD static inline double tdx_lattice(uint64_t seed, int oct, int64_t ix, int64_t iy) {
    uint64_t h = tdx_mix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(oct + 1));
    h = tdx_mix64(h ^ ((uint64_t)ix * 0xD6E8FEB86659FD93ULL));
    h = tdx_mix64(h ^ ((uint64_t)iy * 0xA0761D6478BD642FULL));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

/* elevation of cell (x = column, y = row) of an n-cell-wide surface.
 * base_wl: wavelength of the first octave (a power of two, normally the next power of two >= n/2);
 * octaves halve the wavelength down to 2 cells. */
TDX_HD static inline float tdx_synth_elev(uint64_t seed, int64_t x, int64_t y, int64_t base_wl) {
    double sum = 0.0, norm = 0.0, amp = 1.0;
    int oct = 0;
    for (int64_t wl = base_wl; wl >= 2; wl >>= 1, oct++) {
        const int64_t ix = x / wl, iy = y / wl;            /* x,y >= 0 */
        const double tx = (double)(x - ix * wl) / (double)wl;
        const double ty = (double)(y - iy * wl) / (double)wl;
        const double sx = tx * tx * (3.0 - 2.0 * tx);
        const double sy = ty * ty * (3.0 - 2.0 * ty);
        const double v00 = tdx_lattice(seed, oct, ix, iy), v10 = tdx_lattice(seed, oct, ix + 1, iy);
        const double v01 = tdx_lattice(seed, oct, ix, iy + 1), v11 = tdx_lattice(seed, oct, ix + 1, iy + 1);
        const double a = v00 + sx * (v10 - v00);
        const double b = v01 + sx * (v11 - v01);
        sum = sum + amp * (a + sy * (b - a));
        norm = norm + amp;
        amp = amp * TDX_SYNTH_GAIN;
    }
    return (float)(1000.0 * (sum / norm));
}

static inline int64_t tdx_synth_base_wl(int64_t n) {
    int64_t wl = 2;
    while (wl * 2 < n) wl *= 2;      /* largest power of two < n, at least 2 */
    return wl;
}

#endif
