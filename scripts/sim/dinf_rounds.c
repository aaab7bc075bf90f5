/* CPU model of the D-infinity tile dependency sweep's ROUND structure (scripts/sim_dinf_rounds.py drives it): for a given angle raster,
 * how many rounds and tile activations does a schedule need in which a tile finishes, per activation, every cell whose contributors are
 * final (in-tile chains included), and a tile is re-activated in round r + 1 when a neighbouring tile finished a contributor of one of its
 * cells in round r?   round(c) = max over contributors n of round(n) + (tile(n) != tile(c));   activations(tile) = #distinct rounds of its cells.
 * With a HALO of h cells a tile may also evaluate (redundantly, not written back) cells of its neighbours within h cells of its border:
 *   a cell's round under halo h is modelled through the bounding box of its WHOLE upstream set: if that box lies within the tile grown by h, the
 *   tile can produce the cell in round 0 whatever tiles the box touches.  (Lower bound of the benefit: later rounds gain as well.)
 * Reads: nx ny, then nx*ny float32 angles (nodata = -FLT_MAX, no direction = -1) from stdin.  proportions as in prop() (src/commonLib.cpp:75-90). */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static const int d1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1}, d2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};
int main(int argc, char** argv) {
    long nx, ny;
    if (scanf("%ld %ld\n", &nx, &ny) != 2) return 1;
    const long N = nx * ny;
    float* ang = malloc(N * 4);
    if (fread(ang, 4, N, stdin) != (size_t)N) return 2;
    /* receivers: angle a in [ (k-1) pi/4, k pi/4 ] -> neighbours k and k+1 (square cells) */
    int8_t* r1 = malloc(N); int8_t* r2 = malloc(N);
    int32_t* indeg = calloc(N, 4);
    const double q = atan2(1.0, 1.0);
    for (long c = 0; c < N; c++) {
        r1[c] = r2[c] = 0;
        const float a = ang[c];
        if (!(a >= 0.f)) continue;
        int k = (int)floor((double)a / q);
        if (k > 7) k = 7;
        const double f = (double)a / q - k;      /* share of neighbour k + 2 (1-based k+1 .. ) */
        const int ka = k + 1, kb = (k + 1) % 8 + 1;
        const long x = c % nx, y = c / nx;
        if (1.0 - f > 1e-5) { long xn = x + d1[ka], yn = y + d2[ka]; if (xn >= 0 && xn < nx && yn >= 0 && yn < ny && ang[xn + yn * nx] > -FLT_MAX) { r1[c] = ka; indeg[xn + yn * nx]++; } }
        if (f > 1e-5) { long xn = x + d1[kb], yn = y + d2[kb]; if (xn >= 0 && xn < nx && yn >= 0 && yn < ny && ang[xn + yn * nx] > -FLT_MAX) { r2[c] = kb; indeg[xn + yn * nx]++; } }
    }
    long* order = malloc(N * 8); long no = 0;
    int32_t* deg = malloc(N * 4); memcpy(deg, indeg, N * 4);
    for (long c = 0; c < N; c++) if (ang[c] > -FLT_MAX && deg[c] == 0) order[no++] = c;
    for (long i = 0; i < no; i++) {
        const long c = order[i], x = c % nx, y = c / nx;
        for (int j = 0; j < 2; j++) { const int k = j ? r2[c] : r1[c]; if (!k) continue; const long n = (x + d1[k]) + (y + d2[k]) * nx; if (--deg[n] == 0) order[no++] = n; }
    }
    fprintf(stderr, "%ld cells with data in dependency order (of %ld)\n", no, N);
    int32_t* rnd = malloc(N * 4);
    int16_t *bx0 = malloc(N * 2), *bx1 = malloc(N * 2), *by0 = malloc(N * 2), *by1 = malloc(N * 2);
    int32_t* depth = malloc(N * 4);
    for (int a = 1; a < argc; a++) {
        const int TS = atoi(argv[a]);
        const long tx = (nx + TS - 1) / TS, ty = (ny + TS - 1) / TS;
        for (long c = 0; c < N; c++) { rnd[c] = 0; depth[c] = 0; bx0[c] = bx1[c] = (int16_t)(c % nx); by0[c] = by1[c] = (int16_t)(c / nx); }
        for (long i = 0; i < no; i++) {
            const long c = order[i], x = c % nx, y = c / nx, t = (x / TS) + (y / TS) * tx;
            for (int j = 0; j < 2; j++) {
                const int k = j ? r2[c] : r1[c]; if (!k) continue;
                const long xn = x + d1[k], yn = y + d2[k], n = xn + yn * nx, tn = (xn / TS) + (yn / TS) * tx;
                const int r = rnd[c] + (tn != t);
                if (r > rnd[n]) rnd[n] = r;
                if (depth[c] + 1 > depth[n]) depth[n] = depth[c] + 1;
                if (bx0[c] < bx0[n]) bx0[n] = bx0[c];
                if (bx1[c] > bx1[n]) bx1[n] = bx1[c];
                if (by0[c] < by0[n]) by0[n] = by0[c];
                if (by1[c] > by1[n]) by1[n] = by1[c];
            }
        }
        int maxr = 0; long maxd = 0;
        for (long i = 0; i < no; i++) { if (rnd[order[i]] > maxr) maxr = rnd[order[i]]; if (depth[order[i]] > maxd) maxd = depth[order[i]]; }
        long* cells_in = calloc(maxr + 1, 8); long* tiles_in = calloc(maxr + 1, 8);
        /* activations: distinct rounds per tile */
        char* seen = calloc((size_t)tx * ty * (size_t)(maxr + 1 > 4096 ? 1 : 1), 1);
        (void)seen;
        long act = 0;
        {   /* per tile: bitmap of rounds (rounds can be many: use a small open hash per tile via sorting) */
            int32_t* tr = malloc(no * 8); long m = 0;
            for (long i = 0; i < no; i++) { const long c = order[i]; cells_in[rnd[c]]++; tr[2 * m] = (int32_t)(((c % nx) / TS) + ((c / nx) / TS) * tx); tr[2 * m + 1] = rnd[c]; m++; }
            /* count distinct (tile, round) pairs with a hash set */
            const size_t H = 1ull << 26; uint64_t* hs = calloc(H, 8);
            for (long i = 0; i < m; i++) {
                const uint64_t key = ((uint64_t)(uint32_t)tr[2 * i] << 24 | (uint32_t)tr[2 * i + 1]) + 1;
                size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 38);
                while (hs[h] && hs[h] != key) h = (h + 1) & (H - 1);
                if (!hs[h]) { hs[h] = key; act++; tiles_in[tr[2 * i + 1]]++; }
            }
            free(hs); free(tr);
        }
        printf("TS %d: tiles %ld, rounds %d, longest path %ld cells, tile activations %ld (%.2f per tile)\n", TS, tx * ty, maxr + 1, maxd, act, (double)act / (tx * ty));
        printf("   cells finished in round 0..7: "); for (int r = 0; r <= 7 && r <= maxr; r++) printf("%ld ", cells_in[r]); printf("\n   tiles active in round 0..11: "); for (int r = 0; r <= 11 && r <= maxr; r++) printf("%ld ", tiles_in[r]); printf("\n");
        /* halo model: cells finished in round 0 when the tile is grown by h */
        for (int h = 2; h <= 16; h *= 2) {
            long r0 = 0;
            for (long i = 0; i < no; i++) {
                const long c = order[i], x = c % nx, y = c / nx;
                const long X0 = (x / TS) * TS - h, X1 = (x / TS) * TS + TS - 1 + h, Y0 = (y / TS) * TS - h, Y1 = (y / TS) * TS + TS - 1 + h;
                if (bx0[c] >= X0 && bx1[c] <= X1 && by0[c] >= Y0 && by1[c] <= Y1) r0++;
            }
            printf("   halo %2d: %ld cells (%.2f %%) could be finished in round 0 (plain: %ld = %.2f %%); redundant work x%.2f\n", h, r0, 100.0 * r0 / no, cells_in[0], 100.0 * cells_in[0] / no,
                   (double)(TS + 2 * h) * (TS + 2 * h) / ((double)TS * TS));
        }
        free(cells_in); free(tiles_in); free(seen);
    }
    return 0;
}
