// Step model of the in-tile Kahn phase of ad8_tile_local_kernel (taudem_amd/csrc/aread8.hip) for a given D8 direction raster: a step = one LDS round trip of a
// wave; four waves per 64 x 64 tile, a lane owns a 16-row column segment and the sources (cells without in-tile contributors) in it.
//   nested   the walk loop inside the loop over the lane's sources (rounds 2-5a): a wave pays, source after source, for the longest walk of any of its lanes
//   flat     one loop per lane: a lane whose walk has ended starts its next source in the next step (round 5, second half: the default)
//   queue    idle lanes pull sources from a tile-wide list (two extra steps per source) - NOT built: the list would cost 8-16 KB of LDS per tile (6 -> 4 tiles per CU)
// usage: kahn_steps N p.bin   (N x N int16 directions, row-major; e.g. the restatement's p of a 2048^2 fractal DEM:
//        nested 271.2, flat 165.2, queue 97.1 steps per tile; 4 013 hops per tile; longest in-tile path 79.4 hops)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define TS 64
static const int d1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1}, d2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};
int n; int16_t* P;
static int tgt[TS*TS], indeg[TS*TS], arr[TS*TS];
int main(int argc, char** argv) {
    n = atoi(argv[1]);
    FILE* f = fopen(argv[2], "rb"); P = malloc((size_t)n*n*2); fread(P, 2, (size_t)n*n, f); fclose(f);
    long long tot_q = 0, tot_nested = 0, tot_flat = 0, tot_hops = 0, tot_longest = 0; int tiles = 0;
    for (int ty = 0; ty < n/TS; ty++) for (int tx = 0; tx < n/TS; tx++) {
        // topology
        for (int ly = 0; ly < TS; ly++) for (int lx = 0; lx < TS; lx++) {
            int p = P[(size_t)(ty*TS+ly)*n + tx*TS+lx]; int t = -1;
            if (p >= 1 && p <= 8) { int x = lx + d1[p], y = ly + d2[p]; if (x >= 0 && x < TS && y >= 0 && y < TS) t = y*TS+x; }
            tgt[ly*TS+lx] = t;
        }
        memset(indeg, 0, sizeof indeg);
        for (int c = 0; c < TS*TS; c++) if (tgt[c] >= 0) indeg[tgt[c]]++;
        for (int mode = 0; mode < 3; mode++) {
            static int qlist[TS*TS]; int qn = 0, qhead = 0; int wait[256]; for (int l=0;l<256;l++) wait[l]=0;
            if (mode == 2) { for (int c = 0; c < TS*TS; c++) if (indeg[c] == 0) qlist[qn++] = c; }
            memset(arr, 0, sizeof arr);
            // lane state: 4 waves x 64 lanes; lane owns column lx, rows wv*16..+15
            int cur[256], walking[256]; uint32_t src[256]; int fresh[256];
            for (int l = 0; l < 256; l++) { int lx = l & 63, wv = l >> 6; src[l] = 0; cur[l] = -1; walking[l] = 0; fresh[l]=0;
                for (int r = 0; r < 16; r++) if (indeg[(wv*16+r)*TS+lx] == 0) src[l] |= 1u << r; }
            long long steps[4] = {0,0,0,0};
            // global synchronous time: every wave does one step per tick if it has work
            int done_w[4] = {0,0,0,0};
            int outer_active[4] = {0,0,0,0};   // nested: are we inside an inner loop
            long long hops = 0;
            for (long long tick = 0; ; tick++) {
                int alldone = 1;
                // collect arrivals in this tick then apply (atomics: order within tick arbitrary; emulate sequentially)
                for (int wv = 0; wv < 4; wv++) {
                    if (done_w[wv]) continue;
                    int any = 0;
                    if (mode == 0) {
                        // nested: if no lane walking, start next source for every lane that has one (costs a step: the LDS reads), else hop
                        int anywalk = 0; for (int l = wv*64; l < wv*64+64; l++) anywalk |= walking[l];
                        if (!anywalk) {
                            int anysrc = 0;
                            for (int l = wv*64; l < wv*64+64; l++) if (src[l]) { int r = __builtin_ctz(src[l]); src[l] &= src[l]-1; int c = ((l>>6)*16+r)*TS+(l&63); cur[l] = tgt[c]; walking[l] = cur[l] >= 0; anysrc = 1; }
                            if (!anysrc) { done_w[wv] = 1; continue; }
                            steps[wv]++; any = 1;
                        } else {
                            for (int l = wv*64; l < wv*64+64; l++) if (walking[l]) { int t = cur[l]; arr[t]++; hops++; if (arr[t] != indeg[t]) walking[l] = 0; else { cur[l] = tgt[t]; walking[l] = cur[l] >= 0; } }
                            steps[wv]++; any = 1;
                        }
                    } else if (mode == 2) {
                        int anywork = 0;
                        for (int l = wv*64; l < wv*64+64; l++) {
                            if (!walking[l] && wait[l] == 0 && qhead < qn) { cur[l] = qlist[qhead++]; wait[l] = 2; }
                            if (wait[l] > 0) { anywork = 1; wait[l]--; if (wait[l] == 0) { walking[l] = 1; fresh[l] = 1; } continue; }
                            if (walking[l]) { anywork = 1; int t = cur[l]; if (!fresh[l]) { arr[t]++; hops++; } int cont = fresh[l] || arr[t] == indeg[t]; fresh[l] = 0; if (!cont) walking[l] = 0; else { cur[l] = tgt[t]; walking[l] = cur[l] >= 0; } }
                        }
                        if (!anywork) { done_w[wv] = 1; continue; }
                        steps[wv]++; any = 1;
                    } else {
                        int anywork = 0;
                        for (int l = wv*64; l < wv*64+64; l++) {
                            if (!walking[l] && src[l]) { int r = __builtin_ctz(src[l]); src[l] &= src[l]-1; cur[l] = ((l>>6)*16+r)*TS+(l&63); walking[l] = 1; fresh[l] = 1; }
                            if (walking[l]) { anywork = 1; int t = cur[l]; if (!fresh[l]) { arr[t]++; hops++; } int cont = fresh[l] || arr[t] == indeg[t]; fresh[l] = 0; if (!cont) walking[l] = 0; else { cur[l] = tgt[t]; walking[l] = cur[l] >= 0; } }
                        }
                        if (!anywork) { done_w[wv] = 1; continue; }
                        steps[wv]++; any = 1;
                    }
                    if (any) alldone = 0;
                }
                if (alldone) break;
            }
            long long mx = 0; for (int w = 0; w < 4; w++) if (steps[w] > mx) mx = steps[w];
            if (mode == 0) tot_nested += mx; else if (mode == 1) tot_flat += mx; else tot_q += mx;
            if (mode == 0) tot_hops += hops;
        }
        // longest in-tile path
        { static int depth[TS*TS]; int order_done = 0; (void)order_done; int mxd = 0;
          for (int c = 0; c < TS*TS; c++) { int d = 0, t = c; while (tgt[t] >= 0 && d < 5000) { t = tgt[t]; d++; } if (d > mxd) mxd = d; depth[c] = d; }
          tot_longest += mxd; }
        tiles++;
    }
    printf("queue %.1f; ", (double)tot_q/tiles); printf("tiles %d: mean steps per tile nested %.1f, flat %.1f, hops per tile %.1f, longest path %.1f\n", tiles, (double)tot_nested/tiles, (double)tot_flat/tiles, (double)tot_hops/tiles, (double)tot_longest/tiles);
    return 0;
}
