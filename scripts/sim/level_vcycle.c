// CPU model of the tile engine's round schedule for a breadth-first level field of flat resolution (flats.hpp), with and without
// COARSE CORRECTIONS (8 x 8 blocks, the review's proposal of round 5): how many rounds does the fine level need, how many the coarse one?
//
//   level_vcycle N lvl.i32 mask.u8 [after=8] [every=0] [max_cycles=1] [levels=1]
//
// lvl: N x N int32 markers as flatk::classify_kernel leaves them (-1 outside the queue, 0 in the queue and unreached, 1 / 2 seeds);
// mask: N x N eligibility masks (bit k-1: neighbour k may hand its level over).  Fixed point: v(c) = min(v(c), 1 + min over mask of v(n)).
// Round model: every active 64 x 64 tile runs to its local fixed point against the halo the PREVIOUS round left (tiles of a round run
// side by side on the GPU); a tile whose rim cell can improve a cell of a neighbouring tile activates it for the next round.
// Coarse level: a block is FULL when every cell's mask holds all in-block neighbours (so any two cells of it are <= 7 steps apart
// inside the block); M(B) >= max of the field over B; for a full neighbour A whose common edge / corner is open: M(B) <= M(A) + 8;
// restriction M(B) = min(max over B, 7 + min over B) of the current fine field; prolongation v = min(v, M(block)).
// Writes nothing; prints round / activation counts and checks the result against the run without corrections.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define TS 64
#define INF 0x3fffffff
static const int d1[9] = {0, 1, 1, 0, -1, -1, -1, 0, 1}, d2[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};
static inline int opp(int k) { return ((k - 1 + 4) & 7) + 1; }
static inline int val(int32_t g) { return g > 0 ? g : INF; }

static int g_chain = 0;          // 1: a tile that is activated and is FULL / PLAIN (every cell movable with all 8 neighbours eligible) is processed in the round that activates it, transitively
typedef struct {
    int nx, ny, inc, tx, ty;
    int32_t* v;
    uint8_t* m;
    uint8_t* act;    // per tile: 0 idle, 1 halo, 2 full
    uint8_t* nxt;
    uint8_t* plain;  // per tile (chain mode)
    long rounds, activations, changed_tiles, chained;
} Field;

static void field_init(Field* f, int nx, int ny, int inc, int32_t* v, uint8_t* m) {
    f->nx = nx; f->ny = ny; f->inc = inc; f->v = v; f->m = m;
    f->tx = (nx + TS - 1) / TS; f->ty = (ny + TS - 1) / TS;
    f->act = calloc((size_t)f->tx * f->ty, 1); f->nxt = calloc((size_t)f->tx * f->ty, 1);
    f->rounds = f->activations = f->changed_tiles = f->chained = 0;
    f->plain = calloc((size_t)f->tx * f->ty, 1);
    for (int ty = 0; ty < f->ty; ty++)
        for (int tx = 0; tx < f->tx; tx++) {
            int ok = (tx + 1) * TS <= nx && (ty + 1) * TS <= ny;
            for (int j = 0; j < TS && ok; j++)
                for (int i = 0; i < TS && ok; i++) ok = m[(size_t)(ty * TS + j) * nx + tx * TS + i] == 0xFF;
            f->plain[(size_t)ty * f->tx + tx] = (uint8_t)ok;
        }
}
static void activate_around(Field* f, int x, int y, uint8_t flag) {
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            int xn = x + dx, yn = y + dy;
            if (xn < 0 || yn < 0 || xn >= f->nx || yn >= f->ny) continue;
            uint8_t* a = &f->act[(size_t)(yn / TS) * f->tx + xn / TS];
            if (*a < flag) *a = flag;
        }
}

// one tile to its local fixed point; out = the tile's new values (TS*TS, row pitch TS); returns 1 if something changed
static int32_t W[(TS + 2) * (TS + 2)];
static int Q[TS * TS * 64];
static int relax_tile(const Field* f, int tx, int ty, int full, int32_t* out) {
    const int x0 = tx * TS, y0 = ty * TS, P = TS + 2, nx = f->nx, ny = f->ny, inc = f->inc;
    for (int j = 0; j < P; j++)
        for (int i = 0; i < P; i++) {
            int x = x0 - 1 + i, y = y0 - 1 + j;
            W[j * P + i] = (x >= 0 && y >= 0 && x < nx && y < ny) ? val(f->v[(size_t)y * nx + x]) : INF;
        }
    static uint8_t M[TS * TS], inq[TS * TS];
    for (int j = 0; j < TS; j++)
        for (int i = 0; i < TS; i++) {
            int x = x0 + i, y = y0 + j;
            M[j * TS + i] = (x < nx && y < ny) ? f->m[(size_t)y * nx + x] : 0;
        }
    memset(inq, 0, sizeof inq);
    int qh = 0, qt = 0, changed = 0;
    const int QN = TS * TS * 64;
    // first look: every cell (full) or the perimeter cells pull from their neighbours
    for (int j = 0; j < TS; j++)
        for (int i = 0; i < TS; i++) {
            if (!full && !(i == 0 || j == 0 || i == TS - 1 || j == TS - 1)) continue;
            uint8_t mk = M[j * TS + i];
            if (!mk) continue;
            int c = (j + 1) * P + i + 1, best = INF;
            for (int k = 1; k <= 8; k++)
                if (mk & (1u << (k - 1))) { int w = W[c + d2[k] * P + d1[k]]; if (w < best) best = w; }
            if (best < INF && best + inc < W[c]) { W[c] = best + inc; changed = 1; if (!inq[j * TS + i]) { inq[j * TS + i] = 1; Q[qt++ % QN] = j * TS + i; } }
        }
    while (qh != qt) {
        int t = Q[qh++ % QN]; inq[t] = 0;
        int j = t / TS, i = t % TS, c = (j + 1) * P + i + 1, vc = W[c];
        for (int k = 1; k <= 8; k++) {   // neighbour k of c receives from c if its mask selects direction opp(k)
            int ii = i + d1[k], jj = j + d2[k];
            if (ii < 0 || jj < 0 || ii >= TS || jj >= TS) continue;
            uint8_t mk = M[jj * TS + ii];
            if (!(mk & (1u << (opp(k) - 1)))) continue;
            int n = (jj + 1) * P + ii + 1;
            if (vc + inc < W[n]) { W[n] = vc + inc; changed = 1; if (!inq[jj * TS + ii]) { inq[jj * TS + ii] = 1; Q[qt++ % QN] = jj * TS + ii; } }
        }
    }
    if (changed)
        for (int j = 0; j < TS; j++)
            for (int i = 0; i < TS; i++) out[j * TS + i] = W[(j + 1) * P + i + 1];
    return changed;
}


// ---- macro blocks: aligned K x K blocks of full / plain tiles solved as ONE unit per round (what a closed-form rim update + lazy interior fill would do) ----
static int g_macro = 0;            // largest K (power of two), 0 = off
static int32_t* g_unit = NULL;     // per tile: -1 = ordinary tile, else index into g_blocks
typedef struct { int tx0, ty0, k; } Block;
static Block* g_blocks = NULL; static int g_nblocks = 0;
static void build_blocks(const Field* f) {
    const size_t nt = (size_t)f->tx * f->ty;
    g_unit = malloc(nt * sizeof(int32_t));
    for (size_t t = 0; t < nt; t++) g_unit[t] = -1;
    g_blocks = malloc(nt * sizeof(Block)); g_nblocks = 0;
    long covered = 0;
    for (int k = g_macro; k >= 2; k >>= 1)
        for (int by = 0; by + k <= f->ty; by += k)
            for (int bx = 0; bx + k <= f->tx; bx += k) {
                int ok = 1;
                for (int j = 0; j < k && ok; j++)
                    for (int i = 0; i < k && ok; i++) { size_t t = (size_t)(by + j) * f->tx + bx + i; ok = f->plain[t] && g_unit[t] < 0; }
                if (!ok) continue;
                for (int j = 0; j < k; j++) for (int i = 0; i < k; i++) g_unit[(size_t)(by + j) * f->tx + bx + i] = g_nblocks;
                g_blocks[g_nblocks++] = (Block){bx, by, k}; covered += (long)k * k;
            }
    long np = 0; for (size_t t = 0; t < nt; t++) np += f->plain[t];
    fprintf(stderr, "macro blocks (K <= %d): %d blocks cover %ld of %ld full / plain tiles\n", g_macro, g_nblocks, covered, np);
}
// a rectangular region of cells to its local fixed point against the ring around it (queue-based); out: w x h values; returns 1 if changed
static int relax_region(const Field* f, int x0, int y0, int w, int h, int32_t* out) {
    const int P = w + 2, nx = f->nx, ny = f->ny, inc = f->inc;
    int32_t* Wv = malloc((size_t)P * (h + 2) * sizeof(int32_t));
    for (int j = 0; j < h + 2; j++)
        for (int i = 0; i < P; i++) {
            int x = x0 - 1 + i, y = y0 - 1 + j;
            Wv[(size_t)j * P + i] = (x >= 0 && y >= 0 && x < nx && y < ny) ? val(f->v[(size_t)y * nx + x]) : INF;
        }
    int* q = malloc((size_t)w * h * sizeof(int) * 4); uint8_t* in = calloc((size_t)w * h, 1);
    size_t qh = 0, qt = 0; const size_t QN = (size_t)w * h * 4; int changed = 0;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            if (!(i == 0 || j == 0 || i == w - 1 || j == h - 1)) continue;
            const uint8_t mk = f->m[(size_t)(y0 + j) * nx + x0 + i];
            if (!mk) continue;
            size_t c = (size_t)(j + 1) * P + i + 1; int best = INF;
            for (int k = 1; k <= 8; k++) if (mk & (1u << (k - 1))) { int wv = Wv[c + d2[k] * P + d1[k]]; if (wv < best) best = wv; }
            if (best < INF && best + inc < Wv[c]) { Wv[c] = best + inc; changed = 1; if (!in[(size_t)j * w + i]) { in[(size_t)j * w + i] = 1; q[qt++ % QN] = j * w + i; } }
        }
    while (qh != qt) {
        int t = q[qh++ % QN]; in[t] = 0;
        int j = t / w, i = t % w; size_t c = (size_t)(j + 1) * P + i + 1; int vc = Wv[c];
        for (int k = 1; k <= 8; k++) {
            int ii = i + d1[k], jj = j + d2[k];
            if (ii < 0 || jj < 0 || ii >= w || jj >= h) continue;
            if (!(f->m[(size_t)(y0 + jj) * nx + x0 + ii] & (1u << (opp(k) - 1)))) continue;
            size_t n = (size_t)(jj + 1) * P + ii + 1;
            if (vc + inc < Wv[n]) { Wv[n] = vc + inc; changed = 1; if (!in[(size_t)jj * w + ii]) { in[(size_t)jj * w + ii] = 1; q[qt++ % QN] = jj * w + ii; } }
        }
    }
    if (changed) for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) out[(size_t)j * w + i] = Wv[(size_t)(j + 1) * P + i + 1];
    free(Wv); free(q); free(in);
    return changed;
}
// rounds until nothing is active or max_rounds (<= 0: no bound); returns the rounds run; prints per-round counts when verbose
static long run_rounds(Field* f, long max_rounds, int verbose, const char* tag) {
    const size_t nt = (size_t)f->tx * f->ty;
    long r = 0;
    int32_t* buf = NULL; size_t bufcap = 0;
    int* tiles = malloc(nt * sizeof(int));
    for (;;) {
        size_t na = 0;
        for (size_t t = 0; t < nt; t++) if (f->act[t]) tiles[na++] = (int)t;
        if (na == 0) break;
        if (max_rounds > 0 && r >= max_rounds) break;
        if (na * TS * TS > bufcap) { bufcap = na * TS * TS; buf = realloc(buf, bufcap * sizeof(int32_t)); }
        uint8_t* chg = calloc(na, 1);
        // macro blocks with an active tile: solved whole, on the same snapshot as the ordinary tiles (committed below)
        int nb_act = 0; int* bact = NULL; int32_t** bout = NULL; uint8_t* bchg = NULL;
        if (g_macro && f->inc == 1 && g_unit) {
            uint8_t* seen = calloc((size_t)g_nblocks + 1, 1);
            bact = malloc(sizeof(int) * (g_nblocks + 1));
            for (size_t a = 0; a < na; a++) { int u = g_unit[tiles[a]]; if (u >= 0 && !seen[u]) { seen[u] = 1; bact[nb_act++] = u; } }
            free(seen);
            bout = malloc(sizeof(int32_t*) * (nb_act + 1)); bchg = calloc((size_t)nb_act + 1, 1);
            for (int b = 0; b < nb_act; b++) {
                const Block B = g_blocks[bact[b]]; const int w = B.k * TS, h = B.k * TS;
                bout[b] = malloc((size_t)w * h * sizeof(int32_t));
                bchg[b] = (uint8_t)relax_region(f, B.tx0 * TS, B.ty0 * TS, w, h, bout[b]);
            }
        }
        for (size_t a = 0; a < na; a++) chg[a] = (g_macro && g_unit && f->inc == 1 && g_unit[tiles[a]] >= 0) ? 0 : (uint8_t)relax_tile(f, tiles[a] % f->tx, tiles[a] / f->tx, f->act[tiles[a]] >= 2, buf + a * TS * TS);
        memset(f->nxt, 0, nt);
        long nchg = 0;
        for (size_t a = 0; a < na; a++) {
            if (!chg[a]) continue;
            nchg++;
            const int tx = tiles[a] % f->tx, ty = tiles[a] / f->tx, x0 = tx * TS, y0 = ty * TS;
            const int32_t* o = buf + a * TS * TS;
            for (int j = 0; j < TS && y0 + j < f->ny; j++)
                for (int i = 0; i < TS && x0 + i < f->nx; i++) {
                    const size_t idx = (size_t)(y0 + j) * f->nx + x0 + i;
                    if (o[j * TS + i] == val(f->v[idx])) continue;
                    f->v[idx] = o[j * TS + i];
                    if (i == 0 || j == 0 || i == TS - 1 || j == TS - 1) {   // a rim cell moved: which cells of other tiles can it improve?
                        for (int k = 1; k <= 8; k++) {
                            int x = x0 + i + d1[k], y = y0 + j + d2[k];
                            if (x < 0 || y < 0 || x >= f->nx || y >= f->ny) continue;
                            if (x / TS == tx && y / TS == ty) continue;
                            const size_t n = (size_t)y * f->nx + x;
                            if (!(f->m[n] & (1u << (opp(k) - 1)))) continue;
                            // (judged against the neighbour's value before this round's write-backs are complete: an upper bound of it, so never a miss)
                            if (o[j * TS + i] + f->inc < val(f->v[n])) { uint8_t* q = &f->nxt[(size_t)(y / TS) * f->tx + x / TS]; if (!*q) *q = 1; }
                        }
                    }
                }
        }
        for (int b = 0; b < nb_act; b++) {   // commit the macro blocks; a changed RIM cell activates the tiles outside the block it can improve
            const Block B = g_blocks[bact[b]]; const int w = B.k * TS, h = B.k * TS, x0 = B.tx0 * TS, y0 = B.ty0 * TS;
            if (bchg[b]) {
                nchg++;
                for (int j = 0; j < h; j++)
                    for (int i = 0; i < w; i++) {
                        const size_t idx = (size_t)(y0 + j) * f->nx + x0 + i;
                        const int32_t nv = bout[b][(size_t)j * w + i];
                        if (nv == val(f->v[idx])) continue;
                        f->v[idx] = nv;
                        if (i == 0 || j == 0 || i == w - 1 || j == h - 1)
                            for (int k = 1; k <= 8; k++) {
                                int x = x0 + i + d1[k], y = y0 + j + d2[k];
                                if (x < 0 || y < 0 || x >= f->nx || y >= f->ny) continue;
                                if (x >= x0 && x < x0 + w && y >= y0 && y < y0 + h) continue;
                                const size_t n = (size_t)y * f->nx + x;
                                if ((f->m[n] & (1u << (opp(k) - 1))) && nv + f->inc < val(f->v[n])) f->nxt[(size_t)(y / TS) * f->tx + x / TS] = 1;
                            }
                    }
            }
            free(bout[b]);
        }
        free(bact); free(bout); free(bchg);
        if (g_chain) {   // plain tiles among the newly activated ones: processed now, in place, transitively
            int again = 1;
            int32_t* tb = malloc(sizeof(int32_t) * TS * TS);
            while (again) {
                again = 0;
                for (size_t t = 0; t < nt; t++) {
                    if (!f->nxt[t] || !f->plain[t]) continue;
                    f->nxt[t] = 0; again = 1; f->chained++;
                    const int tx = (int)(t % f->tx), ty = (int)(t / f->tx), x0 = tx * TS, y0 = ty * TS;
                    if (!relax_tile(f, tx, ty, 0, tb)) continue;
                    for (int j = 0; j < TS; j++)
                        for (int i = 0; i < TS; i++) {
                            const size_t idx = (size_t)(y0 + j) * f->nx + x0 + i;
                            if (tb[j * TS + i] == val(f->v[idx])) continue;
                            f->v[idx] = tb[j * TS + i];
                            if (i == 0 || j == 0 || i == TS - 1 || j == TS - 1)
                                for (int k = 1; k <= 8; k++) {
                                    int x = x0 + i + d1[k], y = y0 + j + d2[k];
                                    if (x < 0 || y < 0 || x >= f->nx || y >= f->ny || (x / TS == tx && y / TS == ty)) continue;
                                    const size_t n = (size_t)y * f->nx + x;
                                    if ((f->m[n] & (1u << (opp(k) - 1))) && tb[j * TS + i] + f->inc < val(f->v[n])) f->nxt[(size_t)(y / TS) * f->tx + x / TS] = 1;
                                }
                        }
                }
            }
            free(tb);
        }
        free(chg);
        if (verbose) fprintf(stderr, "%s round %ld: active %zu changed %ld\n", tag, f->rounds, na, nchg);
        f->activations += (long)na; f->changed_tiles += nchg; f->rounds++; r++;
        memcpy(f->act, f->nxt, nt);
    }
    free(buf); free(tiles);
    return r;
}

// ---- coarse level -----------------------------------------------------------------------------------------------------------------
typedef struct { Field f; int cf; uint8_t* full; } Coarse;
static void coarse_build(Coarse* c, const Field* fine, int cf) {
    const int nxc = (fine->nx + cf - 1) / cf, nyc = (fine->ny + cf - 1) / cf;
    int32_t* v = malloc((size_t)nxc * nyc * sizeof(int32_t));
    uint8_t* m = calloc((size_t)nxc * nyc, 1);
    c->cf = cf; c->full = calloc((size_t)nxc * nyc, 1);
    long nfull = 0;
    for (int by = 0; by < nyc; by++)
        for (int bx = 0; bx < nxc; bx++) {
            int ok = (bx + 1) * cf <= fine->nx && (by + 1) * cf <= fine->ny;
            for (int j = 0; j < cf && ok; j++)
                for (int i = 0; i < cf && ok; i++) {
                    const uint8_t mk = fine->m[(size_t)(by * cf + j) * fine->nx + bx * cf + i];
                    for (int k = 1; k <= 8; k++) {
                        int ii = i + d1[k], jj = j + d2[k];
                        if (ii < 0 || jj < 0 || ii >= cf || jj >= cf) continue;
                        if (!(mk & (1u << (k - 1)))) { ok = 0; break; }
                    }
                }
            c->full[(size_t)by * nxc + bx] = (uint8_t)ok; nfull += ok;
            v[(size_t)by * nxc + bx] = ok ? 0 : -1;
        }
    // links: B receives from its neighbour A (direction K) when A is full and the cells of B on the common edge hold the bit straight across (corner: the diagonal bit)
    for (int by = 0; by < nyc; by++)
        for (int bx = 0; bx < nxc; bx++) {
            if (!c->full[(size_t)by * nxc + bx]) continue;
            uint8_t cm = 0;
            for (int K = 1; K <= 8; K++) {
                int ax = bx + d1[K], ay = by + d2[K];
                if (ax < 0 || ay < 0 || ax >= nxc || ay >= nyc || !c->full[(size_t)ay * nxc + ax]) continue;
                int ok = 1;
                for (int j = 0; j < cf && ok; j++)
                    for (int i = 0; i < cf && ok; i++) {
                        // cells of B whose neighbour K lies in A
                        int ii = i + d1[K], jj = j + d2[K];
                        const int ox = ii < 0 ? -1 : (ii >= cf ? 1 : 0), oy = jj < 0 ? -1 : (jj >= cf ? 1 : 0);
                        const int inA = ox == d1[K] && oy == d2[K];
                        if (!inA) continue;
                        if (!(fine->m[(size_t)(by * cf + j) * fine->nx + bx * cf + i] & (1u << (K - 1)))) ok = 0;
                    }
                if (ok) cm |= (uint8_t)(1u << (K - 1));
            }
            m[(size_t)by * nxc + bx] = cm;
        }
    field_init(&c->f, nxc, nyc, cf, v, m);
    fprintf(stderr, "coarse level %d x %d: %ld full blocks (%.1f %% of the queue's cells)\n", nxc, nyc, nfull, 0.0);
}
// restriction; flags the coarse tiles that see a lowered block; returns the blocks lowered
static long restrict_level(Coarse* c, const Field* fine) {
    const int cf = c->cf, nxc = c->f.nx, nyc = c->f.ny;
    long lowered = 0;
    for (int by = 0; by < nyc; by++)
        for (int bx = 0; bx < nxc; bx++) {
            if (!c->full[(size_t)by * nxc + bx]) continue;
            int mn = INF, mx = 0;
            for (int j = 0; j < cf; j++)
                for (int i = 0; i < cf; i++) {
                    int w = val(fine->v[(size_t)(by * cf + j) * fine->nx + bx * cf + i]);
                    if (w < mn) mn = w;
                    if (w > mx) mx = w;
                }
            if (mn == INF) continue;
            int u = mn + (cf - 1) * fine->inc;
            if (mx < u) u = mx;
            int32_t* p = &c->f.v[(size_t)by * nxc + bx];
            if (u < val(*p)) { *p = u; lowered++; activate_around(&c->f, bx, by, 2); }
        }
    return lowered;
}
static long prolong_level(const Coarse* c, Field* fine) {
    const int cf = c->cf, nxc = c->f.nx, nyc = c->f.ny;
    long lowered = 0;
    for (int by = 0; by < nyc; by++)
        for (int bx = 0; bx < nxc; bx++) {
            if (!c->full[(size_t)by * nxc + bx]) continue;
            const int u = val(c->f.v[(size_t)by * nxc + bx]);
            if (u == INF) continue;
            for (int j = 0; j < cf; j++)
                for (int i = 0; i < cf; i++) {
                    const int x = bx * cf + i, y = by * cf + j;
                    int32_t* p = &fine->v[(size_t)y * fine->nx + x];
                    if (u < val(*p)) { *p = u; lowered++; activate_around(fine, x, y, 2); }
                }
        }
    return lowered;
}

static void* slurp(const char* fn, size_t bytes) {
    FILE* f = fopen(fn, "rb");
    if (!f) { perror(fn); exit(1); }
    void* p = malloc(bytes);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read\n", fn); exit(1); }
    fclose(f);
    return p;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static long cycle(Coarse* cs, int nlev, int lev, Field* fine, long* coarse_rounds, int verbose) {
    // restrict fine -> cs[lev], (recurse), relax, prolong
    long low = restrict_level(&cs[lev], fine);
    if (lev + 1 < nlev) cycle(cs, nlev, lev + 1, &cs[lev].f, coarse_rounds, verbose);
    long r = run_rounds(&cs[lev].f, 0, verbose > 1, lev ? "  coarse2" : " coarse");
    coarse_rounds[lev] += r;
    long pl = prolong_level(&cs[lev], fine);
    if (verbose) fprintf(stderr, "  level %d correction: %ld blocks restricted lower, %ld rounds, %ld cells lowered\n", lev + 1, low, r, pl);
    return pl;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: level_vcycle N lvl.i32 mask.u8 [after=8] [every=0] [max_cycles=1] [levels=1] [verbose=1]\n"); return 2; }
    const int N = atoi(argv[1]);
    const long after = argc > 4 ? atol(argv[4]) : 8, every = argc > 5 ? atol(argv[5]) : 0, max_cycles = argc > 6 ? atol(argv[6]) : 1;
    const int nlev = argc > 7 ? atoi(argv[7]) : 1, verbose = argc > 8 ? atoi(argv[8]) : 1;
    const size_t n = (size_t)N * N;
    int32_t* v0 = slurp(argv[2], n * 4);
    uint8_t* m = slurp(argv[3], n);
    double t0 = now();
    g_chain = getenv("SIM_CHAIN") != NULL;
    g_macro = getenv("SIM_MACRO") ? atoi(getenv("SIM_MACRO")) : 0;
    // ---- reference: no corrections
    int32_t* vr = malloc(n * 4); memcpy(vr, v0, n * 4);
    Field ref; field_init(&ref, N, N, 1, vr, m);
    for (size_t c = 0; c < n; c++) if (m[c] || v0[c] > 0) { size_t y = c / N, x = c % N; ref.act[(y / TS) * ref.tx + x / TS] = 2; }
    if (g_macro) build_blocks(&ref);
    run_rounds(&ref, 0, verbose > 2, "plain");
    int mxl = 0; long unreached = 0;
    for (size_t c = 0; c < n; c++) { if (vr[c] > mxl) mxl = vr[c]; unreached += vr[c] == 0; }
    { long np = 0; for (size_t t = 0; t < (size_t)ref.tx * ref.ty; t++) np += ref.plain[t];
      printf("plain%s: %ld rounds, %ld activations (%ld changed, %ld chained in-round on %ld full / plain tiles), deepest level %d, unreached %ld  [%.1f s]\n", g_chain ? " + chain" : "", ref.rounds, ref.activations, ref.changed_tiles, ref.chained, np, mxl, unreached, now() - t0); }
    if (max_cycles <= 0) return 0;
    // ---- with corrections
    int32_t* v = malloc(n * 4); memcpy(v, v0, n * 4);
    Field f; field_init(&f, N, N, 1, v, m);
    memcpy(f.act, ref.act, 0);
    for (size_t c = 0; c < n; c++) if (m[c] || v0[c] > 0) { size_t y = c / N, x = c % N; f.act[(y / TS) * f.tx + x / TS] = 2; }
    Coarse cs[4];
    for (int l = 0; l < nlev; l++) coarse_build(&cs[l], l ? &cs[l - 1].f : &f, 8);
    long coarse_rounds[4] = {0, 0, 0, 0}, cycles = 0;
    run_rounds(&f, after, verbose > 2, "fine");
    for (;;) {
        size_t na = 0;
        for (size_t t = 0; t < (size_t)f.tx * f.ty; t++) na += f.act[t] != 0;
        if (na == 0 || cycles >= max_cycles) break;
        if (verbose) fprintf(stderr, "after %ld fine rounds (%zu tiles active): correction %ld\n", f.rounds, na, cycles + 1);
        cycle(cs, nlev, 0, &f, coarse_rounds, verbose);
        cycles++;
        run_rounds(&f, every > 0 && cycles < max_cycles ? every : 0, verbose > 2, "fine");
    }
    run_rounds(&f, 0, verbose > 2, "fine");
    long diff = 0;
    for (size_t c = 0; c < n; c++) diff += v[c] != vr[c];
    printf("corrected (after %ld, every %ld, <= %ld cycles, %d coarse levels): %ld fine rounds, %ld activations (%ld changed); %ld cycles, coarse rounds %ld / %ld / %ld, coarse activations %ld; differs from plain in %ld cells  [%.1f s]\n",
           after, every, max_cycles, nlev, f.rounds, f.activations, f.changed_tiles, cycles, coarse_rounds[0], coarse_rounds[1], coarse_rounds[2], cs[0].f.activations, diff, now() - t0);
    return diff != 0;
}
