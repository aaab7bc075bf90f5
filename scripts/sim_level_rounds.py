"""CPU model of the tile engine's round schedule for the incfall level field of flat resolution (the companion of sim_tile_rounds.py):
seeds and masks as flatk::classify_kernel makes them (numpy, first iteration), 64x64 tiles, every active tile relaxed to its local
fixed point per round against the previous round's halo, neighbours of a tile whose rim changed are active in the next round.

    python scripts/sim_level_rounds.py [n=2048] [filter|chain]

`filter`: activate a neighbour only if a changed rim cell (new level v) touches a cell x of it that is in the queue and holds more than
v + 1 in the halo AS LOADED (an upper bound of its current level: the test never misses an improvement; a cell outside the queue
reads -1 and never moves).
`chain` (round 4, the review's "carry the front rim-to-rim across plain tiles"): a tile all of whose 4096 cells are in the queue and free to move
(no seed, no cell outside the flat: in such a tile the level field is the chessboard distance transform of its ring) is processed IN THE ROUND
that activates it, transitively - the best case of a persistent workgroup that follows the front through such tiles - and the rounds / launches
that are left are counted.  Analysis tool (uses oracle/); the result is checked against the global fixed point."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle as O
O.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
FILTER = len(sys.argv) > 2 and sys.argv[2] == 'filter'
CHAIN = len(sys.argv) > 2 and sys.argv[2] == 'chain'
import os
TS = int(os.environ.get("SIM_TS", "64"))   # SIM_TS=128: what a coarser tile geometry would make of the round count
INF = 0x3fffffff
def nb(a, dy, dx, fill):
    out = np.full_like(a, fill); H, W = a.shape
    y0, y1 = max(0, -dy), H - max(0, dy); x0, x1 = max(0, -dx), W - max(0, dx)
    out[y0:y1, x0:x1] = a[y0 + dy:y1 + dy, x0 + dx:x1 + dx]
    return out
dem = O.synth_dem(N, 1234)
fel = O.pitremove(dem)
p, sd8 = O.d8flowdir(fel)[:2]
flat = (sd8 == 0); flat[0, :] = flat[-1, :] = False; flat[:, 0] = flat[:, -1] = False
p0 = np.where(flat, 0, p).astype(np.int16)
d1 = [0, 1, 1, 0, -1, -1, -1, 0, 1]; d2 = [0, 0, -1, -1, -1, 0, 1, 1, 1]
low = np.zeros_like(flat); quirk = np.zeros_like(flat); anyq = np.zeros_like(flat)
for k in range(1, 9):   # (dontCross cannot shut out an in-queue neighbour in the first iteration: DESIGN.md 4.2)
    zn = nb(fel, d2[k], d1[k], np.float32(0)); pn = nb(p0, d2[k], d1[k], -32768)
    zd = fel - zn; inq = pn == 0
    lo = flat & (zd >= 0) & (pn > 0) & (pn < 9)
    low |= lo
    eq = flat & ~lo & (zd == 0)
    anyq |= eq & inq
    quirk |= eq & ~inq
movable = flat & ~low & anyq
val = np.where(flat, np.where(low, 1, np.where(quirk, 2, INF)), INF).astype(np.int64)   # not in the queue: +inf for good
inq_all = flat
def fixpoint(v, mov):
    it = 0
    while True:
        m = np.minimum.reduce([nb(v, d2[k], d1[k], INF) for k in range(1, 9)])
        wn = np.where(mov, np.minimum(v, m + 1), v)
        it += 1
        if (wn == v).all(): return v, it
        v = wn
t0 = time.time()
import os
REF_FILE = os.environ.get("SIM_REF")     # SIM_REF=file.npy: compare with (or, if absent, write) a saved result instead of the slow global iteration
if REF_FILE and os.path.exists(REF_FILE): ref = np.load(REF_FILE); print(f"reference field from {REF_FILE}, {int(flat.sum())} flat cells", flush=True)
elif REF_FILE: ref = None; print(f"no reference yet: this run's field goes to {REF_FILE}, {int(flat.sum())} flat cells", flush=True)
else:
    ref, levels = fixpoint(val.copy(), movable)
    print(f"global fixed point: {levels} levels, {int(flat.sum())} flat cells  [{time.time()-t0:.1f}s]", flush=True)
nt = N // TS
Vp = np.full((N + 2, N + 2), INF, np.int64); Vp[1:-1, 1:-1] = val
Qp = np.zeros((N + 2, N + 2), bool); Qp[1:-1, 1:-1] = inq_all
def relax_tile(win, mov):
    w = win.copy()
    while True:
        c = w[1:-1, 1:-1]
        m = np.minimum.reduce([w[0:-2, 0:-2], w[0:-2, 1:-1], w[0:-2, 2:], w[1:-1, 0:-2], w[1:-1, 2:], w[2:, 0:-2], w[2:, 1:-1], w[2:, 2:]])
        new = np.where(mov, np.minimum(c, m + 1), c)
        if np.array_equal(new, c): return c
        w[1:-1, 1:-1] = new
active = flat.reshape(nt, TS, nt, TS).any(axis=(1, 3))
fps = movable.reshape(nt, TS, nt, TS).all(axis=(1, 3))      # full, plain, seedless tiles: every cell in the queue and free to move
print(f"tiles {nt * nt}, with flat cells {int(active.sum())}, full / plain / seedless {int(fps.sum())}", flush=True)
rnd = 0; tot = 0; tot_fps = 0; tail_act = 0; tail_fps = 0; chained = 0
while active.any():
    cur = Vp.copy()
    nxt = np.zeros((nt, nt), bool)
    nact = int(active.sum()); nchg = 0
    nfps = int((active & fps).sum()); tot_fps += nfps
    if nact <= 64: tail_act += nact; tail_fps += nfps
    work = list(zip(*np.nonzero(active)))
    wi = 0
    while wi < len(work):
        ty, tx = work[wi]; wi += 1
        if CHAIN and wi > nact: cur = Vp       # a chained tile sees what this round has written so far (same workgroup)
        y0, x0 = ty * TS, tx * TS
        win = cur[y0:y0 + TS + 2, x0:x0 + TS + 2]
        old = win[1:-1, 1:-1].copy()
        new = relax_tile(win, movable[y0:y0 + TS, x0:x0 + TS])
        if np.array_equal(new, old): continue
        nchg += 1
        Vp[y0 + 1:y0 + TS + 1, x0 + 1:x0 + TS + 1] = new
        d = new != old
        if FILTER:
            H = win; Q = Qp[y0:y0 + TS + 2, x0:x0 + TS + 2]
            def can_move(vals, chg, h, q):   # rim line (TS) against the halo line (TS + 2) beside it
                best = np.full(TS + 2, INF, np.int64)
                vv = np.where(chg, vals, INF)
                best[0:-2] = np.minimum(best[0:-2], vv); best[1:-1] = np.minimum(best[1:-1], vv); best[2:] = np.minimum(best[2:], vv)
                return bool(np.any(q & (h > best + 1)))
            tests = ((-1, 0, can_move(new[0, :], d[0, :], H[0, :], Q[0, :])), (1, 0, can_move(new[-1, :], d[-1, :], H[-1, :], Q[-1, :])),
                     (0, -1, can_move(new[:, 0], d[:, 0], H[:, 0], Q[:, 0])), (0, 1, can_move(new[:, -1], d[:, -1], H[:, -1], Q[:, -1])),
                     (-1, -1, bool(d[0, 0] and Q[0, 0] and H[0, 0] > new[0, 0] + 1)), (-1, 1, bool(d[0, -1] and Q[0, -1] and H[0, -1] > new[0, -1] + 1)),
                     (1, -1, bool(d[-1, 0] and Q[-1, 0] and H[-1, 0] > new[-1, 0] + 1)), (1, 1, bool(d[-1, -1] and Q[-1, -1] and H[-1, -1] > new[-1, -1] + 1)))
        else:
            tests = ((-1, 0, np.any(d[0, :])), (1, 0, np.any(d[-1, :])), (0, -1, np.any(d[:, 0])), (0, 1, np.any(d[:, -1])), (-1, -1, d[0, 0]), (-1, 1, d[0, -1]),
                     (1, -1, d[-1, 0]), (1, 1, d[-1, -1]))
        for dy, dx, hit in tests:
            if hit:
                yy, xx = ty + dy, tx + dx
                if 0 <= yy < nt and 0 <= xx < nt:
                    if CHAIN and fps[yy, xx] and nact <= 64:
                        work.append((yy, xx)); chained += 1          # processed in this same round (it may come up again later in the round)
                    else: nxt[yy, xx] = True
    tot += nact
    if rnd < 12 or nact > 50 or rnd % 10 == 0: print(f"round {rnd:3d}: active {nact:5d} (full/plain/seedless {nfps:5d}) changed {nchg:5d} ({100.0*nchg/nact:5.1f}%) processed {len(work):5d} [{time.time()-t0:6.1f}s]", flush=True)
    active = nxt; rnd += 1
    if rnd > 2000: break
if ref is None: np.save(REF_FILE, Vp[1:-1, 1:-1]); ref = Vp[1:-1, 1:-1]
print("rounds", rnd, "total activations", tot, "of them on full/plain/seedless tiles", tot_fps, "| rounds with <= 64 active tiles: activations", tail_act, "on full/plain/seedless tiles", tail_fps,
      "| chained in-round activations", chained, "| tiles", nt * nt, "equals the global fixed point", bool(np.array_equal(Vp[1:-1, 1:-1], ref)))
