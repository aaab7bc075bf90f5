"""CPU model of the tile engine's round schedule (taudem_amd/csrc/tile_relax.hpp) for PitRemove: 64x64 tiles, coarse-to-fine
start, every active tile relaxed to its local fixed point per round (Jacobi across tiles: a tile reads the previous round's halo),
neighbours of a tile whose rim changed are active in the next round.  Prints per round: active tiles, tiles that changed at
all, tiles whose rim changed; checks the result bit-for-bit against the oracle.  Analysis tool (uses oracle/): schedule ideas can
be counted here before they are written in HIP.

    python scripts/sim_tile_rounds.py [n=2048] [filter]

`filter`: activate a neighbour only if a changed rim cell is lower than a cell of that neighbour it touches (as loaded).
`fresh`:  the same test against the neighbour's CURRENT values and elevations (v < W(x) and Z(x) < W(x) for a touched cell x):
          1001 / 990 / 630 active tiles in rounds 1-3 become 993 / 796 / 295, 4489 activations in total become 3718.
`live`:   activate a neighbour only if it had an unsettled cell on the facing rim when it last ran: 989 / 615, i.e. nothing.
Findings (2048^2): 1024, 1001, 990, 630, 240, 134, ... active tiles per round - the GPU's own counts for the 2048^2 level
are 1024, 1002, 990, 700, 299, 160 -; from round 2 on only 30-55 % of the activated tiles change anything (their neighbour's
rim moved, but the cells that see it are already settled), and the filter does not catch those (990 -> 990, 630 -> 609):
the settled state of the neighbour's cells is not visible from the activating tile."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle as O
O.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
TS = int(sys.argv[3]) if len(sys.argv) > 3 else 64
FILTER = len(sys.argv) > 2 and sys.argv[2] == 'filter'
FRESH = len(sys.argv) > 2 and sys.argv[2] == 'fresh'   # like `filter`, but against the neighbour's CURRENT rim values and elevations: v < W(x) and Z(x) < W(x) for a cell x it touches
LIVE = len(sys.argv) > 2 and sys.argv[2] == 'live'   # activate a neighbour only if, at ITS last activation (or at the start), it had an unsettled cell on the rim that faces this tile
Z = O.synth_dem(N, 1234).astype(np.float32)
INF = np.float32(3.0e38)
# coarse start: 8x8 max pooling, exact coarse solve, prolongation (one level)
Zc = Z.reshape(N // 8, 8, N // 8, 8).max(axis=(1, 3))
Wc = O.pitremove(Zc, -9999.0)
W = np.repeat(np.repeat(Wc, 8, axis=0), 8, axis=1).astype(np.float32)
W = np.maximum(W, Z)
W[0, :] = Z[0, :]; W[-1, :] = Z[-1, :]; W[:, 0] = Z[:, 0]; W[:, -1] = Z[:, -1]
ref = O.pitremove(Z, -9999.0)
assert (W >= ref).all()
nt = N // TS
def relax_tile(win, zwin):
    """win: (TS+2, TS+2) values incl. halo; zwin: (TS, TS) elevations.  Returns new interior."""
    w = win.copy()
    it = 0
    while True:
        it += 1
        c = w[1:-1, 1:-1]
        m = np.minimum.reduce([w[0:-2, 0:-2], w[0:-2, 1:-1], w[0:-2, 2:], w[1:-1, 0:-2], w[1:-1, 2:], w[2:, 0:-2], w[2:, 1:-1], w[2:, 2:]])
        new = np.where(c > zwin, np.maximum(zwin, np.minimum(c, m)), c)
        if np.array_equal(new, c):
            return c, it
        w[1:-1, 1:-1] = new
Wp = np.full((N + 2, N + 2), INF, np.float32)
Wp[1:-1, 1:-1] = W
fixed = np.zeros((N, N), bool); fixed[0, :] = fixed[-1, :] = fixed[:, 0] = fixed[:, -1] = True
Zeff = np.where(fixed, INF, Z)  # edge cells never change: treat z = +inf so that c > z is false
active = np.ones((nt, nt), bool)
rim_live = np.ones((nt, nt, 8), bool)   # top, bottom, left, right, then the corners NW NE SW SE; all "live" until a tile has run once
LIVE_BIT = {(-1, 0): 0, (1, 0): 1, (0, -1): 2, (0, 1): 3, (-1, -1): 4, (-1, 1): 5, (1, -1): 6, (1, 1): 7}   # side of the NEIGHBOUR that faces us, by (dy, dx) seen from it
skipped = 0
rnd = 0
t0 = time.time()
tot_act = 0
while active.any():
    cur = Wp.copy()
    nxt = np.zeros((nt, nt), bool)
    nact = int(active.sum()); nchg = 0; nrim = 0; its = 0
    for ty, tx in zip(*np.nonzero(active)):
        y0, x0 = ty * TS, tx * TS
        win = cur[y0:y0 + TS + 2, x0:x0 + TS + 2]
        old = win[1:-1, 1:-1]
        new, it = relax_tile(win, Zeff[y0:y0 + TS, x0:x0 + TS])
        its += it
        if LIVE:   # what the tile publishes at the end of its activation: unsettled cells per side / corner of its rim
            lv = new > Zeff[y0:y0 + TS, x0:x0 + TS]
            rim_live[ty, tx] = [lv[0, :].any(), lv[-1, :].any(), lv[:, 0].any(), lv[:, -1].any(), lv[0, 0], lv[0, -1], lv[-1, 0], lv[-1, -1]]
        if not np.array_equal(new, old):
            nchg += 1
            Wp[y0 + 1:y0 + TS + 1, x0 + 1:x0 + TS + 1] = new
            d = new != old
            rim = False
            if FRESH:
                Hw = Wp[y0:y0 + TS + 2, x0:x0 + TS + 2]            # current values around the tile (neighbours may have moved since the load)
                Hz = np.full((TS + 2, TS + 2), INF, np.float32)      # elevations around the tile (outside the raster: never movable)
                ya, yb, xa, xb = max(y0 - 1, 0), min(y0 + TS + 1, N), max(x0 - 1, 0), min(x0 + TS + 1, N)
                Hz[ya - (y0 - 1):yb - (y0 - 1), xa - (x0 - 1):xb - (x0 - 1)] = Zeff[ya:yb, xa:xb]
                def can_move(vals, chg, hw, hz):   # rim line (TS) against the halo line (TS+2) beside it: some touched cell x with v < W(x) and Z(x) < W(x)
                    mov = hz < hw
                    best = np.full(TS + 2, INF, np.float32)          # lowest changed rim value each halo cell touches
                    vv = np.where(chg, vals, INF)
                    best[0:-2] = np.minimum(best[0:-2], vv); best[1:-1] = np.minimum(best[1:-1], vv); best[2:] = np.minimum(best[2:], vv)
                    return bool(np.any(mov & (best < hw)))
                tests = ((-1, 0, can_move(new[0, :], d[0, :], Hw[0, :], Hz[0, :])), (1, 0, can_move(new[-1, :], d[-1, :], Hw[-1, :], Hz[-1, :])),
                         (0, -1, can_move(new[:, 0], d[:, 0], Hw[:, 0], Hz[:, 0])), (0, 1, can_move(new[:, -1], d[:, -1], Hw[:, -1], Hz[:, -1])),
                         (-1, -1, bool(d[0, 0] and Hz[0, 0] < Hw[0, 0] and new[0, 0] < Hw[0, 0])), (-1, 1, bool(d[0, -1] and Hz[0, -1] < Hw[0, -1] and new[0, -1] < Hw[0, -1])),
                         (1, -1, bool(d[-1, 0] and Hz[-1, 0] < Hw[-1, 0] and new[-1, 0] < Hw[-1, 0])), (1, 1, bool(d[-1, -1] and Hz[-1, -1] < Hw[-1, -1] and new[-1, -1] < Hw[-1, -1])))
            elif FILTER:
                # flag a neighbour only if a changed rim cell is now LOWER than one of the (up to 3) cells of that neighbour it touches,
                # as this tile saw them when it loaded its halo (an upper bound of their current values: never misses an improvement)
                H = win  # (TS+2, TS+2) as loaded
                def lower_than_halo(vals, chg, h):   # vals/chg: rim line (TS), h: halo line (TS+2) beside it
                    hm = np.maximum(np.maximum(h[0:-2], h[1:-1]), h[2:])
                    return bool(np.any(chg & (vals < hm)))
                tests = ((-1, 0, lower_than_halo(new[0, :], d[0, :], H[0, :])), (1, 0, lower_than_halo(new[-1, :], d[-1, :], H[-1, :])),
                         (0, -1, lower_than_halo(new[:, 0], d[:, 0], H[:, 0])), (0, 1, lower_than_halo(new[:, -1], d[:, -1], H[:, -1])),
                         (-1, -1, bool(d[0, 0] and new[0, 0] < H[0, 0])), (-1, 1, bool(d[0, -1] and new[0, -1] < H[0, -1])),
                         (1, -1, bool(d[-1, 0] and new[-1, 0] < H[-1, 0])), (1, 1, bool(d[-1, -1] and new[-1, -1] < H[-1, -1])))
            else:
                tests = ((-1, 0, np.any(d[0, :])), (1, 0, np.any(d[-1, :])), (0, -1, np.any(d[:, 0])), (0, 1, np.any(d[:, -1])), (-1, -1, d[0, 0]), (-1, 1, d[0, -1]),
                         (1, -1, d[-1, 0]), (1, 1, d[-1, -1]))
            for dy, dx, hit in tests:
                if hit:
                    yy, xx = ty + dy, tx + dx
                    if 0 <= yy < nt and 0 <= xx < nt:
                        if LIVE and not rim_live[yy, xx, LIVE_BIT[(-dy, -dx)]]:
                            skipped += 1
                            continue
                        nxt[yy, xx] = True; rim = True
            nrim += rim
    tot_act += nact
    print(f"round {rnd:3d}: active {nact:5d} changed {nchg:5d} ({100.0*nchg/nact:5.1f}%) rim-changed {nrim:5d} mean jacobi iters {its/nact:6.1f}  [{time.time()-t0:6.1f}s]", flush=True)
    active = nxt
    rnd += 1
    if rnd > 400: break
if LIVE: print("activations skipped by the liveness test", skipped)
print("total activations", tot_act, "tiles", nt * nt, "bit-exact", np.array_equal(Wp[1:-1, 1:-1].view(np.uint32), ref.view(np.uint32)))
