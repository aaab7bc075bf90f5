"""CPU model of the tile engine's round schedule (taudem_amd/csrc/tile_relax.hpp) for PitRemove: 64x64 tiles, coarse-to-fine
start, every active tile relaxed to its local fixed point per round (Jacobi across tiles: a tile reads the previous round's halo),
neighbours of a tile whose rim changed are active in the next round.  Prints per round: active tiles, tiles that changed at
all, tiles whose rim changed; checks the result bit-for-bit against the oracle.  Analysis tool (uses oracle/): schedule ideas can
be counted here before they are written in HIP.

    python scripts/sim_tile_rounds.py [n=2048] [filter]

`filter`: activate a neighbour only if a changed rim cell is lower than a cell of that neighbour it touches (as loaded).
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
        if not np.array_equal(new, old):
            nchg += 1
            Wp[y0 + 1:y0 + TS + 1, x0 + 1:x0 + TS + 1] = new
            d = new != old
            rim = False
            if FILTER:
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
                        nxt[yy, xx] = True; rim = True
            nrim += rim
    tot_act += nact
    print(f"round {rnd:3d}: active {nact:5d} changed {nchg:5d} ({100.0*nchg/nact:5.1f}%) rim-changed {nrim:5d} mean jacobi iters {its/nact:6.1f}  [{time.time()-t0:6.1f}s]", flush=True)
    active = nxt
    rnd += 1
    if rnd > 400: break
print("total activations", tot_act, "tiles", nt * nt, "bit-exact", np.array_equal(Wp[1:-1, 1:-1].view(np.uint32), ref.view(np.uint32)))
