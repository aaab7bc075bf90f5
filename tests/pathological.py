"""Inputs that are nothing like a fractal surface (VERDICT r04 missing #5): what the reference computes on them is the contract
(its comparisons: src/flood.cpp:307-330, src/linearpart.h:470-483), the restatement is pinned to the reference, and the tile
schedules must reach the same bits whatever the dependency structure looks like."""
import numpy as np

NODATA = -9999.0


def plane(ny=300, nx=400, z=7.0):
    return np.full((ny, nx), z, np.float32)


def ramp(ny=257, nx=300, ax=1.0, ay=0.0):
    y, x = np.mgrid[0:ny, 0:nx]
    return (ax * x + ay * y).astype(np.float32)


def checkerboard_pits(ny=256, nx=320):
    """every second cell of every second row is a one-cell pit in a gently tilted plane"""
    y, x = np.mgrid[0:ny, 0:nx]
    z = (100.0 + 0.01 * x + 0.02 * y).astype(np.float32)
    z[(y % 2 == 1) & (x % 2 == 1)] -= np.float32(5.0)
    return z


def spiral(n=512, pitch=8):
    """An Archimedean-style square spiral channel: a wall plateau at 1000 with a one-cell channel that winds from the centre to the
    raster's edge, descending by one unit per cell - the longest flow path is ~ n^2 / pitch cells, its dependency depth in tiles ~ n / pitch
    turns x the tiles of a turn.  pitch = distance between two windings."""
    z = np.full((n, n), 1000.0, np.float32)
    cy = cx = n // 2
    y, x = cy, cx
    path = [(y, x)]
    step, d = pitch, 0
    dirs = [(0, 1), (1, 0), (0, -1), (-1, 0)]
    done = False
    while not done:
        for _ in range(2):
            dy, dx = dirs[d % 4]
            for _ in range(step):
                y += dy; x += dx
                if y < 0 or y >= n or x < 0 or x >= n:
                    done = True
                    break
                path.append((y, x))
            d += 1
            if done:
                break
        step += pitch
    L = len(path)
    for i, (yy, xx) in enumerate(path):
        z[yy, xx] = np.float32(900.0 - 0.5 * i * (800.0 / (0.5 * L + 1)) / 800.0 * 1.0) if False else np.float32(900.0 - i * (800.0 / L))
    return z


def one_row(n=3000):
    return (np.sin(np.arange(n, dtype=np.float64) * 0.01) * 50.0).astype(np.float32).reshape(1, n)


def one_column(n=3000):
    return one_row(n).reshape(n, 1).copy()


def all_nodata(ny=130, nx=200):
    return np.full((ny, nx), NODATA, np.float32)


def one_data_cell(ny=130, nx=200):
    z = all_nodata(ny, nx)
    z[ny // 2, nx // 2] = 5.0
    return z


def with_specials(base, kind, seed=5, count=40):
    """`base` with `count` interior cells replaced by NaN / +Inf / -Inf"""
    z = base.copy()
    rng = np.random.default_rng(seed)
    ys = rng.integers(2, z.shape[0] - 2, count)
    xs = rng.integers(2, z.shape[1] - 2, count)
    z[ys, xs] = {"nan": np.float32(np.nan), "+inf": np.float32(np.inf), "-inf": np.float32(-np.inf)}[kind]
    return z


def fractal(oracle, ny=300, nx=333, seed=11):
    return oracle.synth_dem((ny, nx), seed)


CASES = {
    "plane": lambda o: plane(),
    "ramp_x": lambda o: ramp(ax=1.0, ay=0.0),
    "ramp_-x": lambda o: ramp(ax=-1.0, ay=0.0),
    "ramp_y": lambda o: ramp(ax=0.0, ay=1.0),
    "ramp_-y": lambda o: ramp(ax=0.0, ay=-1.0),
    "ramp_diag": lambda o: ramp(ax=1.0, ay=1.0),
    "ramp_antidiag": lambda o: ramp(ax=1.0, ay=-1.0),
    "ramp_shallow": lambda o: ramp(ax=1e-3, ay=0.0),      # steps below float32 resolution of the slope: runs of equal cells
    "checkerboard_pits": lambda o: checkerboard_pits(),
    "spiral": lambda o: spiral(384, 8),
    "one_row": lambda o: one_row(),
    "one_column": lambda o: one_column(),
    "two_rows": lambda o: np.vstack([one_row(), one_row()[:, ::-1]]),
    "all_nodata": lambda o: all_nodata(),
    "one_data_cell": lambda o: one_data_cell(),
    "nan_cells": lambda o: with_specials(fractal(o), "nan"),
    "+inf_cells": lambda o: with_specials(fractal(o), "+inf"),
    "-inf_cells": lambda o: with_specials(fractal(o), "-inf"),
}
