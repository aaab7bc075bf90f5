"""The step model of ad8_tile_local_kernel's in-tile Kahn schedules (scripts/sim/kahn_steps.c) on a small raster of the restatement's directions: it must build, every
schedule must make the same number of hops (one per in-tile flow: the schedule is free, src/aread8.cpp:220-304), and one loop per lane must not need more wave steps than
the loop nest it replaced (docs/experiments_r05.md section 4)."""
import os
import re
import subprocess

import numpy as np


def test_flat_schedule_needs_fewer_steps(tmp_path, oracle):
    n = 256
    dem = oracle.synth_dem((n, n), 1234)
    fel = oracle.pitremove(dem, -9999.0)
    p, _, _ = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    raw = tmp_path / "p.bin"
    np.ascontiguousarray(p, dtype=np.int16).tofile(raw)
    exe = tmp_path / "kahn_steps"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "sim", "kahn_steps.c")
    subprocess.run(["gcc", "-O2", "-w", "-o", str(exe), src], check=True)
    out = subprocess.run([str(exe), str(n), str(raw)], check=True, capture_output=True, text=True).stdout
    m = re.search(r"queue ([\d.]+); tiles (\d+): mean steps per tile nested ([\d.]+), flat ([\d.]+), hops per tile ([\d.]+), longest path ([\d.]+)", out)
    assert m, out
    queue, tiles, nested, flat, hops, longest = float(m[1]), int(m[2]), float(m[3]), float(m[4]), float(m[5]), float(m[6])
    assert tiles == (n // 64) ** 2
    # hops per tile = in-tile flows: cells whose target lies in the same tile
    d1 = np.array([0, 1, 1, 0, -1, -1, -1, 0, 1]); d2 = np.array([0, 0, -1, -1, -1, 0, 1, 1, 1])
    yy, xx = np.mgrid[0:n, 0:n]
    valid = (p >= 1) & (p <= 8)
    pc = np.where(valid, p, 0)
    ty, tx = yy + d2[pc], xx + d1[pc]
    intile = valid & (ty // 64 == yy // 64) & (tx // 64 == xx // 64) & (ty >= 0) & (ty < n) & (tx >= 0) & (tx < n)
    assert abs(hops * tiles - int(intile.sum())) < 0.5 * tiles
    assert longest <= flat <= nested and queue <= flat
