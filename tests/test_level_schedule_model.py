"""The CPU model of the level fields' round schedule (scripts/sim/level_vcycle.c; docs/experiments_r06.md section 2) on a small raster of the restatement's first flat
iteration: it must build; the coarse corrections, the chaining through full tiles and the macro blocks must all reach the plain schedule's field cell by cell (the model
checks that itself and says so); and the corrected schedule must not need fewer rounds than the plain one by more than the model found at scale - i.e. the finding the
round was built on (an additive bound does not shorten a min-plus relaxation) is reproducible from the repo."""
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nb(a, dy, dx, fill):
    out = np.full_like(a, fill)
    H, W = a.shape
    y0, y1 = max(0, -dy), H - max(0, dy)
    x0, x1 = max(0, -dx), W - max(0, dx)
    out[y0:y1, x0:x1] = a[y0 + dy:y1 + dy, x0 + dx:x1 + dx]
    return out


def test_corrections_do_not_shorten_a_level_relaxation(tmp_path, oracle):
    n = 1024
    dem = oracle.synth_dem((n, n), 1234)
    fel = oracle.pitremove(dem, -9999.0)
    p, sd8, _ = oracle.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    flat = sd8 == 0
    flat[0, :] = flat[-1, :] = False
    flat[:, 0] = flat[:, -1] = False
    hasdir = (~flat) & (p >= 1) & (p <= 8)
    d1 = [0, 1, 1, 0, -1, -1, -1, 0, 1]
    d2 = [0, 0, -1, -1, -1, 0, 1, 1, 1]
    low = np.zeros_like(flat)
    quirk = np.zeros_like(flat)
    fm = np.zeros(flat.shape, np.uint8)
    for k in range(1, 9):   # the seeds and masks of incfall as flatk::classify_kernel makes them (first iteration: dontCross cannot fire)
        zn, hn, qn = _nb(fel, d2[k], d1[k], np.float32(0)), _nb(hasdir, d2[k], d1[k], False), _nb(flat, d2[k], d1[k], False)
        zd = fel - zn
        lo = flat & (zd >= 0) & hn
        low |= lo
        eq = flat & ~lo & (zd == 0)
        fm |= (eq & qn).astype(np.uint8) << (k - 1)
        quirk |= eq & ~qn
    lvl = np.where(flat, np.where(low, 1, np.where(quirk, 2, 0)), -1).astype(np.int32)
    fm[low] = 0
    fm[~flat] = 0
    lvl.tofile(tmp_path / "lvl.i32")
    fm.tofile(tmp_path / "fm.u8")
    exe = tmp_path / "level_vcycle"
    subprocess.run(["gcc", "-O2", "-w", "-o", str(exe), os.path.join(ROOT, "scripts", "sim", "level_vcycle.c")], check=True)

    def run(env, *args):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([str(exe), str(n), str(tmp_path / "lvl.i32"), str(tmp_path / "fm.u8"), *args], capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stdout + r.stderr      # (non-zero = a variant's field differs from the plain one)
        return r.stdout

    out = run({}, "4", "4", "20", "1", "0")
    plain = int(re.search(r"plain: (\d+) rounds", out)[1])
    m = re.search(r"corrected .*: (\d+) fine rounds, .* (\d+) cycles, .* differs from plain in (\d+) cells", out)
    assert m, out
    corrected, cycles, diff = int(m[1]), int(m[2]), int(m[3])
    assert diff == 0 and cycles >= 1
    assert corrected >= 0.7 * plain, (plain, corrected)      # 168 -> 151 at 16384^2, 85 -> 73 at 8192^2: never the factor a correction would have to pay for itself
    chain = int(re.search(r"plain \+ chain: (\d+) rounds", run({"SIM_CHAIN": "1"}, "8", "0", "0", "1", "0"))[1])
    macro = int(re.search(r"plain: (\d+) rounds", run({"SIM_MACRO": "8"}, "8", "0", "0", "1", "0"))[1])
    assert chain <= macro <= plain, (chain, macro, plain)
