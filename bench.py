#!/usr/bin/env python
"""Benchmark of the PitRemove -> D8FlowDir -> AreaD8 pipeline (BASELINE.json metric: Mcells/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size n | --nx X --ny Y]

One "step" = one pass of the three-stage pipeline over one synthetic fractal DEM that is already
resident in HBM (generated on the device by tdx_synth_dem_dev).
  N = 1   BASELINE.json configs[1]: 16384 x 16384 on one MI355X.
  N > 1   ONE raster of 65536 columns x 8192*N rows, row-partitioned like linearpart (src/linearpart.h:133-134) into
          65536 x 8192 strips, one per GPU - at N = 8 this is BASELINE.json configs[3], the 65536 x 65536 pipeline (weak
          scaling over N = 2, 4, 8: fixed strip per GPU).  One process per GPU: when the driver has not launched the ranks
          itself (RANK unset) `python bench.py --gpus N` re-executes itself under torch.distributed.run.  Halo rows and
          cross-strip dependencies travel through the library's native RCCL transport (taudem_amd/csrc/comm.cpp: grouped
          ncclSend/ncclRecv + ncclAllReduce on the compute stream); torch.distributed only bootstraps the ncclUniqueId and
          carries the timing barrier.  TDX_BENCH_BACKEND=gloo: host-staged Python transport (several ranks may share a GPU;
          functional check only).
--nx/--ny set the TOTAL raster explicitly (ny rows are split over the ranks).  Rank 0 prints ONE JSON line.

Beside the pipeline, the N = 1 line carries (outside the timed region, one step each, `--no-extras` skips them)
  config3        BASELINE.json configs[2]: DinfFlowDir + AreaDinf on a 32768 x 32768 DEM (ms, Mcells/s, the accumulation sweeps' roofline)
  config4_strip  BASELINE.json configs[3] as ONE GPU sees it: the pipeline on a 65536 x 8192 strip (no neighbours)
  config5_strip  BASELINE.json configs[4] as ONE GPU sees it: DinfDecayAccum with weights, decay multipliers and 64 outlets on a
                 65536 x 8192 strip
  flowalg_16384  GridNet, weighted AreaD8, AreaDinf, DinfUpDependence, DinfRevAccum at 16384 x 16384 (ms per call)
and `--workload decay` times configs[4] itself: DinfDecayAccum -wg -o on ONE raster of 65536 columns x 8192*N rows in row strips
(the D-infinity angles come from PitRemove -> DinfFlowDir on the same strips, outside the timed region).
`--in-process` (N > 1): the N strips are N rank threads of THIS process on the library's own rank group (tdx_group: RCCL when every
rank has a GPU, peer copies when ranks share one - how eight 65536 x 8192 strips run on one 288 GB GPU); functional check, no torchrun.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
# algorithmic bytes per cell: pitremove z4+fel4, d8flowdir fel4+p2+sd8 4, aread8 p2+ad8 4
BYTES_PER_CELL = {"pitremove": 8, "d8flowdir": 10, "aread8": 6}
KCLASS_STAGE = {"relax": "pitremove", "bfs": "d8flowdir", "flatdir": "d8flowdir", "accum": "aread8"}


# FETCH_SIZE correction per kernel, CALIBRATED on scripts/micro/tilebw.hip's known byte counts (profiles/r04a_pmc_tilebw_*_summary.json, round 4): the counter
# tallies every request at 64 B - a wave row of 4-byte lanes (two 128-B requests) reads as half its bytes (factor 2.0: k4<0>, 2 147 MB read, 1 074 MB counted),
# a row of uint8 lanes (one 64-B request) as all of them, the level fields' mix of one int16 and one uint8 per lane and row as two thirds (factor 1.5: k16,
# 805 MB read, 537 MB counted).  WRITE_SIZE matched the bytes written (factor 1.0).
FETCH_FACTOR = {"LevelOp": 1.5, "PitOp": 2.0, "ad8_tile": 2.0}


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summaries of this same command: the newest
    profiles/rNN?_pmc_{fetch,write}_summary.json pair (scripts/gpu_profile.sh + scripts/pmc_summary.py; round 1's pair has no
    prefix).  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is corrected with the factor calibrated for the kernel's access width
    (FETCH_FACTOR above; MI355X_MICROARCH.md's factor 2 holds for 128-B requests only).  Returns (bytes, note) or (None, reason)."""
    import glob

    try:
        pairs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_fetch_summary.json"))) or [os.path.join(ROOT, "profiles", "pmc_fetch_summary.json")]
        ffile = pairs[-1]
        wfile = ffile.replace("pmc_fetch_summary", "pmc_write_summary")
        f = json.load(open(ffile))
        w = json.load(open(wfile))
        fk = [k for k in f if kernel_substr in k]
        wk = [k for k in w if kernel_substr in k]
        if not fk or not wk:
            return None, "kernel not in the PMC summaries"
        factor = FETCH_FACTOR.get(kernel_substr, 2.0)
        fetch = f[fk[0]]["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0 * factor
        write = w[wk[0]]["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
        return fetch + write, (f"profiles/{os.path.basename(ffile)} + {os.path.basename(wfile)}: ({factor}*FETCH_SIZE + WRITE_SIZE) KiB per dispatch; factor calibrated "
                               "on scripts/micro/tilebw.hip (profiles/r04a_pmc_tilebw_fetch_summary.json)")
    except Exception as e:  # no summaries committed
        return None, f"no PMC summary ({e.__class__.__name__})"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=0, help="square DEM edge length (cells); default 16384 at N = 1")
    ap.add_argument("--nx", type=int, default=0, help="columns of the whole raster")
    ap.add_argument("--ny", type=int, default=0, help="rows of the whole raster (split over the ranks)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-sample", type=int, default=2048, help="edge length of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--workload", choices=("d8", "decay"), default="d8", help="d8: PitRemove->D8FlowDir->AreaD8 (the metric); decay: DinfDecayAccum -wg -o on strips")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the config3 / config5_strip legs after the timed region")
    ap.add_argument("--in-process", action="store_true", help="N > 1: rank threads of this process (tdx_group) instead of one process per rank")
    ap.add_argument("--segments", type=int, default=0, choices=(0, 1, 2), help="--in-process: segment trace of the timed steps (2: one rank on the device at a "
                    "time) -> projected_ngpu_ms in the line (taudem_amd.distributed.project_critical_path)")
    ap.add_argument("--segments-out", default="", help="--segments: also write every rank's segment list to this JSON file (scripts/project_8gpu.py reads it)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: one process per GPU under torch.distributed.run (the launch the driver
    uses itself); the children's stdout (rank 0's JSON line) passes through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def offline_reference():
    """The reference's own timings at 2048^2 and 4096^2 (8 MPI ranks, build container) that came with the committed digests
    (tests/golden/large_digests.json), the fitted exponent of its flat resolution and the extrapolation to 16384^2."""
    import math
    try:
        d = json.load(open(os.path.join(ROOT, "tests", "golden", "large_digests.json")))
    except Exception:
        return None
    rows = {}
    for n in ("2048", "4096"):
        if n in d:
            t = d[n]["ref_seconds"]
            flats = t["d8flowdir"]["Resolve Flat time"]
            total = t["pitremove"]["Compute time"] + t["d8flowdir"]["Compute Slope time"] + flats + t["aread8"]["Compute time"]
            rows[n] = {"ranks": d[n]["ranks"], "pitremove_s": t["pitremove"]["Compute time"], "slope_s": t["d8flowdir"]["Compute Slope time"],
                       "flats_s": flats, "aread8_s": t["aread8"]["Compute time"], "pipeline_s": total, "mcells_per_s": int(n) ** 2 / total / 1e6}
    out = {"where": "build container, 8 host cores, mpiexec -n 8, reference built -O3 (oracle/Makefile)", "runs": rows}
    if "2048" in rows and "4096" in rows:
        expo = math.log(rows["4096"]["flats_s"] / rows["2048"]["flats_s"]) / math.log(4.0)     # seconds ~ cells^expo
        lin = rows["4096"]["pipeline_s"] - rows["4096"]["flats_s"]
        est = rows["4096"]["flats_s"] * 16.0 ** expo + lin * 16.0
        out["flat_resolution_exponent_in_cells"] = expo
        out["extrapolated_16384"] = {"pipeline_s": est, "mcells_per_s": 16384.0 ** 2 / est / 1e6, "note": "EXTRAPOLATION: flats_s * 16^exponent + linear stages * 16"}
    return out


def cpu_baseline(sample_n, seed):
    """Times the reference's own tools (oracle/_ref, built from /root/reference) on a bounded sample
    of the same workload, on this box's host cores; falls back to the C restatement ("port")."""
    import numpy as np

    import taudem_amd as T
    from oracle import oracle as O

    cores = os.cpu_count() or 1
    dem = O.synth_dem(sample_n, seed)
    cells = dem.size
    if O.ref_available("pitremove") and os.path.exists(O.MPIEXEC):
        try:
            with tempfile.TemporaryDirectory() as d:
                f = lambda s: os.path.join(d, s)  # noqa: E731
                T.write_raster(f("dem.tif"), dem, -9999.0, geotransform=(0.0, 30.0, 0.0, 30.0 * sample_n, 0.0, -30.0))
                ranks = max(1, min(cores, sample_n // 64))
                _, _, t1 = O.run_ref("pitremove", ["-z", f("dem.tif"), "-fel", f("fel.tif")], ranks, timeout=1800)
                _, _, t2 = O.run_ref("d8flowdir", ["-fel", f("fel.tif"), "-p", f("p.tif"), "-sd8", f("sd8.tif")], ranks, timeout=3600)
                _, _, t3 = O.run_ref("aread8", ["-p", f("p.tif"), "-ad8", f("ad8.tif")], ranks, timeout=1800)
                secs = t1["Compute time"] + t2["Compute Slope time"] + t2["Resolve Flat time"] + t3["Compute time"]
                return {"value": cells / secs / 1e6, "unit": "Mcells/s", "cores": ranks, "kind": "reference", "host_cores": cores,
                        "offline_reference": offline_reference(),
                        "sample": f"{sample_n}x{sample_n} synthetic DEM seed {seed}: reference pitremove+d8flowdir+aread8 under mpiexec -n {ranks}, "
                                  f"sum of the tools' own compute times = {secs:.2f} s (pitremove {t1['Compute time']:.2f}, slope "
                                  f"{t2['Compute Slope time']:.2f}, flats {t2['Resolve Flat time']:.2f}, aread8 {t3['Compute time']:.2f}); "
                                  "its flat resolution is superlinear (see offline_reference), so the rate falls with size: do not divide the GPU value by this one"}
        except Exception as e:  # e.g. MPICH runtime missing on the box
            sys.stderr.write(f"bench: reference baseline unavailable ({e}); timing the C restatement instead\n")
    t0 = time.time()
    fel = O.pitremove(dem, -9999.0)
    p, _, _ = O.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    O.aread8(p, -32768)
    secs = time.time() - t0
    return {"value": cells / secs / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": f"{sample_n}x{sample_n} synthetic DEM seed {seed}: C restatement (oracle/taudem_oracle.c), 1 thread, {secs:.2f} s"}


# ---- BASELINE.json configs[4] / configs[2] legs -------------------------------------------------------------------------------------
DECAY_BYTES_PER_CELL = 16    # dinfdecayaccum ang 4 + dm 4 + dsca 4, + 4 with weights
AREADINF_BYTES_PER_CELL = 8  # ang 4 + sca 4


def hashed_uniform(torch, device, nx, y0, nrows, salt, lo, hi, out):
    """out[r, c] = U[lo, hi) as a function of (global row y0 + r, column c, salt) only - the same raster whatever the partition.
    (32-bit integer mix carried in int64 lanes, in chunks of rows so that the temporaries stay small.)"""
    cols = torch.arange(nx, device=device, dtype=torch.int64).view(1, -1) * 0x85EBCA77
    for r0 in range(0, nrows, 1024):
        r1 = min(nrows, r0 + 1024)
        x = (torch.arange(y0 + r0, y0 + r1, device=device, dtype=torch.int64).view(-1, 1) * 0x9E3779B1 + cols + salt) & 0xFFFFFFFF
        x = ((x ^ (x >> 15)) * 0x2C1B3C6D) & 0xFFFFFFFF
        x = ((x ^ (x >> 12)) * 0x297A2D39) & 0xFFFFFFFF
        x = x ^ (x >> 15)
        out[r0:r1] = (lo + (hi - lo) * ((x >> 8).to(torch.float32) * (1.0 / 16777216.0))).to(torch.float32)


class DecayStrip:
    """BASELINE.json configs[4] on one strip: synthetic DEM -> PitRemove -> DinfFlowDir (outside the timed region), weights ~ U[0,1),
    decay multipliers ~ U[0.9,1) (SURVEY.md 8d iii), outlets on the highest-accumulation cell of each block of an 8 x 8 lattice over
    the WHOLE raster (8d iv: up to 64 outlets; a block's rows lie in one strip, so every rank finds its own outlets - the upstream
    closure then crosses the strips through the halo exchange).  step() = DinfDecayAccum -wg -o."""

    def __init__(self, torch, ctx, comm, nx, ny, y0, nyl, seed, T):
        from taudem_amd.distributed import StripPipeline
        self.pipe = StripPipeline(ctx, comm, nx, nyl)
        dev = torch.device(f"cuda:{ctx.device}")
        dem = self.pipe.empty(torch.float32)
        ctx.synth_dem((nyl, nx), seed=seed, x0=0, y0=y0, base_wavelength=T.synth_base_wavelength(max(nx, ny)), out=dem[1:nyl + 1])
        fel, _ = self.pipe.pitremove(dem, -9999.0)
        self.ang, slp, self.flowdir_stats = self.pipe.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
        del dem, fel
        sca, self.areadinf_stats = self.pipe.areadinf(self.ang, dx=30.0, dy=30.0, out=slp)
        # outlets: per lattice block the owned cell with the largest unweighted D-infinity area
        bh, bw = max(1, ny // 8), max(1, nx // 8)
        ox, oy = [], []
        for by in range(8):
            g0, g1 = by * bh, (ny if by == 7 else (by + 1) * bh)
            if g0 < y0 or g1 > y0 + nyl:
                continue                       # another rank's block (blocks nest in strips for 1, 2, 4, 8 ranks)
            for bx in range(8):
                c0, c1 = bx * bw, (nx if bx == 7 else (bx + 1) * bw)
                blk = sca[1 + g0 - y0:1 + g1 - y0, c0:c1]
                k = int(torch.argmax(blk))
                oy.append(g0 - y0 + 1 + k // (c1 - c0))       # strip-array row
                ox.append(c0 + k % (c1 - c0))
        self.outlets = (ox, oy)
        self.w = sca                            # the area raster is not needed any more: reuse it for the weights
        hashed_uniform(torch, dev, nx, y0, nyl, 0x1234567 + seed, 0.0, 1.0, self.w[1:nyl + 1])
        self.dm = self.pipe.empty(torch.float32)
        hashed_uniform(torch, dev, nx, y0, nyl, 0x7654321 + seed, 0.9, 1.0, self.dm[1:nyl + 1])
        self.out = self.pipe.empty(torch.float32)
        self.nyl = nyl

    def step(self):
        _, st = self.pipe.dinfdecayaccum(self.ang, self.dm, weights=self.w, outlets=self.outlets, out=self.out)
        return st

    def evaluated_cells(self, torch):
        return int((self.out[1:self.nyl + 1] != -3.402823466e38).sum())


def config3_leg(torch, ctx, seed):
    """BASELINE.json configs[2]: DinfFlowDir + AreaDinf on a 32768 x 32768 synthetic DEM, one step after one warm-up step."""
    n = 32768
    dem = ctx.synth_dem(n, seed=seed)
    fel = ctx.pitremove(dem, -9999.0)
    ang, slp, sca = dem, torch.empty_like(fel), torch.empty_like(fel)   # (the raw surface is not needed any more)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, s1 = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0, out=(ang, slp), stats=True)
        _, s2 = ctx.areadinf(ang, dx=30.0, dy=30.0, out=sca, stats=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    cells = float(n) * n
    sweep_ms, sweep_launches = s2["ms_accum"], max(1, s2["launches_accum"])
    achieved = AREADINF_BYTES_PER_CELL * cells / (sweep_ms * 1e-3) / 1e9
    return {"workload": f"{n}x{n} synthetic fractal DEM (seed {seed}, pit-filled), DinfFlowDir + AreaDinf in HBM on one GPU; parity at this size: "
                        "tests/test_gpu_fullsize.py::test_dinf_config3_at_32768 (sweep verifier + the restatement's linear-time checks)",
            "ms_per_step": ms, "mcells_per_s": cells / ms / 1e3, "dinfflowdir_ms": s1["ms_total"], "areadinf_ms": s2["ms_total"],
            "dinfflowdir_classes_ms": {k: s1["ms_" + k] for k in ("stencil", "bfs", "flatdir", "misc")},
            "areadinf_classes_ms": {k: s2["ms_" + k] for k in ("stencil", "accum")}, "areadinf_rounds": s2["rounds"],
            "roofline": {"bound": "hbm", "kernel": "areadinf: dsweep32::sweep_kernel (bulk rounds) + dsweep64::sweep_kernel (tail)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "launches": sweep_launches,
                         "avg_launch_ms": sweep_ms / sweep_launches, "algorithmic_bytes_per_cell": AREADINF_BYTES_PER_CELL,
                         "note": "dependency sweep: bound by the longest flow path (tile crossings x in-tile chain), not by bandwidth"}}


def config5_strip_leg(torch, ctx, seed, T):
    """BASELINE.json configs[4] as one GPU sees it: DinfDecayAccum -wg -o on a 65536 x 8192 raster (no neighbours)."""
    nx, ny = 65536, 8192
    job = DecayStrip(torch, ctx, None, nx, ny, 0, ny, seed, T)
    job.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = job.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    cells = float(nx) * ny
    return {"workload": f"{nx}x{ny} strip of the synthetic DEM (seed {seed}): DinfDecayAccum with weights, decay multipliers and {len(job.outlets[0])} outlets "
                        "(highest D-infinity area per block of an 8x8 lattice), in HBM on one GPU",
            "ms_per_step": ms, "mcells_per_s": cells / ms / 1e3, "rounds": st["rounds"], "cells_in_the_outlets_catchments": job.evaluated_cells(torch),
            "classes_ms": {k: st["ms_" + k] for k in ("stencil", "bfs", "accum", "misc")},
            "algorithmic_gb_per_s": DECAY_BYTES_PER_CELL * cells / (ms * 1e-3) / 1e9}


def config4_strip_leg(torch, ctx, seed, T):
    """BASELINE.json configs[3] as one GPU sees it: PitRemove -> D8FlowDir -> AreaD8 on ONE 65536 x 8192 strip of the 65536 x 65536 DEM (the generator
    is called with the whole raster's wavelength; no neighbours, so the halo exchanges and votes of the 8-GPU run are not in this number)."""
    nx, ny = 65536, 8192
    dem = ctx.synth_dem((ny, nx), seed=seed, base_wavelength=T.synth_base_wavelength(65536))
    fel = torch.empty_like(dem)
    p = torch.empty((ny, nx), dtype=torch.int16, device=dem.device)
    sd8, ad8 = torch.empty_like(dem), torch.empty_like(dem)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, s1 = ctx.pitremove(dem, -9999.0, out=fel, stats=True)
        _, _, s2 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, out=(p, sd8), stats=True)
        _, s3 = ctx.aread8(p, -32768, out=ad8, stats=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    cells = float(nx) * ny
    return {"workload": f"{nx}x{ny} strip (rows 0..8191 of the 65536x65536 synthetic DEM, seed {seed}): PitRemove->D8FlowDir->AreaD8 in HBM on one GPU, no neighbours",
            "ms_per_step": ms, "mcells_per_s": cells / ms / 1e3, "stage_ms": {"pitremove": s1["ms_total"], "d8flowdir": s2["ms_total"], "aread8": s3["ms_total"]},
            "flats_initial": s2["flats_initial"], "levels_fall": s2["levels_fall"], "max_level_per_iteration": {"fall": s2["levels_fall_max"], "rise": s2["levels_rise_max"], "int16_limit": 32766}, "pit_rounds": s1["rounds"],
            "note": "8 such strips = configs[3]: the `config4_8strips_one_gpu` leg of this line runs them as eight rank threads on this GPU (exchange / all-reduce counts, "
                    "projected 8-GPU critical path); profiles/r05*_8strips_65536_d8.json"}


def comm_latency_leg(ctx, nx=65536, reps=1000):
    """Latency of the strip protocol's two collectives over RCCL with ONE rank talking to itself (all a one-GPU box can run: taudem_amd/csrc/comm.cpp,
    tdx_rccl_latency): a grouped ncclSend/ncclRecv of one boundary row (nx float32) and the termination vote (ncclAllReduce of one device int64 + the
    device-to-host read of the result + the wait for it).  No link is involved, so these are LOWER bounds of the latencies between GPUs; they replace the
    assumed 10 / 30 us as the projection's defaults, next to a pessimistic 50 / 100 us."""
    import ctypes as C
    out = (C.c_double * 2)()
    rc = ctx._lib.tdx_rccl_latency(ctx._h, reps, nx * 4, out)
    if rc != 0:
        raise RuntimeError("tdx_rccl_latency failed")
    return {"rccl_one_rank_self": {"exchange_us": out[0], "vote_us": out[1], "bytes_per_direction": nx * 4, "repetitions": reps,
                                   "note": "one rank sending to itself: the software path of a collective without a link - a lower bound of the 8-GPU latencies"}}


def config4_8strips_leg(torch, T, args, rccl_latency_us=None, lone_stage_ms=None):
    """BASELINE.json configs[3] itself - the 65536 x 65536 raster in EIGHT strips of 65536 x 8192 - on the ONE GPU the driver gives this run: eight rank
    threads on the library's rank group (peer transport), every halo exchange, vote and cross-strip dependency of the 8-GPU protocol included.  One timed
    step with the ranks sharing the device (a functional figure, not a throughput), then one traced step with the ranks taking turns on it, from which
    the critical path of the run with one GPU per rank is PROJECTED (sum over the segments between collectives of the slowest rank + collectives x
    assumed latencies; scripts/project_8gpu.py)."""
    import copy
    a = copy.copy(args)
    a.workload, a.warmup, a.steps, a.segments, a.segments_out = "d8", 1, 1, 2, ""
    a.rccl_latency_us = rccl_latency_us
    line = strips_in_process(torch, T, a, 8, 65536, 65536)
    if lone_stage_ms and "projected_ngpu_ms" in line:   # work of the eight strips against eight lone strips (no neighbours): what the exchanges' re-relaxations add
        line["projected_ngpu_ms"]["redundant_work"] = {k: v["sum_over_ranks_ms"] / (8.0 * lone_stage_ms[k]) for k, v in line["projected_ngpu_ms"]["per_stage"].items()
                                                       if lone_stage_ms.get(k)}
    keep = ("ms_per_step", "stage_ms_per_step_rank0", "comm", "checks", "max_level_per_iteration", "projected_ngpu_ms")
    out = {"workload": line["config"]["workload"], "note": "eight ranks share ONE GPU: ms_per_step is functional only; projected_ngpu_ms is a projection, not a measurement"}
    out.update({k: line[k] for k in keep if k in line})
    return out


def flowalg_leg(torch, ctx, seed):
    """The accumulation tools beside the headline path at 16384^2, one call each after a warm-up call, library-side HIP-event
    time of the call: GridNet, weighted AreaD8, AreaDinf, DinfUpDependence, DinfRevAccum - so that the dependency sweeps' targets are under the driver's clock."""
    n = 16384
    dem = ctx.synth_dem(n, seed=seed)
    fel = ctx.pitremove(dem, -9999.0)
    del dem
    p, _ = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, want_slope=False)
    ang, slp = ctx.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
    del fel, slp
    g = torch.Generator(device=p.device).manual_seed(7)
    w = torch.rand((n, n), device=p.device, dtype=torch.float32, generator=g)
    dg32 = (torch.rand((n, n), device=p.device, generator=g) < 0.01).to(torch.int32)
    ms = {}

    def timed(name, fn):
        fn()
        torch.cuda.synchronize()
        ms[name] = fn()[-1]["ms_total"]
    timed("gridnet", lambda: ctx.gridnet(p, -32768, 30.0, 30.0, stats=True))
    timed("aread8_weighted", lambda: ctx.aread8(p, weights=w, stats=True))
    timed("areadinf", lambda: ctx.areadinf(ang, dx=30.0, dy=30.0, stats=True))
    timed("dinfupdependence", lambda: ctx.dinfupdependence(ang, dg32, dx=30.0, dy=30.0, stats=True))
    timed("dinfrevaccum", lambda: ctx.dinfrevaccum(ang, w, dx=30.0, dy=30.0, stats=True))
    return {"workload": f"{n}x{n} synthetic fractal DEM (seed {seed}, pit-filled), D8 / D-infinity directions from the library, random weights / indicator grid in HBM; "
                        "ms per call (one call after a warm-up call)", "ms": ms}


def decay_line(world, nx, ny, args, ms_per_step, st, comm_info, job_cells_evaluated, functional):
    total = float(nx) * float(ny)
    out = {"metric": "Mcells/s (DinfDecayAccum -wg -o)", "value": total / ms_per_step / 1e3, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"{nx} columns x {ny} rows synthetic fractal DEM (seed {args.seed}) in {world} row strip(s): DinfDecayAccum with weights, decay "
                                  "multipliers and up to 64 outlets (BASELINE.json configs[4] at 8 strips of 65536 x 8192)", "nx": nx, "ny": ny,
                      "total_cells": int(total), "cells_in_the_outlets_catchments": job_cells_evaluated},
           "rank0_classes_ms": {k: st["ms_" + k] for k in ("stencil", "bfs", "accum", "misc")}, "rank0_rounds": st["rounds"], "outer_rounds": st["cells_evaluated"]}
    if comm_info:
        out["comm"] = comm_info
    if functional:
        out["functional_only"] = "ranks share GPUs: a check of the strip protocol at this size, not a throughput figure"
    return out


def run_in_process(args):
    """N strips as N rank threads of this process (taudem_amd.distributed.StripGroup): the library's own rank group."""
    import torch

    import taudem_amd as T

    world = args.gpus
    nx, ny = (args.nx or args.ny or 65536), (args.ny or args.nx or 8192 * world)
    if args.size:
        nx, ny = args.size, args.size * world
    line = strips_in_process(torch, T, args, world, nx, ny)
    print(json.dumps(line), flush=True)


def strips_in_process(torch, T, args, world, nx, ny):
    """One raster of nx columns x ny rows in `world` row strips, one rank thread per strip on the library's rank group; returns the line.  With
    args.segments the timed steps are followed by ONE traced step (option "segment_trace") from which the critical path of a run with one GPU per
    rank is projected (taudem_amd.distributed.project_critical_path)."""
    import threading

    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows

    ndev = torch.cuda.device_count()
    devices = [r % ndev for r in range(world)]
    parts = partition_rows(ny, world)
    bar = threading.Barrier(world)
    lib = T.load()

    def counters(c):
        import ctypes as C
        e, a = C.c_int64(), C.c_int64()
        lib.tdx_context_comm_counters(c._h, C.byref(e), C.byref(a))
        return e.value, a.value

    with StripGroup(world, nx, devices) as grp:
        def rank_main(r, c, comm):
            torch.cuda.set_device(devices[r])
            y0, y1 = parts[r]
            nyl = y1 - y0
            res = {}
            if args.workload == "decay":
                job = DecayStrip(torch, c, comm, nx, ny, y0, nyl, args.seed, T)
                step = job.step
            else:
                pipe = StripPipeline(c, comm, nx, nyl)
                dem = pipe.empty(torch.float32)
                c.synth_dem((nyl, nx), seed=args.seed, x0=0, y0=y0, base_wavelength=T.synth_base_wavelength(max(nx, ny)), out=dem[1:nyl + 1])
                fel, p, sd8, ad8 = pipe.empty(torch.float32), pipe.empty(torch.int16), pipe.empty(torch.float32), pipe.empty(torch.float32)
                marks = []

                def step():
                    c0 = counters(c)
                    _, s1 = pipe.pitremove(dem, -9999.0, out=fel)
                    c1 = counters(c)
                    _, _, s2 = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0, out=(p, sd8))
                    c2 = counters(c)
                    _, s3 = pipe.aread8(p, -32768, contcheck=False, out=ad8)   # (no edge contamination: every cell with a direction gets its count)
                    c3 = counters(c)
                    marks.append([(b[0] - a[0], b[1] - a[1]) for a, b in ((c0, c1), (c1, c2), (c2, c3))])
                    return s1, s2, s3
            try:
                for _ in range(args.warmup):
                    step()
                bar.wait(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    st = step()
                torch.cuda.synchronize(); bar.wait()
                res["elapsed"] = time.perf_counter() - t0
                if args.segments:   # the group's own transport: one exchange / one vote as the protocol issues them, all ranks at once
                    import ctypes as C
                    lat = (C.c_double * 2)()
                    bar.wait()
                    if lib.tdx_comm_latency(c._h, comm.ptr(), 200, nx * 4, lat) == 0:
                        res["latency_us"] = (lat[0], lat[1])
                    bar.wait()
                if args.segments:   # one more step, traced (mode 2: the rank threads take turns on the device - not a step to time)
                    c.set_option("segment_trace", args.segments)
                    bar.wait()
                    step()
                    torch.cuda.synchronize(); bar.wait()
                    res["segments"] = c.segments()
                    c.set_option("segment_trace", 0)
            except BaseException:
                bar.abort()     # a rank that fails must not leave the others at the timing barrier
                raise
            res["stats"] = st
            if args.workload == "decay":
                res["evaluated"] = job.evaluated_cells(torch)
                res["outlets"] = len(job.outlets[0])
                res["counters"] = counters(c)
            else:
                res["marks"] = marks[-1]
                # conservation of AreaD8 on this strip (src/aread8.cpp:231-256): every owned cell with a direction is counted exactly once at the
                # cells that drain out of the raster or into a cell without direction - checked through the global count of evaluated cells
                own = ad8[1:nyl + 1]
                res["ad8_max"] = float(own.max())
                res["ad8_evaluated"] = int((own >= 1).sum())
                res["p_valid"] = int(((p[1:nyl + 1] >= 1) & (p[1:nyl + 1] <= 8)).sum())
            return res
        res = grp.run(rank_main)
        transport = grp.transport
    elapsed = max(r["elapsed"] for r in res)
    ms = elapsed / args.steps * 1e3
    functional = len(set(devices)) < world
    if args.workload == "decay":
        e, a = res[0]["counters"]
        line = decay_line(world, nx, ny, args, ms, res[0]["stats"], {"transport": transport + " (in-process rank group)", "rank0_exchanges_total": e,
                                                                     "rank0_allreduces_total": a, "outlets": sum(r["outlets"] for r in res)},
                          sum(r["evaluated"] for r in res), functional)
    else:
        total = float(nx) * float(ny)
        s1, s2, s3 = res[0]["stats"]
        last = res[0]["marks"]
        evaluated, valid = sum(r["ad8_evaluated"] for r in res), sum(r["p_valid"] for r in res)
        line = {"metric": "Mcells/s (PitRemove->D8FlowDir->AreaD8 pipeline)", "value": total / ms / 1e3, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{nx} columns x {ny} rows synthetic fractal DEM (seed {args.seed}) row-partitioned into {world} strips of {nx} x {ny // world} "
                                       f"(+ halo rows) on {len(set(devices))} GPU(s), PitRemove->D8FlowDir->AreaD8 in HBM", "nx": nx, "ny": ny, "total_cells": int(total)},
                "stage_ms_per_step_rank0": {"pitremove": s1["ms_total"], "d8flowdir": s2["ms_total"], "aread8": s3["ms_total"]},
                "comm": {"transport": transport + " (in-process rank group)",
                         "exchanges_per_step": {"pitremove": last[0][0], "d8flowdir": last[1][0], "aread8": last[2][0]},
                         "allreduces_per_step": {"pitremove": last[0][1], "d8flowdir": last[1][1], "aread8": last[2][1]}},
                "checks": {"ad8_cells_evaluated": evaluated, "cells_with_direction": valid, "every_directed_cell_evaluated": evaluated == valid,
                           "ad8_max": max(r["ad8_max"] for r in res)},
                "max_level_per_iteration": {"fall": s2["levels_fall_max"], "rise": s2["levels_rise_max"], "int16_limit": 32766, "flat_iterations": s2["flat_iterations"]}}
        if functional:
            line["functional_only"] = "ranks share GPUs: a check of the strip protocol at this size, not a throughput figure"
    if args.segments:
        from taudem_amd.distributed import project_critical_path
        logs = [r["segments"] for r in res]
        # collective latencies of the projection: the RCCL figures measured on this box with one rank (a lower bound: no link), if the caller has them
        # (args.rccl_latency_us), else the assumed 10 / 30 us; the sensitivity rows show what the projection does with slower collectives
        meas = getattr(args, "rccl_latency_us", None)
        ex_us, vote_us = (meas if meas else (10.0, 30.0))
        proj = project_critical_path(logs, ex_us, vote_us)
        line["projected_ngpu_ms"] = {"per_step": proj["total_ms"],
                                     "per_stage": {k: {kk: vv for kk, vv in v.items() if kk != "phases"} for k, v in proj["per_stage"].items()},
                                     "phases": {k: v["phases"] for k, v in proj["per_stage"].items()},
                                     "assumed": dict(proj["assumed"], source="measured: RCCL, one rank to itself (lower bound)" if meas else "assumed"),
                                     "sensitivity": [{"exchange_us": e, "vote_us": v, "per_step": project_critical_path(logs, e, v)["total_ms"]}
                                                     for e, v in ((10.0, 30.0), (ex_us, vote_us), (50.0, 100.0))]}
        if all("latency_us" in r for r in res):
            line.setdefault("comm", {})["latency_us"] = {"transport": transport + " (in-process rank group, " + str(world) + " rank threads on " + str(len(set(devices))) + " GPU(s))",
                                                        "exchange_us": max(r["latency_us"][0] for r in res), "vote_us": max(r["latency_us"][1] for r in res),
                                                        "note": "one boundary-row exchange / one termination vote as the protocol issues them, slowest rank, 200 repetitions"}
        if args.segments_out:
            with open(args.segments_out, "w") as f:
                json.dump({"world": world, "nx": nx, "ny": ny, "workload": args.workload, "steps": 1, "mode": args.segments, "logs": logs}, f)
    return line


def main():
    args = parse()
    if args.in_process and args.gpus > 1:
        return run_in_process(args)
    import torch
    import torch.distributed as dist

    import taudem_amd as T
    from taudem_amd.distributed import RcclStripComm, StripComm, StripPipeline, partition_rows

    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("TDX_BENCH_BACKEND", "nccl")   # "gloo": several ranks may share one GPU (functional check only)
    if args.gpus > 1 or world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev}"))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()
    else:
        dev = 0
        torch.cuda.set_device(0)
    # the whole raster: nx columns x ny rows
    if args.nx or args.ny:
        nx, ny = args.nx or args.ny, args.ny or args.nx
    elif args.size:
        nx, ny = args.size, args.size * world
    elif world == 1:
        nx = ny = 16384                       # BASELINE.json configs[1]
    else:
        nx, ny = 65536, 8192 * world          # 65536 x 8192 strips; world = 8: BASELINE.json configs[3] (65536 x 65536)
    ctx = T.Context(dev)
    device = torch.device(f"cuda:{dev}")

    if args.workload == "decay":
        # BASELINE.json configs[4]: DinfDecayAccum -wg -o on row strips (one rank per GPU; world = 1: one raster, no neighbours)
        if not (args.nx or args.ny or args.size):
            nx, ny = 65536, 8192 * world
        y0, y1 = partition_rows(ny, world)[rank]
        comm = None if world == 1 else (RcclStripComm(ctx, nx) if backend == "nccl" else StripComm(nx, device=dev))
        job = DecayStrip(torch, ctx, comm, nx, ny, y0, y1 - y0, args.seed, T)

        def sync():
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            job.step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st = job.step()
        sync()
        elapsed = time.perf_counter() - t0
        evaluated, outlets = job.evaluated_cells(torch), len(job.outlets[0])
        if dist.is_initialized():
            t = torch.tensor([elapsed, float(evaluated), float(outlets)], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
            elapsed, evaluated, outlets = float(t[0]), int(t[1]), int(t[2])
        info = None
        if comm is not None:
            info = {"transport": getattr(comm, "backend", backend), "rank0_exchanges_total": comm.exchanges, "rank0_allreduces_total": comm.allreduces, "outlets": outlets}
        line = json.dumps(decay_line(world, nx, ny, args, elapsed / args.steps * 1e3, st, info, evaluated, backend != "nccl" and world > 1))
        sys.stdout.flush()
        if dist.is_initialized():
            dist.barrier()
        if rank == 0:
            print(line, flush=True)
        if dist.is_initialized():
            sys.stderr.flush()
            os._exit(0)
        return

    force_strips = os.environ.get("TDX_BENCH_FORCE_STRIPS") == "1"   # the strip path (and its transport) with one rank
    if force_strips and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
    if world == 1 and not force_strips:
        # one nx x ny raster on one GPU
        y0, nyl = 0, ny
        dem = ctx.synth_dem((ny, nx), seed=args.seed)
        fel = torch.empty_like(dem)
        p = torch.empty((ny, nx), dtype=torch.int16, device=device)
        sd8 = torch.empty_like(dem)
        ad8 = torch.empty_like(dem)
        comm = None

        def step():
            _, s1 = ctx.pitremove(dem, -9999.0, out=fel, stats=True)
            _, _, s2 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, out=(p, sd8), stats=True)
            _, s3 = ctx.aread8(p, -32768, out=ad8, stats=True)
            return s1, s2, s3
    else:
        # ONE raster of nx columns x ny rows, row-partitioned like linearpart (src/linearpart.h:133-134): rank r owns
        # ny // world rows (the remainder goes to the last rank) plus one halo row on each side.
        y0, y1 = partition_rows(ny, world)[rank]
        nyl = y1 - y0
        comm = RcclStripComm(ctx, nx) if backend == "nccl" else StripComm(nx, device=dev)
        pipe = StripPipeline(ctx, comm, nx, nyl)
        dem = pipe.empty(torch.float32)
        ctx.synth_dem((nyl, nx), seed=args.seed, x0=0, y0=y0, base_wavelength=T.synth_base_wavelength(max(nx, ny)), out=dem[1:nyl + 1])
        fel = pipe.empty(torch.float32)
        p = pipe.empty(torch.int16)
        sd8 = pipe.empty(torch.float32)
        ad8 = pipe.empty(torch.float32)
        comm_marks = []

        def step():
            c0 = (comm.exchanges, comm.allreduces)
            _, s1 = pipe.pitremove(dem, -9999.0, out=fel)
            c1 = (comm.exchanges, comm.allreduces)
            _, _, s2 = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0, out=(p, sd8))
            c2 = (comm.exchanges, comm.allreduces)
            _, s3 = pipe.aread8(p, -32768, out=ad8)
            c3 = (comm.exchanges, comm.allreduces)
            comm_marks.append([(b[0] - a[0], b[1] - a[1]) for a, b in ((c0, c1), (c1, c2), (c2, c3))])
            return s1, s2, s3

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = None
    for _ in range(args.steps):
        st = step()
        if acc is None:
            acc = [dict(s) for s in st]
        else:
            for a, s in zip(acc, st):
                for k, v in s.items():
                    if k.startswith("ms_") or k.startswith("launches_"):
                        a[k] += v
    barrier()
    elapsed = time.perf_counter() - t0
    # One more pass of the same step, outside the timed region, with every launch of the tile-relaxation kernel
    # bracketed by HIP events on the library's stream (two event records per launch would cost ~5 % inside it):
    # the dominant kernel's average launch duration for the roofline object.
    ctx.set_option("kernel_timing", 1)
    prof = step()
    ctx.set_option("kernel_timing", 0)
    barrier()
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = None
    if rank == 0:
        cells = float(nx) * float(nyl)       # cells of rank 0's strip (the kernel-level figures below are rank 0's)
        total_cells = float(nx) * float(ny)
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells * args.steps / elapsed / 1e6
        stage_ms = {"pitremove": acc[0]["ms_total"], "d8flowdir": acc[1]["ms_total"], "aread8": acc[2]["ms_total"]}
        # kernel classes over the timed region (HIP events on the library's own stream, rank 0's strip)
        klass = {}
        for a in acc:
            for name in ("stencil", "relax", "bfs", "flatdir", "accum", "misc"):
                klass[name] = klass.get(name, 0.0) + a["ms_" + name]
                klass["n_" + name] = klass.get("n_" + name, 0) + a["launches_" + name]
        # per-stage view of the dominant kernel class: the tile-relaxation kernel of pitremove ("relax") and of
        # flat resolution ("bfs") are the same kernel template (tilek::relax_kernel) with different operators
        per_stage = {"pitremove/relax_kernel<PitOp>": (prof[0]["ms_tilek"], prof[0]["launches_tilek"], "pitremove"),
                     "d8flowdir/relax_kernel<LevelOp>": (prof[1]["ms_tilek"], prof[1]["launches_tilek"], "d8flowdir"),
                     "aread8/tile kernels": (prof[2]["ms_stencil"], prof[2]["launches_stencil"], "aread8"),
                     "aread8/big-cell evaluation": (prof[2]["ms_misc"], prof[2]["launches_misc"], "aread8")}
        dom = max(per_stage, key=lambda k: per_stage[k][0])
        dms, dlaunch, stage = per_stage[dom]
        launches = max(1, dlaunch)
        avg_ms = dms / launches
        bytes_per_launch = BYTES_PER_CELL[stage] * cells / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # the streaming 3x3 stencil that the 40 %-of-HBM target of BASELINE.json is about: D8 slope pass
        slope_ms = acc[1]["ms_stencil"] / max(1, acc[1]["launches_stencil"])
        slope_gbs = 10.0 * cells / (slope_ms * 1e-3) / 1e9 if slope_ms > 0 else 0.0
        traffic, traffic_note = pmc_traffic("LevelOp" if "LevelOp" in dom else ("PitOp" if "PitOp" in dom else "ad8_tile"))
        out = {
            "metric": "Mcells/s (PitRemove->D8FlowDir->AreaD8 pipeline)",
            "value": value,
            "unit": "Mcells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"{nx}x{ny} synthetic fractal DEM (seed {args.seed}), PitRemove->D8FlowDir->AreaD8 in HBM on one GPU; parity at this "
                                    "configuration: tests/test_gpu_large_golden.py (digests of the reference / restatement outputs)" if comm is None else
                                    f"{nx} columns x {ny} rows synthetic fractal DEM (seed {args.seed}) row-partitioned over {world} GPU(s) into strips of "
                                    f"{nx} x {ny // world} (+ halo rows), PitRemove->D8FlowDir->AreaD8 in HBM"),
                       "nx": nx, "ny": ny, "cells_per_gpu": int(total_cells / world), "total_cells": int(total_cells),
                       "multi_gpu": ("row strips, halo rows + cross-strip dependencies over " + getattr(comm, "backend", backend)) if comm is not None else "single GPU"},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "stage_mcells_per_s": {k: total_cells / (v / args.steps) / 1e3 for k, v in stage_ms.items() if v > 0},
            "kernel_class_ms_per_step": {k: klass[k] / args.steps for k in ("stencil", "relax", "bfs", "flatdir", "accum", "misc")},
            "kernel_class_launches_per_step": {k: klass["n_" + k] / args.steps for k in ("stencil", "relax", "bfs", "flatdir", "accum", "misc")},
            "roofline": {"bound": "hbm", "kernel": dom, "stage": stage, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note, "avg_launch_ms": avg_ms, "launches_per_step": launches,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "dependency-driven sweep: bound by (critical path in tiles) x launch, not by bandwidth (SURVEY.md 8d)"},
            "roofline_streaming_stencil": {"kernel": "d8_slope_kernel", "algorithmic_bytes_per_cell": 10, "avg_launch_ms": slope_ms,
                                           "achieved": slope_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": slope_gbs / HBM_PEAK_GBS},
            "flats": {"initial": acc[1]["flats_initial"], "left": acc[1]["flats_left"], "iterations": acc[1]["flat_iterations"],
                      "levels_fall": acc[1]["levels_fall"], "levels_rise": acc[1]["levels_rise"], "max_level_per_iteration": {"fall": acc[1]["levels_fall_max"], "rise": acc[1]["levels_rise_max"], "int16_limit": 32766}, "pit_rounds": acc[0]["rounds"],
                      "ad8_big_cells": acc[2]["cells_evaluated"], "ad8_outer_rounds": acc[2]["rounds"]},
        }
        if comm is not None:
            last = comm_marks[-1]
            out["comm"] = {"transport": getattr(comm, "backend", backend),
                           "exchanges_per_step": {"pitremove": last[0][0], "d8flowdir": last[1][0], "aread8": last[2][0]},
                           "allreduces_per_step": {"pitremove": last[0][1], "d8flowdir": last[1][1], "aread8": last[2][1]}}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.seed)
        if world == 1 and comm is None and not args.no_extras and not (args.size or args.nx or args.ny):
            # the other single-GPU configurations of BASELINE.json, one step each, outside the timed region (an error in one of them must
            # not cost the line of record)
            del dem, fel, p, sd8, ad8
            torch.cuda.empty_cache()
            for key, leg in (("config3", lambda: config3_leg(torch, ctx, args.seed)), ("config4_strip", lambda: config4_strip_leg(torch, ctx, args.seed, T)),
                             ("config5_strip", lambda: config5_strip_leg(torch, ctx, args.seed, T)), ("flowalg_16384", lambda: flowalg_leg(torch, ctx, args.seed))):
                try:
                    out[key] = leg()
                except Exception as e:   # noqa: BLE001
                    out[key] = {"error": f"{e.__class__.__name__}: {e}"}
                torch.cuda.empty_cache()
            # last: the eight strips need ~190 GB of their own - this context's scratch arena (sized by config3) goes first
            rccl_us = None
            try:
                out["comm_latency_us"] = comm_latency_leg(ctx)
                rccl_us = (out["comm_latency_us"]["rccl_one_rank_self"]["exchange_us"], out["comm_latency_us"]["rccl_one_rank_self"]["vote_us"])
            except Exception as e:   # noqa: BLE001
                out["comm_latency_us"] = {"error": f"{e.__class__.__name__}: {e}"}
            ctx.close()
            torch.cuda.empty_cache()
            try:
                lone = out.get("config4_strip", {}).get("stage_ms")
                out["config4_8strips_one_gpu"] = config4_8strips_leg(torch, T, args, rccl_us, lone)
                if "comm" in out["config4_8strips_one_gpu"] and "latency_us" in out["config4_8strips_one_gpu"]["comm"]:
                    out["comm_latency_us"]["peer_8_rank_threads_one_gpu"] = out["config4_8strips_one_gpu"]["comm"]["latency_us"]
            except Exception as e:   # noqa: BLE001
                out["config4_8strips_one_gpu"] = {"error": f"{e.__class__.__name__}: {e}"}
        line = json.dumps(out)
    # The JSON line is the LAST thing on stdout: whatever any rank or the C runtimes (RCCL prints a version banner
    # through C stdio) have buffered goes out first, and the ranks leave without running teardown code that prints.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        print(line, flush=True)
    if dist.is_initialized():
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
