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


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summaries of this same command: the newest
    profiles/rNN?_pmc_{fetch,write}_summary.json pair (scripts/gpu_r02_profile.sh + scripts/pmc_summary.py; round 1's pair has no
    prefix).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads
    (MI355X_MICROARCH.md, HBM): it is doubled here.  Returns (bytes, note) or (None, reason)."""
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
        fetch = f[fk[0]]["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0 * 2.0
        write = w[wk[0]]["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
        return fetch + write, f"profiles/{os.path.basename(ffile)} + {os.path.basename(wfile)}: (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch"
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


def main():
    args = parse()
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
                      "levels_fall": acc[1]["levels_fall"], "levels_rise": acc[1]["levels_rise"], "pit_rounds": acc[0]["rounds"],
                      "ad8_big_cells": acc[2]["cells_evaluated"], "ad8_outer_rounds": acc[2]["rounds"]},
        }
        if comm is not None:
            last = comm_marks[-1]
            out["comm"] = {"transport": getattr(comm, "backend", backend),
                           "exchanges_per_step": {"pitremove": last[0][0], "d8flowdir": last[1][0], "aread8": last[2][0]},
                           "allreduces_per_step": {"pitremove": last[0][1], "d8flowdir": last[1][1], "aread8": last[2][1]}}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.seed)
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
