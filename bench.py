#!/usr/bin/env python
"""Benchmark of the PitRemove -> D8FlowDir -> AreaD8 pipeline (BASELINE.json metric: Mcells/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size n]

One "step" = one pass of the three-stage pipeline over one synthetic fractal DEM that is already
resident in HBM (generated on the device by tdx_synth_dem_dev).  Default workload = BASELINE.json
configs[1]: 16384 x 16384 on one MI355X.  For N > 1 the driver launches one process per GPU
(torch.distributed, backend nccl = RCCL) and ONE raster of 16384 columns x 16384*N rows is
row-partitioned over the ranks (weak scaling: 16384 rows per GPU), halo rows and cross-strip
dependencies exchanged through taudem_amd.distributed.StripComm; see DESIGN.md "Multi-GPU".
Rank 0 prints ONE JSON line.
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
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summaries of this same command
    (profiles/pmc_fetch_summary.json, profiles/pmc_write_summary.json; scripts/gpu_pmc.sh + scripts/pmc_summary.py).
    FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads
    (MI355X_MICROARCH.md, HBM): it is doubled here.  Returns (bytes, note) or (None, reason)."""
    try:
        f = json.load(open(os.path.join(ROOT, "profiles", "pmc_fetch_summary.json")))
        w = json.load(open(os.path.join(ROOT, "profiles", "pmc_write_summary.json")))
        fk = [k for k in f if kernel_substr in k]
        wk = [k for k in w if kernel_substr in k]
        if not fk or not wk:
            return None, "kernel not in the PMC summaries"
        fetch = f[fk[0]]["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0 * 2.0
        write = w[wk[0]]["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
        return fetch + write, "profiles/pmc_{fetch,write}_summary.json: (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch"
    except Exception as e:  # no summaries committed
        return None, f"no PMC summary ({e.__class__.__name__})"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=16384, help="DEM edge length (cells)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-sample", type=int, default=1024, help="edge length of the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


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
                return {"value": cells / secs / 1e6, "unit": "Mcells/s", "cores": ranks, "kind": "reference",
                        "sample": f"{sample_n}x{sample_n} synthetic DEM seed {seed}: reference pitremove+d8flowdir+aread8 under mpiexec -n {ranks}, "
                                  f"sum of the tools' own compute times = {secs:.2f} s (pitremove {t1['Compute time']:.2f}, slope "
                                  f"{t2['Compute Slope time']:.2f}, flats {t2['Resolve Flat time']:.2f}, aread8 {t3['Compute time']:.2f}); "
                                  "flat resolution scales as N^1.5 so the rate falls with size"}
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
    from taudem_amd.distributed import StripComm, StripPipeline

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
    n = args.size
    ctx = T.Context(dev)
    device = torch.device(f"cuda:{dev}")

    force_strips = os.environ.get("TDX_BENCH_FORCE_STRIPS") == "1" and dist.is_initialized()   # exercise StripComm with one rank
    if world == 1 and not force_strips:
        # one n x n raster on one GPU (BASELINE.json configs[1])
        dem = ctx.synth_dem(n, seed=args.seed)
        fel = torch.empty_like(dem)
        p = torch.empty((n, n), dtype=torch.int16, device=device)
        sd8 = torch.empty_like(dem)
        ad8 = torch.empty_like(dem)
        comm = None

        def step():
            _, s1 = ctx.pitremove(dem, -9999.0, out=fel, stats=True)
            _, _, s2 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, out=(p, sd8), stats=True)
            _, s3 = ctx.aread8(p, -32768, out=ad8, stats=True)
            return s1, s2, s3
    else:
        # ONE raster of n columns x (n * world) rows, row-partitioned like linearpart (src/linearpart.h:133-134):
        # rank r owns rows [r*n, (r+1)*n) plus one halo row on each side; halo rows and cross-strip dependencies
        # travel through StripComm (RCCL send/recv + all-reduce).  Per-GPU work is fixed: weak scaling.
        comm = StripComm(n, device=dev)
        pipe = StripPipeline(ctx, comm, n, n)
        dem = pipe.empty(torch.float32)
        ctx.synth_dem((n, n), seed=args.seed, x0=0, y0=rank * n, base_wavelength=T.synth_base_wavelength(n), out=dem[1:n + 1])
        fel = pipe.empty(torch.float32)
        p = pipe.empty(torch.int16)
        sd8 = pipe.empty(torch.float32)
        ad8 = pipe.empty(torch.float32)

        def step():
            _, s1 = pipe.pitremove(dem, -9999.0, out=fel)
            _, _, s2 = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0, out=(p, sd8))
            _, s3 = pipe.aread8(p, -32768, out=ad8)
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
        cells = float(n) * float(n)          # cells per GPU
        ms_per_step = elapsed / args.steps * 1e3
        value = world * cells * args.steps / elapsed / 1e6
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
            "config": {"workload": (f"{n}x{n} synthetic fractal DEM, PitRemove->D8FlowDir->AreaD8 in HBM, bit-exact vs reference" if world == 1 else
                                    f"{n} columns x {n * world} rows synthetic fractal DEM row-partitioned over {world} GPUs ({n} rows each + halo rows), "
                                    "PitRemove->D8FlowDir->AreaD8 in HBM"),
                       "cells_per_gpu": int(cells), "total_cells": int(cells * world),
                       "multi_gpu": ("row strips, halo rows + cross-strip dependencies over " + ("RCCL" if backend == "nccl" else backend)) if world > 1 else "single GPU"},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
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
            out["comm"] = {"exchanges_per_step": comm.exchanges / (args.steps + args.warmup), "allreduces_per_step": comm.allreduces / (args.steps + args.warmup)}
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
