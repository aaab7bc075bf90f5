#!/usr/bin/env python
"""Benchmark of the PitRemove -> D8FlowDir -> AreaD8 pipeline (BASELINE.json metric: Mcells/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size n]

One "step" = one pass of the three-stage pipeline over one synthetic fractal DEM that is already
resident in HBM (generated on the device by tdx_synth_dem_dev).  Default workload = BASELINE.json
configs[1]: 16384 x 16384 on one MI355X.  For N > 1 the driver launches one process per GPU
(torch.distributed, backend nccl = RCCL); see DESIGN.md "Multi-GPU" for what each rank processes.
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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        world = dist.get_world_size()
    else:
        torch.cuda.set_device(0)
    dev = local_rank if world > 1 else 0
    n = args.size
    ctx = T.Context(dev)

    # every rank owns one n x n DEM (weak scaling: per-GPU work fixed); different seeds per rank
    dem = ctx.synth_dem(n, seed=args.seed + rank)
    fel = torch.empty_like(dem)
    p = torch.empty((n, n), dtype=torch.int16, device=dem.device)
    sd8 = torch.empty_like(dem)
    ad8 = torch.empty_like(dem)

    def step():
        _, s1 = ctx.pitremove(dem, -9999.0, out=fel, stats=True)
        _, _, s2 = ctx.d8flowdir(fel, -3.0e38, 30.0, 30.0, out=(p, sd8), stats=True)
        _, s3 = ctx.aread8(p, -32768, out=ad8, stats=True)
        return s1, s2, s3

    def barrier():
        if world > 1:
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
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dem.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        cells = float(n) * float(n)
        ms_per_step = elapsed / args.steps * 1e3
        value = world * cells * args.steps / elapsed / 1e6
        stage_ms = {"pitremove": acc[0]["ms_total"] / 1.0, "d8flowdir": acc[1]["ms_total"], "aread8": acc[2]["ms_total"]}
        # dominant kernel class over the timed region (HIP events on the library's stream)
        klass = {}
        for a in acc:
            for name in ("stencil", "relax", "bfs", "flatdir", "accum", "misc"):
                klass[name] = klass.get(name, 0.0) + a["ms_" + name]
                klass["n_" + name] = klass.get("n_" + name, 0) + a["launches_" + name]
        dom = max(("relax", "bfs", "flatdir", "accum", "stencil"), key=lambda k: klass[k])
        stage = KCLASS_STAGE.get(dom, "d8flowdir")
        launches = max(1, klass["n_" + dom])
        avg_ms = klass[dom] / launches
        bytes_per_launch = BYTES_PER_CELL[stage] * cells * args.steps / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
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
            "config": {"workload": f"{n}x{n} synthetic fractal DEM per GPU, PitRemove->D8FlowDir->AreaD8 in HBM, bit-exact vs reference",
                       "cells_per_gpu": int(cells), "multi_gpu": "independent DEM per rank (replicas)" if world > 1 else "single GPU"},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "kernel_class_ms_per_step": {k: klass[k] / args.steps for k in ("stencil", "relax", "bfs", "flatdir", "accum", "misc")},
            "kernel_class_launches_per_step": {k: klass["n_" + k] / args.steps for k in ("stencil", "relax", "bfs", "flatdir", "accum")},
            "roofline": {"bound": "hbm", "kernel": dom, "stage": stage, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms, "launches_per_step": launches / args.steps,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "flats": {"initial": acc[1]["flats_initial"], "left": acc[1]["flats_left"], "iterations": acc[1]["flat_iterations"],
                      "levels_fall": acc[1]["levels_fall"], "levels_rise": acc[1]["levels_rise"], "pit_rounds": acc[0]["rounds"]},
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.seed)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
