#!/usr/bin/env python3
"""Projects the critical path of an N-GPU strip run from the segment traces of an in-process run on ONE GPU.

    python bench.py --gpus 8 --in-process --segments 2 --steps 1 --segments-out gpurun_out/seg_d8.json
    python scripts/project_8gpu.py gpurun_out/seg_d8.json [--exchange-us 10] [--vote-us 30]

A strip run is, on every rank, the same sequence of segments of rank-local device work, each ended by a collective (halo exchange or
all-reduce: the outer loops of src/aread8.cpp:282-303, src/flood.cpp, src/d8.cpp with linearpart::share(), src/linearpart.h:313-384).
With `--segments 2` the rank threads take turns on the device, so that every segment is timed as it would run on a GPU of its own; then

    projected step = sum over segments of (max over ranks) + exchanges x exchange latency + all-reduces x vote latency.

The latencies are ASSUMPTIONS (defaults: 10 us for a one-row RCCL send/recv pair over xGMI, 30 us for an all-reduce of a few int64 plus
its device-to-host read): the output says so.  It is a projection, not a measurement; the 8-GPU run itself is the driver's."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from taudem_amd.distributed import project_critical_path  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--exchange-us", type=float, default=10.0)
    ap.add_argument("--vote-us", type=float, default=30.0)
    ap.add_argument("--use", choices=("wall", "device"), default="wall")
    ap.add_argument("--lone-ms", default="", help="stage=ms,... of ONE strip without neighbours (bench.py: config4_strip.stage_ms): prints the eight strips' work "
                                                 "against eight lone strips (\"redundant work\": what the re-relaxations after the exchanges add)")
    a = ap.parse_args()
    d = json.load(open(a.trace))
    logs = [[tuple(s) for s in lg] for lg in d["logs"]]
    proj = project_critical_path(logs, a.exchange_us, a.vote_us, a.use)
    steps = max(1, d.get("steps", 1))
    print(f"# {d['workload']}: {d['nx']} x {d['ny']} in {d['world']} strips, {steps} step(s); segment time = {a.use}; exchange {a.exchange_us} us, vote {a.vote_us} us")
    print("| stage | segments | exchanges | all-reduces | max-over-ranks work ms | latency ms | projected ms | sum over ranks ms |")
    print("|---|---|---|---|---|---|---|---|")
    for k, v in proj["per_stage"].items():
        print(f"| {k} | {v['segments'] // steps} | {v['exchanges'] // steps} | {v['allreduces'] // steps} | {v['work_ms'] / steps:.2f} | {v['latency_ms'] / steps:.2f} | "
              f"{v['ms'] / steps:.2f} | {v['sum_over_ranks_ms'] / steps:.2f} |")
        for pk, pv in v["phases"].items():
            print(f"|   {k} / {pk} | | | | | | {pv / steps:.2f} | |")
    print(f"| total | | | | | | {proj['total_ms'] / steps:.2f} | |")
    # how much of the projection is the ASSUMED collective latency: the same trace at three pairs of (exchange, vote) microseconds
    print("| sensitivity: exchange us / vote us | projected ms |")
    print("|---|---|")
    for e, v in ((10.0, 30.0), (a.exchange_us, a.vote_us), (50.0, 100.0)):
        print(f"| {e:g} / {v:g} | {project_critical_path(logs, e, v, a.use)['total_ms'] / steps:.2f} |")
    if a.lone_ms:
        lone = {k: float(v) for k, v in (kv.split("=") for kv in a.lone_ms.split(","))}
        print("| stage | sum over ranks ms | world x lone strip ms | redundant work |")
        print("|---|---|---|---|")
        for k, v in proj["per_stage"].items():
            if k in lone:
                print(f"| {k} | {v['sum_over_ranks_ms'] / steps:.1f} | {d['world'] * lone[k]:.1f} | {v['sum_over_ranks_ms'] / steps / (d['world'] * lone[k]):.2f} x |")
    print(json.dumps({"projected_ngpu_ms_per_step": proj["total_ms"] / steps, "assumed": proj["assumed"]}))


if __name__ == "__main__":
    main()
