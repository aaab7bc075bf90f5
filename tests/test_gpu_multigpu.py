"""Multi-GPU plumbing on the (one-GPU) test box: the native transports of taudem_amd/csrc/comm.cpp, `tool --gpus N` (row strips
behind the command-line surface, N rank threads; ranks that share a GPU use the peer transport) against the reference's
rasters, and bench.py's own launch of one process per rank."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import taudem_amd as T
from conftest import bits_equal, describe_diff, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "taudem_amd", "bin")


def run(tool, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([os.path.join(BIN, tool), *args], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_rccl_transport_selftest(ctx):
    """ncclCommInitRank + grouped ncclSend/ncclRecv + ncclAllReduce on the context's stream (one rank talking to itself)."""
    rc = ctx._lib.tdx_rccl_selftest(ctx._h)
    assert rc == 0, T._lib.last_error(ctx._h)


def test_collective_latencies_are_measurable(ctx):
    """tdx_rccl_latency (one rank to itself) and tdx_comm_latency on a peer group of three rank threads: both primitives of the strip protocol answer with
    plausible microsecond figures (bench.py puts them in the line and uses the RCCL pair as the projection's default latencies)."""
    import threading

    out = (C.c_double * 2)()
    assert ctx._lib.tdx_rccl_latency(ctx._h, 50, 4096 * 4, out) == 0, T._lib.last_error(ctx._h)
    assert 0.5 < out[0] < 5000 and 0.5 < out[1] < 5000, tuple(out)
    lib = T.load()
    size, nx = 3, 4096
    devs = (C.c_int32 * size)(0, 0, 0)
    g = C.c_void_p()
    assert lib.tdx_group_create(size, devs, nx, C.byref(g)) == 0, T._lib.last_error(None)
    res, errors = [None] * size, []

    def rank_main(r):
        try:
            o = (C.c_double * 2)()
            rc = lib.tdx_comm_latency(C.c_void_p(lib.tdx_group_context(g, r)), C.c_void_p(lib.tdx_group_comm(g, r)), 20, nx * 4, o)
            assert rc == 0
            res[r] = (o[0], o[1])
        except BaseException as e:   # noqa: BLE001
            errors.append((r, repr(e)))
            lib.tdx_group_abort(g)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(size)]
    [t.start() for t in th]
    [t.join() for t in th]
    lib.tdx_group_destroy(g)
    assert not errors, errors
    assert all(0.5 < a < 1e5 and 0.5 < b < 1e5 for a, b in res), res


def test_release_scratch_and_go_on(ctx, oracle):
    """tdx_context_release_scratch: the arena goes, the next call builds what it needs."""
    dem = oracle.synth_dem((300, 280), 5)
    a = ctx.pitremove(dem, -9999.0)
    ctx.release_scratch()
    b = ctx.pitremove(dem, -9999.0)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(b.view(np.uint32), oracle.pitremove(dem, -9999.0).view(np.uint32))


def test_group_peer_transport_protocol():
    """tdx_group with 3 ranks on one GPU (peer transport): exchange + all-reduce from three threads, checked like the gloo protocol test."""
    import threading

    lib = T.load()
    size, nx = 3, 37
    devs = (C.c_int32 * size)(0, 0, 0)
    g = C.c_void_p()
    assert lib.tdx_group_create(size, devs, nx, C.byref(g)) == 0, T._lib.last_error(None)
    assert lib.tdx_group_transport(g) == b"peer"
    errors = []

    def rank_main(r):
        try:
            comm = C.cast(lib.tdx_group_comm(g, r), C.POINTER(T._lib.TdxComm)).contents
            ctxh = C.c_void_p(lib.tdx_group_context(g, r))
            assert comm.rank == r and comm.size == size and comm.capacity >= 16 * nx
            for it in range(3):
                n = nx * (it + 1)
                up = (np.arange(n) + 100 * r + it).astype(np.uint8)
                dn = (np.arange(n) + 100 * r + 50 + it).astype(np.uint8)
                fill = np.full(n, 255, np.uint8)
                for dst, src in ((comm.send_up, up), (comm.send_down, dn), (comm.recv_up, fill), (comm.recv_down, fill)):
                    assert lib.tdx_copy_to_device(ctxh, dst, src.ctypes.data, n) == 0
                assert comm.exchange(comm.user, n) == 0
                got_up, got_dn = np.empty(n, np.uint8), np.empty(n, np.uint8)
                assert lib.tdx_copy_to_host(ctxh, got_up.ctypes.data, comm.recv_up, n) == 0
                assert lib.tdx_copy_to_host(ctxh, got_dn.ctypes.data, comm.recv_down, n) == 0
                if r > 0:
                    assert np.array_equal(got_up, (np.arange(n) + 100 * (r - 1) + 50 + it).astype(np.uint8)), "recv_up = the upper neighbour's send_down"
                else:
                    assert got_up.min() == 255
                if r < size - 1:
                    assert np.array_equal(got_dn, (np.arange(n) + 100 * (r + 1) + it).astype(np.uint8)), "recv_down = the lower neighbour's send_up"
                else:
                    assert got_dn.min() == 255
            vals = (C.c_int64 * 3)(r + 1, 10 * (r + 1), -r)
            assert comm.allreduce(comm.user, vals, 3, 0) == 0
            assert list(vals) == [6, 60, -3]
            vals = (C.c_int64 * 2)(r, 7 - r)
            assert comm.allreduce(comm.user, vals, 2, 1) == 0
            assert list(vals) == [2, 7]
        except BaseException as e:   # noqa: BLE001
            errors.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(size)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    lib.tdx_group_destroy(g)
    assert not errors, errors


@pytest.mark.parametrize("case,ngpus", [("plain", 2), ("holes", 3), ("rect_dxdy", 4)])
def test_cli_gpus_n_matches_reference_outputs(tmp_path, case, ngpus):
    """pitremove / d8flowdir / aread8 (+ -wg, -o) / dinfflowdir / areadinf / dinfdecayaccum with --gpus N: pixels identical to the
    rasters of the real reference tools (which are themselves rank-count independent)."""
    g = load_golden(case)
    ny, nx = g["dem"].shape
    dx, dy = float(g["dx"]), float(g["dy"])
    gt = (1000.0, dx, 0.0, 5000.0 + dy * ny, 0.0, -dy)
    f = lambda s: str(tmp_path / s)  # noqa: E731
    N = ["--gpus", str(ngpus)]
    T.write_raster(f("dem.tif"), np.ascontiguousarray(g["dem"]), float(g["nodata"]), geotransform=gt)
    T.write_raster(f("w.tif"), np.ascontiguousarray(g["w"]), -9999.0, geotransform=gt)
    T.write_raster(f("dm.tif"), np.ascontiguousarray(g["dm"]), -9999.0, geotransform=gt)
    with open(f("outlets.txt"), "w") as fh:
        for x_, y_ in zip(*g["outlet_xy"]):
            fh.write(f"{float(x_)!r} {float(y_)!r}\n")
    out = run("pitremove", *N, "-z", f("dem.tif"), "-fel", f("fel.tif"), env={"TAUDEM_AMD_STATS": "1"})
    assert f"Processes: {ngpus}" in out
    run("d8flowdir", "-fel", f("fel.tif"), "-p", f("p.tif"), "-sd8", f("sd8.tif"), *N)
    run("aread8", "-p", f("p.tif"), "-ad8", f("ad8.tif"), *N)
    run("aread8", "-p", f("p.tif"), "-ad8", f("ad8w.tif"), "-wg", f("w.tif"), *N)
    run("aread8", "-p", f("p.tif"), "-ad8", f("ad8o.tif"), "-o", f("outlets.txt"), env={"TAUDEM_AMD_GPUS": str(ngpus)})
    run("dinfflowdir", *N, "-fel", f("fel.tif"), "-ang", f("ang.tif"), "-slp", f("slp.tif"))
    run("areadinf", *N, "-ang", f("ang.tif"), "-sca", f("sca.tif"))
    run("dinfdecayaccum", *N, "-ang", f("ang.tif"), "-dm", f("dm.tif"), "-dsca", f("dsca.tif"))
    from conftest import load_golden_gridnet
    h = load_golden_gridnet(case)
    T.write_raster(f("gmask.tif"), np.ascontiguousarray(h["mask_i32"]), -1, geotransform=gt)
    run("gridnet", "-p", f("p.tif"), "-plen", f("plen.tif"), "-tlen", f("tlen.tif"), "-gord", f("gord.tif"), *N)
    run("gridnet", *N, "-p", f("p.tif"), "-plen", f("plenm.tif"), "-tlen", f("tlenm.tif"), "-gord", f("gordm.tif"), "-mask", f("gmask.tif"), "-thresh", str(int(h["gn_thresh"])))
    run("gridnet", *N, "-p", f("p.tif"), "-plen", f("pleno.tif"), "-tlen", f("tleno.tif"), "-gord", f("gordo.tif"), "-o", f("outlets.txt"))
    for name, key, dt in (("plen", "plen", np.float32), ("tlen", "tlen", np.float32), ("gord", "gord", np.int16), ("plenm", "plen_m", np.float32),
                          ("tlenm", "tlen_m", np.float32), ("gordm", "gord_m", np.int16), ("pleno", "plen_o", np.float32), ("tleno", "tlen_o", np.float32),
                          ("gordo", "gord_o", np.int16)):
        a, _ = T.read_raster(f(name + ".tif"), dt)
        assert bits_equal(a, h[key]), describe_diff(a, h[key], f"gridnet {name} --gpus {ngpus}")
    for name, key, dt in (("fel", "fel", np.float32), ("p", "p", np.int16), ("sd8", "sd8", np.float32), ("ad8", "ad8", np.float32), ("ad8w", "ad8_w", np.float32),
                          ("ad8o", "ad8_outlets", np.float32), ("ang", "ang", np.float32), ("slp", "slp", np.float32), ("sca", "sca", np.float32),
                          ("dsca", "dsca", np.float32)):
        a, _ = T.read_raster(f(name + ".tif"), dt)
        assert bits_equal(a, g[key]), describe_diff(a, g[key], f"{name} --gpus {ngpus}")


def _bench(args, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell (no torchrun, no RANK): one process per rank, here two ranks sharing the GPU over gloo."""
    out = _bench(["--gpus", "2", "--nx", "640", "--ny", "512", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"], {"TDX_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["nx"] == 640 and out["config"]["ny"] == 512
    assert out["comm"]["exchanges_per_step"]["pitremove"] > 0 and out["value"] > 0


def test_bench_strip_path_with_native_rccl_comm():
    """The strip path of bench.py on one rank with the native RCCL communicator (ncclCommInitRank on this GPU)."""
    out = _bench(["--nx", "1024", "--ny", "512", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"], {"TDX_BENCH_FORCE_STRIPS": "1"})
    assert out["comm"]["transport"] == "rccl-native" and out["value"] > 0


@pytest.mark.parametrize("workload", ["d8", "decay"])
def test_bench_in_process_rank_group(workload):
    """`bench.py --gpus 4 --in-process`: four strips as four rank threads of one process on the library's own rank group (peer transport on a
    one-GPU box) - the launcher of the eight-strips-on-one-GPU functional runs (profiles/r03*_8strips_*.json), at a size that takes a second."""
    out = _bench(["--gpus", "4", "--in-process", "--workload", workload, "--nx", "640", "--ny", "768", "--steps", "1", "--warmup", "0"], {})
    assert out["n_gpus"] == 4 and out["value"] > 0 and "peer" in out["comm"]["transport"]
    if workload == "d8":
        assert out["checks"]["every_directed_cell_evaluated"] and out["comm"]["exchanges_per_step"]["aread8"] > 0
    else:
        assert out["config"]["cells_in_the_outlets_catchments"] > 0 and out["comm"]["outlets"] == 64


def test_bench_decay_workload_two_ranks_gloo():
    """BASELINE.json configs[4]'s launcher (`bench.py --gpus N --workload decay`, one process per rank) with two ranks sharing the GPU over gloo;
    the same raster through one rank must evaluate the same number of cells."""
    two = _bench(["--gpus", "2", "--workload", "decay", "--nx", "512", "--ny", "512", "--steps", "1", "--warmup", "0"], {"TDX_BENCH_BACKEND": "gloo"})
    one = _bench(["--workload", "decay", "--nx", "512", "--ny", "512", "--steps", "1", "--warmup", "0"], {})
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["cells_in_the_outlets_catchments"] == one["config"]["cells_in_the_outlets_catchments"] > 0


@pytest.mark.parametrize("mode", [1, 2])
def test_segment_trace_of_a_strip_run(oracle, mode):
    """Option "segment_trace" (include/taudem_amd.h: tdx_context_segments): every rank logs the same sequence of segments between collectives - the protocol is
    rank-symmetric -, a call ends with a segment of kind 2, mode 2 (one rank on the device at a time) runs to the end, and the projection accepts the logs."""
    import torch

    from taudem_amd.distributed import StripGroup, StripPipeline, partition_rows, project_critical_path

    ny, nx, world = 300, 260, 3
    dem = oracle.synth_dem((ny, nx), 77)
    fel_o = oracle.pitremove(dem, -9999.0)
    parts = partition_rows(ny, world)
    with StripGroup(world, nx, [0] * world) as grp:
        def rank_main(r, c, comm):
            y0, y1 = parts[r]
            pipe = StripPipeline(c, comm, nx, y1 - y0)
            d = pipe.empty(torch.float32)
            d[1:y1 - y0 + 1] = torch.from_numpy(dem[y0:y1]).cuda()
            c.set_option("segment_trace", mode)
            fel, _ = pipe.pitremove(d, -9999.0)
            p, _, _ = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0)
            a, _ = pipe.aread8(p, -32768)
            seg = c.segments()
            c.set_option("segment_trace", 0)
            assert c.segments() == []
            return seg, fel[1:y1 - y0 + 1].cpu().numpy()
        res = grp.run(rank_main)
    logs = [r[0] for r in res]
    assert bits_equal(np.concatenate([r[1] for r in res], axis=0), fel_o)
    assert len(logs[0]) > 10 and all([s[:3] for s in lg] == [s[:3] for s in logs[0]] for lg in logs)
    stages = [s[0] for s in logs[0]]
    assert stages[0] == "pitremove" and "d8flowdir" in stages and stages[-1] == "aread8"
    assert [s[2] for s in logs[0]].count(2) == 3 and logs[0][-1][2] == 2          # three calls, each closed by an "end of call" segment
    assert all(s[3] >= 0.0 and s[4] >= 0.0 for lg in logs for s in lg)
    proj = project_critical_path(logs)
    assert set(proj["per_stage"]) == {"pitremove", "d8flowdir", "aread8"} and proj["total_ms"] > 0
