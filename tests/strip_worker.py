"""Worker of the multi-rank strip tests (one process per rank, launched by torch.distributed.run).

GPU mode (default): every rank drives the HIP library on cuda:<LOCAL_RANK % device_count> - several
ranks may share one GPU (backend gloo, host-staged exchange) - runs PitRemove -> D8FlowDir -> AreaD8 on
its strip of a seeded raster and rank 0 compares the gathered strips bit-for-bit with the single-strip
run of the same library and with the CPU oracle.
--protocol: no GPU; exercises StripComm's exchange / all-reduce callbacks through the ctypes struct.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def protocol_test():
    import torch
    import torch.distributed as dist

    from taudem_amd.distributed import StripComm, partition_rows

    rank, size = dist.get_rank(), dist.get_world_size()
    nx = 37
    comm = StripComm(nx, device=None)
    assert comm.struct.rank == rank and comm.struct.size == size
    su, sd, ru, rd = comm._bufs
    for it in range(3):
        n = nx * (it + 1)
        su[:n] = torch.arange(n, dtype=torch.int32).add(100 * rank + it).to(torch.uint8)
        sd[:n] = torch.arange(n, dtype=torch.int32).add(100 * rank + 50 + it).to(torch.uint8)
        ru.fill_(255); rd.fill_(255)
        assert comm.struct.exchange(None, n) == 0
        if rank > 0:
            want = torch.arange(n, dtype=torch.int32).add(100 * (rank - 1) + 50 + it).to(torch.uint8)
            assert torch.equal(ru[:n], want), "recv_up must hold the upper neighbour's send_down"
        else:
            assert int(ru[:n].min()) == 255, "no upper neighbour: buffer untouched"
        if rank < size - 1:
            want = torch.arange(n, dtype=torch.int32).add(100 * (rank + 1) + it).to(torch.uint8)
            assert torch.equal(rd[:n], want), "recv_down must hold the lower neighbour's send_up"
        else:
            assert int(rd[:n].min()) == 255
    vals = (C.c_int64 * 3)(rank + 1, 10 * (rank + 1), -rank)
    assert comm.struct.allreduce(None, vals, 3, 0) == 0
    assert list(vals) == [sum(r + 1 for r in range(size)), sum(10 * (r + 1) for r in range(size)), -sum(range(size))]
    vals = (C.c_int64 * 2)(rank, 7 - rank)
    assert comm.struct.allreduce(None, vals, 2, 1) == 0
    assert list(vals) == [size - 1, 7]
    parts = partition_rows(103, size)
    assert parts[0][0] == 0 and parts[-1][1] == 103 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert all(y1 - y0 == 103 // size for y0, y1 in parts[:-1])
    assert comm.exchanges == 3 and comm.allreduces == 2


def make_case(name, O):
    if name == "plain":
        return O.synth_dem((333, 200), 5), {}
    if name == "short_strips":       # strips shorter than a tile, width not a multiple of 64
        return O.synth_dem((130, 257), 9), {}
    if name == "holes":
        dem = O.synth_dem((400, 300), 21)
        yy, xx = np.mgrid[0:400, 0:300]
        dem[(yy - 200) ** 2 + (xx - 150) ** 2 < 1600] = -9999.0
        dem[:3, :] = -9999.0
        return dem, {}
    if name == "big":                # forces the exact re-evaluation path across strip boundaries
        return O.synth_dem((300, 260), 13), {"TDX_AD8_BIG_THRESHOLD": "40"}
    if name == "wide":
        return O.synth_dem((1000, 1400), 3), {}
    raise ValueError(name)


def gpu_test(case):
    import torch
    import torch.distributed as dist

    import taudem_amd as T
    from oracle import oracle as O
    from taudem_amd.distributed import StripComm, StripPipeline, partition_rows

    rank, size = dist.get_rank(), dist.get_world_size()
    ndev = torch.cuda.device_count()
    device = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    torch.cuda.set_device(device)
    dinf = case.endswith("+dinf")
    dem, env = make_case(case.replace("+dinf", ""), O)
    os.environ.update(env)
    ny, nx = dem.shape
    y0, y1 = partition_rows(ny, size)[rank]
    nyl = y1 - y0
    ctx = T.Context(device)
    comm = StripComm(nx, device=device)
    pipe = StripPipeline(ctx, comm, nx, nyl)
    d_dem = pipe.empty(torch.float32)
    d_dem.fill_(12345.0)   # halo rows are the library's business: poison them
    d_dem[1:nyl + 1] = torch.from_numpy(dem[y0:y1]).to(d_dem.device)
    fel, st1 = pipe.pitremove(d_dem, -9999.0)
    p, sd8, st2 = pipe.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    outs = {"fel": fel, "p": p, "sd8": sd8}
    for cc in (True, False):
        a, st3 = pipe.aread8(p, -32768, contcheck=cc)
        outs[f"ad8_{int(cc)}"] = a.clone()
    # weights (exact pull walk across strips) and outlets (upstream closure across strips)
    rng0 = np.random.default_rng(23)
    wd8 = (rng0.random(dem.shape, dtype=np.float32) * 1.0e4).astype(np.float32)
    o_rows = rng0.integers(2, ny - 2, size=6)
    o_cols = rng0.integers(2, nx - 2, size=6)
    d_wd8 = pipe.empty(torch.float32); d_wd8.zero_(); d_wd8[1:nyl + 1] = torch.from_numpy(wd8[y0:y1]).to(d_wd8.device)
    lo = pipe.local_outlets(o_cols, o_rows, y0)
    outs["ad8_w_nc"] = pipe.aread8(p, -32768, weights=d_wd8, contcheck=False)[0].clone()
    outs["ad8_o"] = pipe.aread8(p, -32768, contcheck=True, outlets=lo)[0].clone()
    outs["ad8_w_o_nc"] = pipe.aread8(p, -32768, weights=d_wd8, contcheck=False, outlets=lo)[0].clone()
    if dinf:
        rng = np.random.default_rng(17)
        wgt = (rng.random(dem.shape, dtype=np.float32) * 3.0).astype(np.float32)
        dmf = (0.9 + 0.1 * rng.random(dem.shape, dtype=np.float32)).astype(np.float32)
        d_w = pipe.empty(torch.float32); d_w.zero_(); d_w[1:nyl + 1] = torch.from_numpy(wgt[y0:y1]).to(d_w.device)
        d_dm = pipe.empty(torch.float32); d_dm.zero_(); d_dm[1:nyl + 1] = torch.from_numpy(dmf[y0:y1]).to(d_dm.device)
        ang, slp, std = pipe.dinfflowdir(fel, -3.0e38, 30.0, 20.0)
        outs["ang"], outs["slp"] = ang, slp
        outs["sca"] = pipe.areadinf(ang, dx=30.0, dy=20.0, contcheck=True)[0].clone()
        outs["sca_w_nc"], sta = pipe.areadinf(ang, dx=30.0, dy=20.0, weights=d_w, contcheck=False)
        outs["sca_w_nc"] = outs["sca_w_nc"].clone()
        outs["dsca_nc"] = pipe.dinfdecayaccum(ang, d_dm, dx=30.0, dy=20.0, contcheck=False)[0].clone()
        outs["sca_o_nc"] = pipe.areadinf(ang, dx=30.0, dy=20.0, contcheck=False, outlets=lo)[0].clone()
        outs["dsca_w_o_nc"] = pipe.dinfdecayaccum(ang, d_dm, dx=30.0, dy=20.0, weights=d_w, contcheck=False, outlets=lo)[0].clone()
    gathered = {}
    for k, t in outs.items():
        mine = t[1:nyl + 1].cpu().contiguous()
        if rank == 0:
            parts = [mine]
            for r in range(1, size):
                yy0, yy1 = partition_rows(ny, size)[r]
                buf = torch.empty((yy1 - yy0, nx), dtype=mine.dtype)
                dist.recv(buf, r)
                parts.append(buf)
            gathered[k] = torch.cat(parts, 0).numpy()
        else:
            dist.send(mine, 0)
    if rank == 0:
        def same(a, b):
            if a.dtype == np.float32:
                return np.array_equal(a.view(np.uint32), b.view(np.uint32))
            return np.array_equal(a, b)
        fel_o = O.pitremove(dem, -9999.0)
        p_o, sd8_o, sto = O.d8flowdir(fel_o, -3.0e38, 30.0, 30.0)
        assert same(gathered["fel"], fel_o), f"{case}: fel differs from the oracle ({(gathered['fel'] != fel_o).sum()} cells)"
        assert same(gathered["sd8"], sd8_o), f"{case}: sd8 differs"
        assert same(gathered["p"], p_o), f"{case}: p differs from the oracle ({(gathered['p'] != p_o).sum()} cells)"
        assert st2["flats_initial"] == sto["flats_initial"] and st2["flat_iterations"] == sto["flat_iterations"], (st2, sto)
        for cc in (True, False):
            a_o = O.aread8(p_o, -32768, contcheck=cc)
            g = gathered[f"ad8_{int(cc)}"]
            assert same(g, a_o), f"{case}: ad8 contcheck={cc} differs from the oracle ({(g != a_o).sum()} cells)"
        og = (o_cols.astype(np.int32), o_rows.astype(np.int32))
        for key, kw in (("ad8_w_nc", dict(weights=wd8, contcheck=False)), ("ad8_o", dict(contcheck=True, outlets=og)),
                        ("ad8_w_o_nc", dict(weights=wd8, contcheck=False, outlets=og))):
            a_o = O.aread8(p_o, -32768, weights_nodata=-9999.0, **kw)
            assert same(gathered[key], a_o), f"{case}: {key} differs from the oracle ({(gathered[key] != a_o).sum()} cells)"
        if dinf:
            ang_o, slp_o, _ = O.dinfflowdir(fel_o, -3.0e38, 30.0, 20.0)
            assert same(gathered["ang"], ang_o), f"{case}: ang differs from the oracle ({(gathered['ang'] != ang_o).sum()} cells)"
            assert same(gathered["slp"], slp_o), f"{case}: slp differs"
            def close(a, b, name):   # gate of the north star: 1e-6 relative on D-infinity areas (observed: identical bits)
                ok = (a == b) | (np.abs(a - b) <= 1e-6 * np.abs(b))
                assert ok.all(), f"{case}: {name}: {(~ok).sum()} cells beyond 1e-6 relative"
            close(gathered["sca"], O.areadinf(ang_o, dx=30.0, dy=20.0, contcheck=True), "sca")
            close(gathered["sca_w_nc"], O.areadinf(ang_o, dx=30.0, dy=20.0, weights=wgt, contcheck=False), "sca_w_nc")
            close(gathered["dsca_nc"], O.dinfdecayaccum(ang_o, dmf, dx=30.0, dy=20.0, contcheck=False), "dsca_nc")
            close(gathered["sca_o_nc"], O.areadinf(ang_o, dx=30.0, dy=20.0, contcheck=False, outlets=og), "sca_o_nc")
            close(gathered["dsca_w_o_nc"], O.dinfdecayaccum(ang_o, dmf, dx=30.0, dy=20.0, weights=wgt, contcheck=False, outlets=og), "dsca_w_o_nc")
        print(f"strip_worker {case}: {size} ranks bit-exact vs oracle; pit outer rounds {st1['cells_evaluated']}, "
              f"flat iterations {st2['flat_iterations']}, ad8 outer rounds {st3['rounds']}, exchanges {comm.exchanges}, allreduces {comm.allreduces}",
              flush=True)
    dist.barrier()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--protocol", action="store_true")
    ap.add_argument("--case", default="plain")
    ap.add_argument("--backend", default="gloo")
    args = ap.parse_args()
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(args.backend)
    try:
        if args.protocol:
            protocol_test()
        else:
            gpu_test(args.case)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
