"""Row strips across GPUs: one process per GPU, one horizontal strip (+ one halo row above and below)
per process - the layout of the reference's linearpart<T> (src/linearpart.h:133-162) - with the halo
exchange and the termination votes carried by torch.distributed (backend "nccl" = RCCL over xGMI on a
GPU node; "gloo" with host staging for CPU-side tests and for several ranks sharing one GPU).

The library itself never talks to a communication library: it calls back into :class:`StripComm`
through the ``tdx_comm`` struct of include/taudem_amd.h (exchange one boundary row with each strip
neighbour; all-reduce a few int64 on the host).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TdxStats, check


def partition_rows(ny: int, size: int):
    """Row ranges [y0, y1) per rank: ny // size rows each, the remainder to the last rank
    (src/linearpart.h:133-134)."""
    base = ny // size
    out = []
    for r in range(size):
        y0 = r * base
        y1 = (r + 1) * base if r < size - 1 else ny
        out.append((y0, y1))
    return out


class StripComm:
    """tdx_comm backed by a torch.distributed process group (ranks ordered north to south)."""

    def __init__(self, nx: int, device: int | None = 0, group=None):
        """device=None keeps the four exchange buffers in host memory (protocol tests without a GPU)."""
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.dev = torch.device(f"cuda:{device}") if device is not None else torch.device("cpu")
        self.capacity = 16 * int(nx)
        self._bufs = [torch.zeros(self.capacity, dtype=torch.uint8, device=self.dev) for _ in range(4)]   # send_up send_down recv_up recv_down
        self._host = None
        if self.backend != "nccl" and self.dev.type == "cuda":
            self._host = [torch.zeros(self.capacity, dtype=torch.uint8).pin_memory() for _ in range(4)]
        self.exchanges = 0
        self.allreduces = 0
        self._ex_cb = _lib.EXCHANGE_FN(self._exchange)
        self._ar_cb = _lib.ALLREDUCE_FN(self._allreduce)
        self.struct = _lib.TdxComm(self.rank, self.size, None, self._ex_cb, self._ar_cb, *[C.c_void_p(b.data_ptr()) for b in self._bufs], self.capacity)

    def ptr(self):
        return C.byref(self.struct)

    def _peer(self, delta):
        r = self.rank + delta
        if r < 0 or r >= self.size:
            return None
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def _exchange(self, user, nbytes):
        try:
            torch, dist = self.torch, self.dist
            n = int(nbytes)
            up, down = self._peer(-1), self._peer(+1)
            self.exchanges += 1
            if self.backend == "nccl":
                s_up, s_dn, r_up, r_dn = [b[:n] for b in self._bufs]
                ops = []
                if up is not None:
                    ops += [dist.P2POp(dist.isend, s_up, up, self.group), dist.P2POp(dist.irecv, r_up, up, self.group)]
                if down is not None:
                    ops += [dist.P2POp(dist.isend, s_dn, down, self.group), dist.P2POp(dist.irecv, r_dn, down, self.group)]
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                torch.cuda.synchronize(self.dev)
            else:   # gloo: host staging (or host buffers to begin with)
                staged = self._host is not None
                h = [x[:n] for x in (self._host if staged else self._bufs)]
                if staged:
                    if up is not None:
                        h[0].copy_(self._bufs[0][:n])
                    if down is not None:
                        h[1].copy_(self._bufs[1][:n])
                    torch.cuda.synchronize(self.dev)
                reqs = []
                if up is not None:
                    reqs += [dist.isend(h[0], up, self.group, tag=1), dist.irecv(h[2], up, self.group, tag=2)]
                if down is not None:
                    reqs += [dist.isend(h[1], down, self.group, tag=2), dist.irecv(h[3], down, self.group, tag=1)]
                for r in reqs:
                    r.wait()
                if staged:
                    if up is not None:
                        self._bufs[2][:n].copy_(h[2])
                    if down is not None:
                        self._bufs[3][:n].copy_(h[3])
                    torch.cuda.synchronize(self.dev)
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            import sys
            print(f"StripComm.exchange failed on rank {self.rank}: {e!r}", file=sys.stderr, flush=True)
            return 1

    def _allreduce(self, user, values, count, op):
        try:
            torch, dist = self.torch, self.dist
            self.allreduces += 1
            a = np.ctypeslib.as_array(values, shape=(int(count),))
            t = torch.from_numpy(a.copy())
            if self.backend == "nccl":
                t = t.to(self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX, group=self.group)
            a[:] = t.cpu().numpy()
            return 0
        except Exception as e:
            import sys
            print(f"StripComm.allreduce failed on rank {self.rank}: {e!r}", file=sys.stderr, flush=True)
            return 1


class RcclStripComm:
    """tdx_comm backed by the library's NATIVE RCCL transport (taudem_amd/csrc/comm.cpp): boundary rows travel as grouped
    ncclSend/ncclRecv on the context's stream, termination votes as ncclAllReduce on device values - no Python, no host
    staging and no extra synchronisation in the exchange path.  torch.distributed is only the bootstrap: rank 0's
    ncclUniqueId is broadcast through the process group (any backend) once."""

    def __init__(self, ctx, nx: int, group=None):
        import torch
        import torch.distributed as dist

        self.ctx = ctx
        self._lib = ctx._lib
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            check(self._lib.tdx_rccl_unique_id(ident))
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        check(self._lib.tdx_rccl_comm_create(ctx._h, ident, self.rank, self.size, int(nx), C.byref(h)), ctx._h)
        self._h = h
        self._comm = C.c_void_p(self._lib.tdx_rccl_comm_handle(h))
        self.backend = "rccl-native"
        del torch

    def ptr(self):
        return self._comm

    def _counters(self):
        e, a = C.c_int64(), C.c_int64()
        self._lib.tdx_rccl_comm_counters(self._h, C.byref(e), C.byref(a))
        return e.value, a.value

    @property
    def exchanges(self):
        return self._counters()[0]

    @property
    def allreduces(self):
        return self._counters()[1]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tdx_rccl_comm_destroy(self._h)
            self._h = None


class _GroupComm:
    """The tdx_comm of one rank of a StripGroup (owned by the group)."""

    def __init__(self, ptr, backend):
        self._ptr, self.backend = ptr, backend

    def ptr(self):
        return self._ptr


class StripGroup:
    """N row strips driven from ONE process: the library's own rank group (tdx_group_create, include/taudem_amd.h) - one context and
    one tdx_comm per rank; ranks that each have a device talk over RCCL, ranks that share a device over the in-process peer transport
    (host barriers + device copies).  `run(fn)` calls fn(rank, ctx, comm) on one thread per rank (the C calls release the GIL) and
    returns the results in rank order.  This is what `tool --gpus N` does in C++ (tool_strips.hpp); here it serves the tests and
    bench.py's functional many-strips-on-one-GPU runs."""

    def __init__(self, size: int, nx: int, devices=None):
        from .api import Context

        self._lib = _lib.load()
        self.size, self.nx = int(size), int(nx)
        devs = list(devices) if devices is not None else [0] * self.size
        arr = (C.c_int32 * self.size)(*devs)
        g = C.c_void_p()
        check(self._lib.tdx_group_create(self.size, arr, self.nx, C.byref(g)))
        self._g = g
        self.transport = self._lib.tdx_group_transport(g).decode()
        self.contexts = [Context.borrow(self._lib.tdx_group_context(g, r), devs[r]) for r in range(self.size)]
        self.comms = [_GroupComm(C.c_void_p(self._lib.tdx_group_comm(g, r)), self.transport) for r in range(self.size)]

    def run(self, fn):
        """fn(rank, context, comm) on one thread per rank; returns the list of results.  A rank that raises ABORTS THE GROUP (tdx_group_abort: the other
        ranks' pending and future collectives fail at once instead of waiting for TDX_COMM_TIMEOUT) - whatever it raised, communication error or
        not - and an aborted group is dead: its barrier and its RCCL communicators are gone for good.  run() on a dead group raises at once; build
        a new StripGroup to go on."""
        import threading

        if getattr(self, "_dead", False):
            raise RuntimeError("this StripGroup was aborted by a failed rank in an earlier run(): the rank group is one-shot after a failure - create a new StripGroup")
        out, err = [None] * self.size, []

        def main(r):
            try:
                out[r] = fn(r, self.contexts[r], self.comms[r])
            except BaseException as e:   # noqa: BLE001 - reported below
                import traceback
                err.append((r, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
                # the other ranks must not sit in their collectives until TDX_COMM_TIMEOUT: pending and future waits fail at once
                self._lib.tdx_group_abort(self._g)

        th = [threading.Thread(target=main, args=(r,), daemon=True) for r in range(self.size)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            self._dead = True
            raise RuntimeError("StripGroup rank(s) failed (the group is now dead - create a new one to continue):\n" + "\n".join(f"[rank {r}] {m}" for r, m in sorted(err)))
        return out

    def close(self):
        if getattr(self, "_g", None):
            for c in self.contexts:
                c.close()
            self._lib.tdx_group_destroy(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def project_critical_path(logs, exchange_us: float = 10.0, vote_us: float = 30.0, use: str = "wall"):
    """The critical path of an N-GPU strip run from the ranks' SEGMENT TRACES (Context.segments(), option "segment_trace" = 2: ranks that share a
    GPU take turns on it, so that every segment is timed as on a GPU of its own).  A strip run is, on every rank, the same sequence of
    segments of rank-local work separated by collectives (the outer loop of src/aread8.cpp:282-303 with share() / MPI_Allreduce,
    src/linearpart.h:313-384): a collective completes when the slowest rank arrives, so

        projected time = sum over segments of (max over ranks of the segment's time) + exchanges x exchange_us + all-reduces x vote_us.

    logs[r] = [(stage, phase, kind, device_ms, wall_ms), ...] of rank r.  Returns {"total_ms", "per_stage": {stage: {"ms", "work_ms", "latency_ms",
    "segments", "exchanges", "allreduces", "sum_over_ranks_ms", "phases": {phase: ms}}}, "assumed": {...}}.  A PROJECTION, not a measurement."""
    n = len(logs[0])
    for r, lg in enumerate(logs):
        if len(lg) != n or any(a[:3] != b[:3] for a, b in zip(lg, logs[0])):
            raise ValueError(f"rank {r} went through a different sequence of collectives than rank 0: the protocol is not rank-symmetric")
    col = 4 if use == "wall" else 3
    per, total = {}, 0.0
    for i in range(n):
        stage, phase, kind = logs[0][i][:3]
        work = max(lg[i][col] for lg in logs)
        lat = (exchange_us if kind == 0 else vote_us if kind == 1 else 0.0) * 1e-3
        st = per.setdefault(stage, {"ms": 0.0, "work_ms": 0.0, "latency_ms": 0.0, "segments": 0, "exchanges": 0, "allreduces": 0, "sum_over_ranks_ms": 0.0,
                                    "slowest_rank_histogram": [0] * len(logs), "phases": {}})
        st["ms"] += work + lat; st["work_ms"] += work; st["latency_ms"] += lat; st["segments"] += 1
        st["exchanges"] += kind == 0; st["allreduces"] += kind == 1
        st["sum_over_ranks_ms"] += sum(lg[i][col] for lg in logs)
        st["slowest_rank_histogram"][max(range(len(logs)), key=lambda r: logs[r][i][col])] += 1
        st["phases"][phase or "-"] = st["phases"].get(phase or "-", 0.0) + work + lat
        total += work + lat
    return {"total_ms": total, "per_stage": per,
            "assumed": {"exchange_us": exchange_us, "vote_us": vote_us, "segment_time": use,
                        "note": "projection from one-GPU segment traces (one rank on the device at a time), not an N-GPU measurement"}}


def _tptr(t, dtype, shape, name):
    import torch

    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: need a contiguous CUDA tensor {dtype} of shape {tuple(shape)}")
    return C.c_void_p(t.data_ptr())


class StripPipeline:
    """PitRemove -> D8FlowDir -> AreaD8 on one strip of a row-partitioned raster.  All rasters are CUDA
    tensors of shape (ny_local + 2, nx): row 0 / row -1 are the halo rows the library maintains."""

    def __init__(self, ctx, comm: StripComm | None, nx: int, ny_local: int):
        import torch

        self.ctx, self.comm, self.nx, self.ny_local = ctx, comm, int(nx), int(ny_local)
        self.torch = torch
        self.shape = (self.ny_local + 2, self.nx)
        self._cp = comm.ptr() if comm is not None else None

    def empty(self, dtype):
        return self.torch.empty(self.shape, dtype=dtype, device=f"cuda:{self.ctx.device}")

    def pitremove(self, dem, nodata=-9999.0, fourway=False, out=None):
        torch = self.torch
        fel = out if out is not None else self.empty(torch.float32)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_pitremove_strip(self.ctx._h, self._cp, _tptr(dem, torch.float32, self.shape, "dem"), self.nx, self.ny_local,
                                                float(nodata), None, int(bool(fourway)), _tptr(fel, torch.float32, self.shape, "fel"), C.byref(st)),
              self.ctx._h)
        return fel, st.as_dict()

    def d8flowdir(self, fel, nodata=-3.0e38, dx=1.0, dy=1.0, out=None):
        torch = self.torch
        rows = self.ny_local + 2
        dxc = np.ascontiguousarray(np.broadcast_to(np.asarray(dx, dtype=np.float64), (rows,)))
        dyc = np.ascontiguousarray(np.broadcast_to(np.asarray(dy, dtype=np.float64), (rows,)))
        p = out[0] if out is not None else self.empty(torch.int16)
        sd8 = out[1] if out is not None else self.empty(torch.float32)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_d8flowdir_strip(self.ctx._h, self._cp, _tptr(fel, torch.float32, self.shape, "fel"), self.nx, self.ny_local,
                                                float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data),
                                                _tptr(p, torch.int16, self.shape, "p"), _tptr(sd8, torch.float32, self.shape, "sd8"), C.byref(st)),
              self.ctx._h)
        return p, sd8, st.as_dict()

    @staticmethod
    def _outlets(outlets):
        """(columns, STRIP-ARRAY rows) -> ctypes pointers; None = no outlets."""
        if outlets is None:
            return None, None, -1, ()
        ox = np.ascontiguousarray(np.asarray(outlets[0], dtype=np.int32))
        oy = np.ascontiguousarray(np.asarray(outlets[1], dtype=np.int32))
        return C.c_void_p(ox.ctypes.data), C.c_void_p(oy.ctypes.data), int(ox.size), (ox, oy)

    def local_outlets(self, cols, global_rows, y0):
        """Global (column, row) outlet indices -> strip-array coordinates of the strip that starts at global row y0."""
        return np.asarray(cols, dtype=np.int32), (np.asarray(global_rows, dtype=np.int64) - int(y0) + 1).astype(np.int32)

    def aread8(self, p, nodata=-32768, weights=None, weights_nodata=-9999.0, contcheck=True, outlets=None, out=None):
        torch = self.torch
        ad8 = out if out is not None else self.empty(torch.float32)
        pw = _tptr(weights, torch.float32, self.shape, "weights") if weights is not None else None
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_aread8_strip(self.ctx._h, self._cp, _tptr(p, torch.int16, self.shape, "p"), self.nx, self.ny_local, int(nodata), pw,
                                             float(weights_nodata), int(bool(contcheck)), ox, oy, no, _tptr(ad8, torch.float32, self.shape, "ad8"),
                                             C.byref(st)), self.ctx._h)
        del keep
        return ad8, st.as_dict()

    def _cells(self, dx, dy):
        rows = self.ny_local + 2
        dxc = np.ascontiguousarray(np.broadcast_to(np.asarray(dx, dtype=np.float64), (rows,)))
        dyc = np.ascontiguousarray(np.broadcast_to(np.asarray(dy, dtype=np.float64), (rows,)))
        return dxc, dyc

    def dinfflowdir(self, fel, nodata=-3.0e38, dx=1.0, dy=1.0, out=None):
        torch = self.torch
        dxc, dyc = self._cells(dx, dy)
        ang = out[0] if out is not None else self.empty(torch.float32)
        slp = out[1] if out is not None else self.empty(torch.float32)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_dinfflowdir_strip(self.ctx._h, self._cp, _tptr(fel, torch.float32, self.shape, "fel"), self.nx, self.ny_local,
                                                  float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data),
                                                  _tptr(ang, torch.float32, self.shape, "ang"), _tptr(slp, torch.float32, self.shape, "slp"), C.byref(st)),
              self.ctx._h)
        return ang, slp, st.as_dict()

    def areadinf(self, ang, nodata=-3.402823466e38, dx=1.0, dy=1.0, weights=None, contcheck=True, outlets=None, out=None):
        torch = self.torch
        dxc, dyc = self._cells(dx, dy)
        sca = out if out is not None else self.empty(torch.float32)
        pw = _tptr(weights, torch.float32, self.shape, "weights") if weights is not None else None
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_areadinf_strip(self.ctx._h, self._cp, _tptr(ang, torch.float32, self.shape, "ang"), self.nx, self.ny_local, float(nodata),
                                               C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), pw, int(bool(contcheck)), ox, oy, no,
                                               _tptr(sca, torch.float32, self.shape, "sca"), C.byref(st)), self.ctx._h)
        del keep
        return sca, st.as_dict()

    def dinfupdependence(self, ang, dg, nodata=-3.402823466e38, dx=1.0, dy=1.0):
        """dep = depgrd(ang, dg) on this strip (src/DinfUpDependence.cpp:52): dg int32, dep float32 (nodata -1)."""
        torch = self.torch
        dxc, dyc = self._cells(dx, dy)
        dep = self.empty(torch.float32)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_dinfupdependence_strip(self.ctx._h, self._cp, _tptr(ang, torch.float32, self.shape, "ang"), self.nx, self.ny_local, float(nodata),
                                                       C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), _tptr(dg, torch.int32, self.shape, "dg"),
                                                       _tptr(dep, torch.float32, self.shape, "dep"), C.byref(st)), self.ctx._h)
        return dep, st.as_dict()

    def dinfrevaccum(self, ang, w, nodata=-3.402823466e38, w_nodata=-9999.0, dx=1.0, dy=1.0):
        """racc, dmax = dsaccum(ang, w) on this strip (src/DinfRevAccum.cpp:51)."""
        torch = self.torch
        dxc, dyc = self._cells(dx, dy)
        racc, dmax = self.empty(torch.float32), self.empty(torch.float32)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_dinfrevaccum_strip(self.ctx._h, self._cp, _tptr(ang, torch.float32, self.shape, "ang"), self.nx, self.ny_local, float(nodata),
                                                   C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data), _tptr(w, torch.float32, self.shape, "w"), float(w_nodata),
                                                   _tptr(racc, torch.float32, self.shape, "racc"), _tptr(dmax, torch.float32, self.shape, "dmax"), C.byref(st)), self.ctx._h)
        return racc, dmax, st.as_dict()

    def dinfdecayaccum(self, ang, dm, nodata=-3.402823466e38, dm_nodata=-9999.0, dx=1.0, dy=1.0, weights=None, contcheck=True, outlets=None, out=None):
        torch = self.torch
        dxc, dyc = self._cells(dx, dy)
        dsca = out if out is not None else self.empty(torch.float32)
        pw = _tptr(weights, torch.float32, self.shape, "weights") if weights is not None else None
        ox, oy, no, keep = self._outlets(outlets)
        st = TdxStats()
        torch.cuda.synchronize(self.ctx.device)
        check(self.ctx._lib.tdx_dinfdecayaccum_strip(self.ctx._h, self._cp, _tptr(ang, torch.float32, self.shape, "ang"), self.nx, self.ny_local,
                                                     float(nodata), C.c_void_p(dxc.ctypes.data), C.c_void_p(dyc.ctypes.data),
                                                     _tptr(dm, torch.float32, self.shape, "dm"), float(dm_nodata), pw, int(bool(contcheck)), ox, oy, no,
                                                     _tptr(dsca, torch.float32, self.shape, "dsca"), C.byref(st)), self.ctx._h)
        del keep
        return dsca, st.as_dict()
