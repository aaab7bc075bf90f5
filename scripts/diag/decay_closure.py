"""Diagnosis (round 4): DinfDecayAccum -wg -o on DecayStrip-shaped inputs at several sizes - the GPU's evaluated set in -nc mode against the host closure of
the restatement, mismatch categories of the host check, and (where it finishes) the restatement itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import taudem_amd as T
import bench
from oracle import oracle as O

ND = np.float32(-3.402823466e38)
ctx = T.Context(0)
for nx, ny in ((4096, 512), (8192, 1024), (16384, 2048), (32768, 4096), (65536, 8192)):
    job = bench.DecayStrip(torch, ctx, None, nx, ny, 0, ny, 1234, T)
    ox, oy = job.outlets
    outl = (np.array(ox, dtype=np.int32), np.array(oy, dtype=np.int32) - 1)
    ang, dm, w = (t[1:ny + 1].cpu().numpy() for t in (job.ang, job.dm, job.w))
    mark = O.dinf_outlet_closure(ang, outl, dx=1.0, dy=1.0)
    for cc in (False, True):
        job.pipe.dinfdecayaccum(job.ang, job.dm, weights=job.w, outlets=job.outlets, contcheck=cc, out=job.out)
        torch.cuda.synchronize()
        out = job.out[1:ny + 1].cpu().numpy()
        have = out != ND
        bad, first, queued = O.dinfdecayaccum_check(ang, dm, out, dx=1.0, dy=1.0, weights=w, contcheck=cc, outlets=outl)
        line = {"nx": nx, "ny": ny, "contcheck": cc, "outlets": len(ox), "closure_host": int(mark.sum()), "gpu_has_value": int(have.sum()), "check_bad": bad}
        if not cc:
            miss = (mark != 0) & ~have
            extra = (mark == 0) & have
            line.update(in_host_closure_without_gpu_value=int(miss.sum()), gpu_value_outside_host_closure=int(extra.sum()))
            ys, xs = np.nonzero(miss)
            line["first_missing"] = [(int(x), int(y)) for x, y in zip(xs[:5], ys[:5])]
        if nx * ny <= 40_000_000:
            ref = O.dinfdecayaccum(ang, dm, dx=1.0, dy=1.0, weights=w, contcheck=cc, outlets=outl)
            neq = ref.view(np.uint32) != out.view(np.uint32)
            line["vs_restatement_diff"] = int(neq.sum())
            if neq.any():
                ys, xs = np.nonzero(neq)
                line["first_diff"] = [(int(x), int(y), float(ref[y, x]), float(out[y, x])) for x, y in zip(xs[:5], ys[:5])]
                line["diff_gpu_missing"] = int((neq & (out == ND)).sum()); line["diff_ref_missing"] = int((neq & (ref == ND)).sum())
        print(line, flush=True)
    del job
    torch.cuda.empty_cache()
