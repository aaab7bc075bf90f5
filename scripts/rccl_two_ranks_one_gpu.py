"""Probe: can the native RCCL transport run two ranks on ONE GPU?  (RCCL normally refuses duplicate devices; informational.)
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/rccl_two_ranks_one_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import taudem_amd as T
from taudem_amd.distributed import RcclStripComm

dist.init_process_group("gloo")
ctx = T.Context(0)
try:
    c = RcclStripComm(ctx, 1024)
    print("rank", dist.get_rank(), "native RCCL communicator created on a shared GPU", flush=True)
except Exception as e:   # noqa: BLE001
    print("rank", dist.get_rank(), "refused:", e, flush=True)
