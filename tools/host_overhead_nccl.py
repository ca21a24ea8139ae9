#!/usr/bin/env python3
"""Host-side cost of one sharded dec + rec step when every exchange is a real RCCL batch (1-rank `nccl` group, self-segments through
grouped send / receive: the test hook `_self_p2p`): time to ENQUEUE a step (the host running ahead of the GPU) against the time the GPU
needs for it.  If enqueueing takes longer than executing, an 8-GPU run is bound by the Python driver, not by the kernels.
    python tools/host_overhead_nccl.py [wname level]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
wname = sys.argv[1] if len(sys.argv) > 1 else "db4"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = torch.randn(64, 512, 512, device=dev)
import gc  # noqa: E402
gc.collect()
gc.freeze()          # (a full collection of the interpreter's garbage collector takes 38 ms with torch loaded: not inside a 100-step loop)
for p2p, transport in ((False, "torch"), (True, "torch"), (True, "rccl")):
    for overlap in (False, True):
        eng = sh.ShardedNdDwt([wname] * 3, [512, 512, 64], pres_l2_norm=True, precision="single", device=dev, overlap=overlap, _self_p2p=p2p,
                              transport=transport, two_streams=(overlap and transport == "rccl"))
        for _ in range(5):
            r = eng.rec(eng.dec(x, level))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            r = eng.rec(eng.dec(x, level))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{wname} L{level} exchange={('RCCL send/recv to self' if transport == 'torch' else 'direct RCCL to self') if p2p else 'local copies'} overlap={overlap}: enqueue {(t1 - t0) * 10:.3f} ms per step, "
              f"complete {(t2 - t0) * 10:.3f} ms per step", flush=True)
dist.destroy_process_group()
