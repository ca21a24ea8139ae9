#!/usr/bin/env python3
"""One rank's share of an 8-GPU run on ONE GPU (world size 1: every exchange is a local copy), piece by piece.

    python tools/slab_pieces.py [wname level n_local]      (default db4 3 64; the volume is 512 x 512 x n_local)

Prints ms per dec+rec with the exchange overlapped (pieces) and not (one launch per level and direction), and the HIP-event
time of every engine call of one step (which launch costs what)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sh = importlib.import_module("non-decimated_wavelets_amd.sharded")

wname = sys.argv[1] if len(sys.argv) > 1 else "db4"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nloc = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda", 0)
x = torch.randn(nloc, 512, 512, device=dev)
import gc  # noqa: E402
gc.collect()
gc.freeze()


def timed(eng, steps=200):
    for _ in range(5):
        r = eng.rec(eng.dec(x, level))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.rec(eng.dec(x, level))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"   (host: {(t1 - t0) / steps * 1e3:.3f} ms per step to enqueue)")
    return (time.perf_counter() - t0) / steps * 1e3, r


for overlap, ts in ((False, False), (True, False), (True, True)):
    eng = sh.ShardedNdDwt([wname] * 3, [512, 512, nloc], pres_l2_norm=True, precision="single", device=dev, overlap=overlap, two_streams=ts)
    ms, r = timed(eng)
    err = float((r - x).norm() / x.norm())
    print(f"{wname} L{level} 512x512x{nloc} overlap={overlap} two_streams={ts}: {ms:.4f} ms per dec+rec (round trip {err:.2e})")
    # per-call events of one step
    e = eng.engine
    log = []
    for name in ("analysis", "analysis_split", "analysis_run", "analysis_ends", "synthesis_ext", "synthesis_part", "synthesis_send_parts"):
        fn = getattr(e, name, None)
        if fn is None:
            continue

        def wrap(fn=fn, name=name):
            def g(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **k)
                e1.record()
                log.append((name, e0, e1))
                return out
            return g
        setattr(e, name, wrap())
    for rep in range(3):
        log.clear()
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        r = eng.rec(eng.dec(x, level))
        t1 = torch.cuda.Event(enable_timing=True)
        t1.record()
        torch.cuda.synchronize()
    print("   one step %.4f ms:" % t0.elapsed_time(t1), "  ".join(f"{n}={a.elapsed_time(b) * 1e3:.0f}us" for n, a, b in log))
