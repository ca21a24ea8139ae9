#!/usr/bin/env python3
"""cfg2 (4096^2 fp32, 3 levels): dec / rec with the levels cascaded in one launch (Fwd2C / Inv2C) against one launch per level; the
variants interleaved in batches of 20 back-to-back calls.   python tools/bench2d_cascade.py [wname] [n]"""
import importlib
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

api = importlib.import_module("non-decimated_wavelets_amd.api")
wname = sys.argv[1] if len(sys.argv) > 1 else "db4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
level = 3
x = torch.randn(n, n, device="cuda")
nb = api.num_bands(2, level)
s = torch.cuda.current_stream().cuda_stream
y = torch.randn((nb, n, n), device="cuda")
plans, outs = {}, {}
# name: (variant_fwd, variant_inv, rows per wave, waves)
for name, (vf, vi, chunk, tb) in {"per-level": (9, 9, 0, 0), "cascade/w2048": (11, 11, 0, 2048), "cascade/w512": (11, 11, 0, 512), "cascade/w768": (11, 11, 0, 768),
                                   "cascade/w896": (11, 11, 0, 896), "cascade/w1024": (11, 11, 0, 1024), "cascade/w1152": (11, 11, 0, 1152), "cascade/w1280": (11, 11, 0, 1280),
                                   "cascade/d2/w896": (11, 12, 0, 896), "cascade/d2/w1024": (11, 12, 0, 1024)}.items():
    p = api.Plan([n, n], [wname] * 2, torch.float32, False, True, "reference", max_level=level)
    p.set_variant(fwd=vf, inv=vi)
    p.set_tuning(tb, chunk)
    plans[name] = p
    outs[name] = (torch.empty((nb, n, n), device="cuda"), torch.empty(n, n, device="cuda"))
    p.dec(x.data_ptr(), outs[name][0].data_ptr(), level, s)
    p.rec(y.data_ptr(), outs[name][1].data_ptr(), level, s)
torch.cuda.synchronize()
for name in plans:
    print(f"{name:18s} max |diff to per-level|: dec {float((outs[name][0] - outs['per-level'][0]).abs().max()):.2e}  "
          f"rec {float((outs[name][1] - outs['per-level'][1]).abs().max()):.2e} (max |rec| {float(outs['per-level'][1].abs().max()):.1f})")
res = {k: ([], []) for k in plans}
for r in range(6):
    for name, p in plans.items():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        p.dec(x.data_ptr(), outs[name][0].data_ptr(), level, s)
        ev[0].record()
        for _ in range(20):
            p.dec(x.data_ptr(), outs[name][0].data_ptr(), level, s)
        ev[1].record()
        p.rec(y.data_ptr(), outs[name][1].data_ptr(), level, s)
        ev[2].record()
        for _ in range(20):
            p.rec(y.data_ptr(), outs[name][1].data_ptr(), level, s)
        ev[3].record()
        torch.cuda.synchronize()
        if r >= 1:
            res[name][0].append(ev[0].elapsed_time(ev[1]) / 20 * 1e3)
            res[name][1].append(ev[2].elapsed_time(ev[3]) / 20 * 1e3)
for name in plans:
    print(f"{name:18s} {wname} {n}^2 L{level}: dec {statistics.median(res[name][0]):7.1f} us   rec {statistics.median(res[name][1]):7.1f} us")
