#!/usr/bin/env python3
"""cfg2 (4096^2 fp32, 3 levels): dec with the levels cascaded in one launch (Fwd2C) against one launch per level, interleaved A/B;
rows per wave swept.   python tools/bench2d_cascade.py [wname] [n]"""
import importlib
import os
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
ys = {}
plans = {}
pitches = {}
for name, (var, chunk, tb, pitch) in {"per-level": (9, 0, 0, 0), "cascade": (0, 0, 0, 0), "cascade/w1024": (0, 0, 1024, 0), "cascade/w2048": (0, 0, 2048, 0), "cascade/w3584": (0, 0, 3584, 0),
                               "cascade/24": (0, 24, 0, 0), "cascade/32": (0, 32, 0, 0), "cascade/44": (0, 44, 0, 0), "cascade/64": (0, 64, 0, 0), "cascade/96": (0, 96, 0, 0)}.items():
    pitches[name] = (n * n + pitch) if pitch else 0
    p = api.Plan([n, n], [wname] * 2, torch.float32, False, True, "reference", max_level=level)
    p.set_variant(fwd=var)
    p.set_tuning(tb, chunk)
    plans[name] = p
    ys[name] = torch.empty(nb * (n * n + 1024), device="cuda")
    p.dec(x.data_ptr(), ys[name].data_ptr(), level, s, band_pitch=pitches[name])
torch.cuda.synchronize()
ref = ys["per-level"]
def bands(name):
    pt = pitches[name] or n * n
    return torch.stack([ys[name][b * pt:b * pt + n * n] for b in range(nb)])
for name in plans:
    print(f"{name:14s} max |diff to per-level| = {float((bands(name) - bands('per-level')).abs().max()):.3e}")
import statistics
res = {k: [] for k in plans}
for r in range(6):                                        # batches of 20 back-to-back calls, the variants interleaved batch by batch
    for name, p in plans.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p.dec(x.data_ptr(), ys[name].data_ptr(), level, s, band_pitch=pitches[name])
        e0.record()
        for _ in range(20):
            p.dec(x.data_ptr(), ys[name].data_ptr(), level, s, band_pitch=pitches[name])
        e1.record()
        torch.cuda.synchronize()
        if r >= 1:
            res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name in plans:
    print(f"{name:14s} dec {wname} {n}^2 L{level}: median {statistics.median(res[name]):.1f} us  min {min(res[name]):.1f}  max {max(res[name]):.1f}")
