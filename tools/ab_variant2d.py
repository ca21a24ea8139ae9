#!/usr/bin/env python3
"""Interleaved A/B of synthesis variants (Plan.set_variant(inv=..)) of the fused 2-D path: 3 levels of rec on an n x n fp32 image, timed
round-robin in one process; results compared with the first variant's.  python tools/ab_variant2d.py 0,7 [wname] [n]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
variants = [int(v) for v in sys.argv[1].split(",")]
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
level = 3
plans = {v: api.Plan([n, n], [wname] * 2, torch.float32, False, True, "reference", max_level=level).set_variant(inv=v) for v in variants}
y = torch.randn(api.num_bands(2, level), n, n, device="cuda")
r = {v: torch.empty(n, n, device="cuda") for v in variants}
s = torch.cuda.current_stream().cuda_stream
for v in variants:
    plans[v].rec(y.data_ptr(), r[v].data_ptr(), level, s)
torch.cuda.synchronize()
for v in variants[1:]:
    print(f"variant {v}: max |diff to variant {variants[0]}| = {float((r[v] - r[variants[0]]).abs().max()):.2e} (max |x| {float(r[variants[0]].abs().max()):.2f})")
tot = {v: 0.0 for v in variants}
reps = 40
for k in range(reps + 3):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plans[v].rec(y.data_ptr(), r[v].data_ptr(), level, s)
        e1.record()
        torch.cuda.synchronize()
        if k >= 3:
            tot[v] += e0.elapsed_time(e1)
print(f"rec {wname} {n}^2 L{level}", {v: round(tot[v] / reps / level * 1e3, 2) for v in variants}, "us per level")
