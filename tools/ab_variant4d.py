#!/usr/bin/env python3
"""Interleaved A/B of synthesis variants on cfg5 (256^3 x 32 fp32 db4, 3 levels): rec only, one shared coefficient array.
python tools/ab_variant4d.py 0,9 [reps]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
variants = [int(v) for v in sys.argv[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dims, level = [256, 256, 256, 32], 3
plans = {v: api.Plan(dims, ["db4"] * 4, torch.float32, False, True, "reference", max_level=level).set_variant(inv=v) for v in variants}
nb = api.num_bands(4, level)
y = torch.empty((nb, 32, 256, 256, 256), device="cuda")
for b in range(nb):
    y[b].normal_()
r = torch.empty((32, 256, 256, 256), device="cuda")
s = torch.cuda.current_stream().cuda_stream
tot = {v: 0.0 for v in variants}
for k in range(reps + 1):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plans[v].rec(y.data_ptr(), r.data_ptr(), level, s)
        e1.record()
        torch.cuda.synchronize()
        if k >= 1:
            tot[v] += e0.elapsed_time(e1)
print("cfg5 rec", {v: round(tot[v] / reps, 3) for v in variants}, "ms")
