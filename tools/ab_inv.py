#!/usr/bin/env python3
"""Interleaved A/B of the float synthesis kernel variants (NDWT_VARIANT_INV values) on the cfg3 volume, one level per launch:
every variant gets its own plan (the variant is read at plan creation), timed round-robin in one process.  Each variant's
reconstruction of the same coefficients is compared with variant 0's.
python tools/ab_inv.py 0,5,6,7 [wname] [n]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
variants = [int(v) for v in sys.argv[1].split(",")]
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
level = 3
plans = {}
for v in variants:
    os.environ["NDWT_VARIANT_INV"] = str(v)
    plans[v] = api.Plan([n, n, n], [wname] * 3, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
nb = api.num_bands(3, level)
y = torch.randn((nb, n, n, n), device="cuda")
s = torch.cuda.current_stream().cuda_stream
outs = {v: torch.empty(n, n, n, device="cuda") for v in variants}
for v in variants:
    plans[v].rec(y.data_ptr(), outs[v].data_ptr(), level, s)
torch.cuda.synchronize()
ref = outs[variants[0]]
for v in variants[1:]:
    print(f"variant {v}: max |diff to variant {variants[0]}| / max|ref| = {float((outs[v] - ref).abs().max() / ref.abs().max()):.2e}")
tot = {v: 0.0 for v in variants}
reps = 15
for r in range(reps + 2):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plans[v].rec(y.data_ptr(), outs[v].data_ptr(), level, s)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            tot[v] += e0.elapsed_time(e1)
print("rec", wname, n, {v: round(t / reps / level, 4) for v, t in tot.items()}, "ms per level")
