#!/usr/bin/env python3
"""Interleaved A/B of the row-chunk size of the fused 2-D kernels on cfg2 (4096x4096 fp32 db4 L3).
python tools/ab_2d.py ychunk[,ychunk...] [n]   (0 = the library's choice; NDWT_VARIANT_INV: 1 Inv2S, 0 / 4 Inv2P depth 2 / 4,
6 / 8 the same on half as many waves)"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
ycs = [int(v) for v in sys.argv[1].split(",")]
n1 = n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
level = 3
plan = api.Plan([n1, n2], ["db4", "db4"], torch.float32, False, True, "reference", max_level=3).set_variant_from_env()
x = torch.randn(n2, n1, device="cuda")
y = torch.empty(10, n2, n1, device="cuda")
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
tot = {(w, z): 0.0 for z in ycs for w in ("dec", "rec")}
reps = 30
for k in range(reps + 3):
    for z in ycs:
        plan.set_tuning(0, z)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        plan.dec(x.data_ptr(), y.data_ptr(), level, s)
        ev[1].record()
        plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        ev[2].record()
        torch.cuda.synchronize()
        if k >= 3:
            tot[("dec", z)] += ev[0].elapsed_time(ev[1])
            tot[("rec", z)] += ev[1].elapsed_time(ev[2])
for w in ("dec", "rec"):
    print(w, {z: round(tot[(w, z)] / reps, 4) for z in ycs}, "ms per 3 levels")
