#!/usr/bin/env python3
"""a few ndwt_denoise calls on one volume (profiling target): python tools/run_denoise.py [n] [wname] [level] [reps]
NDWT_FUSED_LEVEL1=0|1|2 selects Plan.set_fused_level1 (0: level-1 detail bands in memory, 2: fused level 1 for 8 taps too)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
plan = api.Plan([n, n, n], [wname] * 3, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
plan.set_fused_level1(int(os.environ.get("NDWT_FUSED_LEVEL1", "1")))
x = torch.randn(n, n, n, device="cuda")
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    plan.denoise(x.data_ptr(), r.data_ptr(), level, 0.5, False, s)
torch.cuda.synchronize()
print("ok", float(r.abs().max()))
