#!/usr/bin/env python3
"""fp64 3-D: per-kernel times of the per-axis path and of the fused variants (NDWT_FP64_FUSED / NDWT_VARIANT_* env)."""
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
dims, level, K = [n, n, n], 3, 5
plan = api.Plan(dims, [wname] * 3, torch.float64, False, True, "reference", max_level=level).set_variant_from_env()
nb = api.num_bands(3, level)
x = torch.randn(n, n, n, device="cuda", dtype=torch.float64)
y = torch.empty((nb, n, n, n), device="cuda", dtype=torch.float64)
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
torch.cuda.synchronize()
plan.set_profiling(True)
t0 = time.perf_counter()
for _ in range(K):
    plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
prof = {k: plan.get_profile(k) for k in range(4)}
V = n ** 3
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("NDWT_")}, "wname": wname, "path": plan.describe(),
                  "ms_per_step": round(dt * 1e3, 3), "roofline_frac": round(2 * level * 9 * V * 8 / dt / 8e12, 4),
                  "kernels_ms_per_launch": {k: (round(v[0] / max(v[1], 1), 4), v[1]) for k, v in prof.items() if v[1]},
                  "rt": float(torch.linalg.vector_norm(r - x) / torch.linalg.vector_norm(x))}))
