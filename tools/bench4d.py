#!/usr/bin/env python3
"""4-D throughput on a cfg5-shaped volume (t-axis per-axis pass + fused 3-D kernels batched over t)."""
import importlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
dims = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (128, 128, 128, 32))]
level = 3
n1, n2, n3, n4 = dims
V = n1 * n2 * n3 * n4
plan = api.Plan(dims, ["db4"] * 4, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
nb = api.num_bands(4, level)
x = torch.randn(n4, n3, n2, n1, device="cuda")
y = torch.empty(nb, n4, n3, n2, n1, device="cuda")
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
torch.cuda.synchronize()
K = 5
plan.set_profiling(True)
t0 = time.perf_counter()
for _ in range(K):
    plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
prof = {k: plan.get_profile(k) for k in range(4)}
plan.set_profiling(False)
print(json.dumps({"config": f"4D fp32 {dims} db4 L3", "path": plan.describe(), "ms_per_step": round(dt * 1e3, 3), "Mvox_s": round(V / dt / 1e6, 1),
                  "roofline_frac": round(2 * level * 17 * V * 4 / dt / 8e12, 4),
                  "rt": float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double())),
                  # pres_l2_norm: the coefficient energy equals the signal energy (a band left unwritten would show here)
                  "energy_ratio": float(sum(float(torch.linalg.vector_norm(y[b].double())) ** 2 for b in range(nb)) ** 0.5
                                        / float(torch.linalg.vector_norm(x.double()))),
                  "coefficient_GiB": round(nb * V * 4 / 2 ** 30, 1),
                  # kinds: 0 fused analysis, 1 fused synthesis, 2 t-axis analysis, 3 t-axis synthesis: (ms per launch, launches)
                  "kernels": {k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in prof.items() if v[1]}}))
