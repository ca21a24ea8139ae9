#!/usr/bin/env python3
"""2-D throughput (BASELINE config 2: 4096x4096 fp32 db4, 3 levels) -- a parity-test config, reported for reference."""
import importlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n1 = n2 = 4096
level = 3
for generic, pitched in ((False, False), (False, True), (True, False)):
    plan = api.Plan([n1, n2], ["db4", "db4"], torch.float32, False, True, "reference", max_level=3).set_variant_from_env()
    plan.set_path(generic)
    bp = plan.band_pitch() if pitched else 0
    x = torch.randn(n2, n1, device="cuda")
    y = torch.empty(10 * (bp if bp else n1 * n2), device="cuda")
    r = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s, band_pitch=bp); plan.rec(y.data_ptr(), r.data_ptr(), level, s, band_pitch=bp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s, band_pitch=bp); plan.rec(y.data_ptr(), r.data_ptr(), level, s, band_pitch=bp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    V = n1 * n2
    print(json.dumps({"config": "2D fp32 4096x4096 db4 L3", "path": plan.describe() if not generic else "axis", "coefficients": "pitched" if pitched else "packed", "ms_per_step": round(dt * 1e3, 4),
                      "Mvox_s": round(V / dt / 1e6, 1), "roofline_frac": round(2 * level * 5 * V * 4 / dt / 8e12, 4),
                      "rt": float(torch.linalg.vector_norm(r - x) / torch.linalg.vector_norm(x))}))
