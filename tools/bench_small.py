#!/usr/bin/env python3
"""Small problems (launch-bound): time per dec+rec through the C ABI, back-to-back calls on one stream."""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
for dims in ([4096], [256, 256], [512, 512], [64, 64, 64], [128, 128, 128], [32, 32, 16, 16]):
    d, level = len(dims), 3
    plan = api.Plan(dims, ["db4"] * d, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
    nb = api.num_bands(d, level)
    shp = tuple(reversed(dims))
    x = torch.randn(*shp, device="cuda")
    y = torch.empty((nb,) + shp, device="cuda")
    r = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    V = 1
    for n in dims:
        V *= n
    print(f"{dims} {plan.describe()}: host enqueue {1e6 * (t1 - t0) / K:.1f} us, total {1e6 * (t2 - t0) / K:.1f} us per dec+rec "
          f"({V / ((t2 - t0) / K) / 1e6:.0f} Mvox/s)", flush=True)
