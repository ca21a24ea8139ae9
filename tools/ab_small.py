#!/usr/bin/env python3
"""Chunk-size sweep on small problems (serial march length vs number of workgroups)."""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
import os
DIMS = [[int(v) for v in os.environ["DIMS"].split("x")]] if "DIMS" in os.environ else ([256, 256], [64, 64, 64], [128, 128, 128])
ZCS = [int(v) for v in os.environ["ZCS"].split(",")] if "ZCS" in os.environ else (0, 2, 3, 4, 5, 7, 10, 14)
for dims in DIMS:
    d, level = len(dims), 3
    plan = api.Plan(dims, ["db4"] * d, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
    nb = api.num_bands(d, level)
    shp = tuple(reversed(dims))
    x = torch.randn(*shp, device="cuda")
    y = torch.empty((nb,) + shp, device="cuda")
    r = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for zc in ZCS:
        plan.set_tuning(0, zc)
        for _ in range(20):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        torch.cuda.synchronize()
        K = int(os.environ.get("REPS", "200"))
        t0 = time.perf_counter()
        for _ in range(K):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        torch.cuda.synchronize()
        out[zc] = round(1e6 * (time.perf_counter() - t0) / K, 1)
    print(dims, out, "us per dec+rec by chunk (0 = library choice)", flush=True)
