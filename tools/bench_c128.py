#!/usr/bin/env python3
"""dec+rec time of a 256^3 complex128 volume (the reference test scripts' input type at the mex path's precision) for db4 .. db7, fused
where instantiated against the per-axis path.  python tools/bench_c128.py [n]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = 3
x = torch.randn(n, n, n, 2, device="cuda", dtype=torch.float64)
y = torch.empty((api.num_bands(3, level), n, n, n, 2), device="cuda", dtype=torch.float64)
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for K in (4, 5, 6, 7):
    for generic in (False, True):
        plan = api.Plan([n, n, n], [f"db{K}"] * 3, torch.float64, True, True, "reference", max_level=level)
        plan.set_path(generic)
        for _ in range(2):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s)
            plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(5):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s)
        e[1].record()
        for _ in range(5):
            plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        e[2].record()
        torch.cuda.synchronize()
        err = float(torch.linalg.vector_norm(r - x) / torch.linalg.vector_norm(x))
        print(f"db{K:<2d} {n}^3 complex128 L{level} {'per-axis' if generic else plan.describe():34s} dec {e[0].elapsed_time(e[1]) / 5:7.3f} ms  "
              f"rec {e[1].elapsed_time(e[2]) / 5:7.3f} ms  round trip {err:.1e}", flush=True)
