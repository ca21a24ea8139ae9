#!/usr/bin/env python3
"""dec+rec time of an n x n fp32 image for db6 .. db10, fused vs per-axis.  python tools/bench2d_long.py [n]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
level = 3
x = torch.randn(n, n, device="cuda")
y = torch.empty(api.num_bands(2, level), n, n, device="cuda")
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for K in (6, 7, 8, 9, 10):
    for generic in (False, True):
        plan = api.Plan([n, n], [f"db{K}"] * 2, torch.float32, False, True, "reference", max_level=level)
        plan.set_path(generic)
        for _ in range(3):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s)
            plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(10):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s)
        e[1].record()
        for _ in range(10):
            plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        e[2].record()
        torch.cuda.synchronize()
        err = float(torch.linalg.vector_norm(r - x) / torch.linalg.vector_norm(x))
        print(f"db{K:<2d} {n}^2 fp32 L{level} {'per-axis' if generic else plan.describe():9s} dec {e[0].elapsed_time(e[1]) / 10:7.3f} ms  "
              f"rec {e[1].elapsed_time(e[2]) / 10:7.3f} ms  round trip {err:.1e}", flush=True)
