#!/usr/bin/env python3
"""dec+rec time of a 3-D fp32 volume for every Daubechies order (which kernels serve it is printed next to the time).
python tools/bench_wavelets.py [n] [orders, e.g. 4,6,7,8,9,10] [f32|f64]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
orders = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(1, 11))
dt = torch.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else torch.float32
level = 3
x = torch.randn(n, n, n, device="cuda", dtype=dt)
y = torch.empty((api.num_bands(3, level), n, n, n), device="cuda", dtype=dt)
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
for K in orders:
    for generic in (False, True):
        plan = api.Plan([n, n, n], [f"db{K}"] * 3, dt, False, True, "reference", max_level=level).set_variant_from_env()
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
        err = float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double()))
        print(f"db{K:<2d} {n}^3 {'fp64' if dt == torch.float64 else 'fp32'} L{level} {'per-axis' if generic else plan.describe():9s} dec {e[0].elapsed_time(e[1]) / 5:7.3f} ms  rec {e[1].elapsed_time(e[2]) / 5:7.3f} ms  "
              f"round trip {err:.1e}", flush=True)
