#!/usr/bin/env python3
"""dec+rec time of interleaved complex64 volumes for db4 .. db8, fused vs per-axis.  python tools/bench_complex_long.py [n] [orders]"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = 3
for K in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (4, 5, 6, 7, 8)):
    for generic in (False, True):
        plan = api.Plan([n, n, n], [f"db{K}"] * 3, torch.float32, True, True, "reference", max_level=level).set_variant_from_env()
        plan.set_path(generic)
        x = torch.randn(n, n, n, 2, device="cuda")
        y = torch.empty((api.num_bands(3, level), n, n, n, 2), device="cuda")
        r = torch.empty_like(x)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
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
        print(f"complex64 db{K} {n}^3 L{level} {'per-axis' if generic else plan.describe():9s} dec {e[0].elapsed_time(e[1]) / 5:7.3f} ms  "
              f"rec {e[1].elapsed_time(e[2]) / 5:7.3f} ms  round trip {err:.1e}", flush=True)
