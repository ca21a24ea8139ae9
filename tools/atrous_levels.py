#!/usr/bin/env python3
"""a-trous mode, 3-D fp32: time of one analysis / synthesis level per tap stride (1, 2, 4) -- dec / rec at level L minus level L-1.
python tools/atrous_levels.py [n] [wname] [atrous|reference]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
mode = sys.argv[3] if len(sys.argv) > 3 else "atrous"
plan = api.Plan([n, n, n], [wname] * 3, torch.float32, False, True, mode, max_level=3).set_variant_from_env()
x = torch.randn(n, n, n, device="cuda")
y = torch.empty((api.num_bands(3, 3), n, n, n), device="cuda")
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
prev = (0.0, 0.0)
for level in (1, 2, 3):
    for _ in range(2):
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
    cur = (e[0].elapsed_time(e[1]) / 10, e[1].elapsed_time(e[2]) / 10)
    print(f"{n}^3 {wname} {mode} level {level} (tap stride {2 ** (level - 1) if mode != 'reference' else 1}): analysis {cur[0] - prev[0]:.3f} ms  synthesis {cur[1] - prev[1]:.3f} ms", flush=True)
    prev = cur
