#!/usr/bin/env python3
"""a-trous (dilated) mode: cfg3 and cfg2 shapes, dec+rec, 3 levels db4 fp32."""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
for dims in ([512, 512, 512], [256, 256, 256], [4096, 4096]):
    d, level = len(dims), 3
    for generic in (False, True):
        plan = api.Plan(dims, ["db4"] * d, torch.float32, False, True, "atrous", max_level=level).set_variant_from_env()
        plan.set_path(generic)
        nb = api.num_bands(d, level)
        shp = tuple(reversed(dims))
        x = torch.randn(*shp, device="cuda")
        y = torch.empty((nb,) + shp, device="cuda")
        r = torch.empty_like(x)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        torch.cuda.synchronize()
        K = 5
        t0 = time.perf_counter()
        for _ in range(K):
            plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        V = 1
        for n in dims:
            V *= n
        print(f"{dims} a-trous {'per-axis' if generic else 'fused on sub-lattices'}: {dt * 1e3:.3f} ms, {V / dt / 1e6:.0f} Mvox/s, frac "
              f"{2 * level * (1 + 2 ** d) * V * 4 / dt / 8e12:.3f}, rt {float(torch.linalg.vector_norm(r - x) / torch.linalg.vector_norm(x)):.2e}", flush=True)
        del plan, x, y, r
