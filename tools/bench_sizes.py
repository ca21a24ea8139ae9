#!/usr/bin/env python3
"""dec / rec per-launch times of one data kind over several cubic sizes (A/B of library variants: NDWT_LIB_VARIANT).
python tools/bench_sizes.py <float32|float64> <real|complex> n|n1xn2xn3 ..."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
dt = torch.float32 if sys.argv[1] == "float32" else torch.float64
cplx = sys.argv[2] == "complex"
level = 3
for arg in sys.argv[3:]:
    dims = [int(v) for v in arg.split("x")] if "x" in arg else [int(arg)] * 3          # n or n1xn2xn3
    n = "x".join(str(v) for v in dims)
    plan = api.Plan(dims, ["db4"] * 3, dt, cplx, True, "reference", max_level=level).set_variant_from_env()
    shp = tuple(reversed(dims)) + ((2,) if cplx else ())
    x = torch.randn(*shp, device="cuda", dtype=dt)
    y = torch.empty((api.num_bands(3, level),) + shp, device="cuda", dtype=dt)
    r = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    torch.cuda.synchronize()
    plan.set_profiling(True)
    for _ in range(5):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    torch.cuda.synchronize()
    a, b = plan.get_profile(0), plan.get_profile(1)
    print(f"{sys.argv[1]} {sys.argv[2]} {n}: analysis {a[0] / max(a[1], 1):.4f} ms  synthesis {b[0] / max(b[1], 1):.4f} ms per launch", flush=True)
    del x, y, r, plan
    torch.cuda.empty_cache()
