#!/usr/bin/env python3
"""Large 1-D and 2-D cases of the non-headline kinds: 1-D fp32/fp64/complex (lane-shift kernel on the contiguous axis),
2-D complex."""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
for dims, dtype, cplx in (([1 << 24], torch.float32, False), ([1 << 24], torch.float64, False), ([1 << 23], torch.float64, True),
                          ([4096, 4096], torch.float32, True), ([4096, 4096], torch.float64, False)):
    d, level = len(dims), 3
    plan = api.Plan(dims, ["db4"] * d, dtype, cplx, True, "reference", max_level=level).set_variant_from_env()
    nb = api.num_bands(d, level)
    shp = tuple(reversed(dims)) + ((2,) if cplx else ())
    x = torch.randn(*shp, device="cuda", dtype=dtype)
    y = torch.empty((nb,) + shp, device="cuda", dtype=dtype)
    r = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    V = 1
    for n in dims:
        V *= n
    esz = x.element_size() * (2 if cplx else 1)
    print(f"{dims} {dtype} cplx={int(cplx)} {plan.describe()}: {dt * 1e3:.3f} ms per dec+rec, {V / dt / 1e6:.0f} Mvox/s, "
          f"roofline frac {2 * level * (1 + 2 ** d) * V * esz / dt / 8e12:.3f}, rt {float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double())):.2e}", flush=True)
