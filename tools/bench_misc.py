#!/usr/bin/env python3
"""Throughput of the non-headline paths: complex 3-D (per-axis kernels), fp64 3-D, a-trous 3-D."""
import importlib
import json
import sys
import time

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")


def run(name, dims, dtype, cplx, dilation, level=3, K=5, pitched=False, wname="db4"):
    plan = api.Plan(dims, [wname] * len(dims), dtype, cplx, True, dilation, max_level=level).set_variant_from_env()
    nb = api.num_bands(len(dims), level)
    bp = plan.band_pitch() if pitched else 0
    shp = tuple(reversed(dims)) + ((2,) if cplx else ())
    x = torch.randn(*shp, device="cuda", dtype=dtype)
    y = torch.empty(nb * (bp if bp else x.numel() // (2 if cplx else 1)) * (2 if cplx else 1), device="cuda", dtype=dtype)
    r = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s, band_pitch=bp); plan.rec(y.data_ptr(), r.data_ptr(), level, s, band_pitch=bp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s, band_pitch=bp); plan.rec(y.data_ptr(), r.data_ptr(), level, s, band_pitch=bp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    V = 1
    for d in dims:
        V *= d
    esz = x.element_size() * (2 if cplx else 1)
    print(json.dumps({"case": name + (" [pitched bands]" if pitched else ""), "path": plan.describe(), "ms_per_step": round(dt * 1e3, 3), "Mvox_s": round(V / dt / 1e6, 1),
                      "roofline_frac": round(2 * level * (1 + 2 ** len(dims)) * V * esz / dt / 8e12, 4),
                      "rt": float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double()))}))


run("3D complex64 256^3 db4 L3 (reference test input type)", [256, 256, 256], torch.float32, True, "reference")
run("3D fp64 256^3 db4 L3", [256, 256, 256], torch.float64, False, "reference")
run("3D complex128 256^3 db4 L3 (reference test input type, mex precision)", [256, 256, 256], torch.float64, True, "reference")
run("3D fp32 256^3 db4 L3 a-trous", [256, 256, 256], torch.float32, False, "atrous")
run("3D fp32 256^3 db4 L3", [256, 256, 256], torch.float32, False, "reference")
for a in (("3D complex64 256^3 db4 L3", [256, 256, 256], torch.float32, True, "reference"), ("3D fp64 256^3 db4 L3", [256, 256, 256], torch.float64, False, "reference"),
          ("3D complex128 256^3 db4 L3", [256, 256, 256], torch.float64, True, "reference"), ("3D fp32 256^3 db4 L3 a-trous", [256, 256, 256], torch.float32, False, "atrous"),
          ("3D fp32 256^3 db4 L3", [256, 256, 256], torch.float32, False, "reference")):
    run(*a, pitched=True)
run("3D fp32 512^3 db6 L4 (cfg4 transform)", [512, 512, 512], torch.float32, False, "reference", level=4, wname="db6")
run("3D fp32 512^3 db6 L4 (cfg4 transform)", [512, 512, 512], torch.float32, False, "reference", level=4, wname="db6", pitched=True)
