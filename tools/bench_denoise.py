#!/usr/bin/env python3
"""cfg3 volume: dec + rec, dec + shrink pass + rec, and ndwt_denoise (threshold fused into the reconstruction kernels)."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n, level = 512, 3
plan = api.Plan([n, n, n], ["db4"] * 3, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
nb = api.num_bands(3, level)
x = torch.randn(n, n, n, device="cuda")
y = torch.empty((nb, n, n, n), device="cuda")
r = torch.empty_like(x)
r2 = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def plain():
    plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)


def three_pass():
    plan.dec(x.data_ptr(), y.data_ptr(), level, s); plan.shrink(y.data_ptr(), level, 0.5, False, s); plan.rec(y.data_ptr(), r.data_ptr(), level, s)


def fused():
    plan.denoise(x.data_ptr(), r2.data_ptr(), level, 0.5, False, s)


print("dec+rec %.3f ms | dec+shrink+rec %.3f ms | ndwt_denoise %.3f ms" % (timed(plain), timed(three_pass), timed(fused)))
three_pass(); fused(); torch.cuda.synchronize()
print("fused vs three-pass: max |diff| = %.3e (max |x| = %.3f)" % (float((r - r2).abs().max()), float(r.abs().max())))
