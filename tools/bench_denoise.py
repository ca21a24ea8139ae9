#!/usr/bin/env python3
"""cfg3 volume: dec + rec, dec + shrink pass + rec, ndwt_denoise with the level-1 detail bands in memory (threshold fused into the
reconstruction kernels' loads: the round-2 path) and ndwt_denoise with level 1 in one launch that recomputes them (Den3).
python tools/bench_denoise.py [n] [wname] [level]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
plan = api.Plan([n, n, n], [wname] * 3, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
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


t_plain, t_three = timed(plain), timed(three_pass)
plan.set_fused_level1(0)
t_mat = timed(fused)
plan.set_fused_level1(2)
t_fused = timed(fused)
print(f"{n}^3 {wname} L{level}: dec+rec {t_plain:.3f} ms | dec+shrink+rec {t_three:.3f} ms | ndwt_denoise, level-1 bands in memory {t_mat:.3f} ms | "
      f"ndwt_denoise, level 1 fused {t_fused:.3f} ms")
plan.set_profiling(True)
fused()
torch.cuda.synchronize()
for k, name in ((0, "fused analysis launches"), (1, "fused synthesis launches (incl. Den3)")):
    ms, cnt = plan.get_profile(k)
    print(f"  {name}: {cnt} launches, {ms:.3f} ms")
plan.set_profiling(False)
three_pass(); fused(); torch.cuda.synchronize()
print("fused vs three-pass: max |diff| = %.3e (max |x| = %.3f)" % (float((r - r2).abs().max()), float(r.abs().max())))
