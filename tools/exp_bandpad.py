#!/usr/bin/env python3
"""Timing experiment (diagnostic library built with -DNDWT_EXP_BANDPAD): does the power-of-two distance between the band
planes cost bandwidth?  NDWT_EXP_BANDPAD=<elements> skews band b by b * pad elements (reads / writes land in the neighbouring
band: results are garbage, the traffic pattern is what is timed).
NDWT_LIB_VARIANT=exp python tools/exp_bandpad.py [pad]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n, level = 512, 1
nb = api.num_bands(3, level)
x = torch.randn(n, n, n, device="cuda")
y = torch.randn((nb + 1, n, n, n), device="cuda")      # one spare band: the skewed pointers stay inside the allocation
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
cases = [(0, 1), (0, 1), (64, 1), (16, 1), (32, 1), (128, 1), (192, 1), (64, 2), (64, 4), (128, 2), (4160, 1), (0, 1)]
if len(sys.argv) > 2:
    cases = [tuple(int(v) for v in c.split('/')) for c in sys.argv[1:]]
elif len(sys.argv) > 1:
    cases = [(int(sys.argv[1]), 1)]
for case in cases:
    pad, div = case[0], case[1]
    os.environ["NDWT_EXP_BANDMASK"] = str(case[2]) if len(case) > 2 else "0"
    os.environ["NDWT_EXP_BANDPAD"] = str(pad)
    os.environ["NDWT_EXP_BANDDIV"] = str(div)
    plan = api.Plan([n, n, n], ["db4"] * 3, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
    for _ in range(3):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s)
        plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(20):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s)
    e[1].record()
    for _ in range(20):
        plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    e[2].record()
    torch.cuda.synchronize()
    print(f"pad {pad:8d} elements, groups of {div}, mask {case[2] if len(case) > 2 else 0}: dec {e[0].elapsed_time(e[1]) / 20:.3f} ms  rec {e[1].elapsed_time(e[2]) / 20:.3f} ms", flush=True)
