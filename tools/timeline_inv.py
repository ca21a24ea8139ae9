#!/usr/bin/env python3
"""Timeline of two consecutive planes of the pair-packed float synthesis kernel (diagnostic build with -DNDWT_STAMPS): the shader
clock at six points of planes 100 and 101 for every wave of a few workgroups.
  VARIANT=stamps NDWT_DEFS=-DNDWT_STAMPS tools/quick_relink.sh ndwt_fused3_f32_invy ndwt_api
  NDWT_LIB_VARIANT=stamps python tools/timeline_inv.py <variant>"""
import ctypes
import importlib
import os
import sys

import torch

os.environ.setdefault("NDWT_LIB_VARIANT", "stamps")
os.environ["NDWT_VARIANT_INV"] = sys.argv[1] if len(sys.argv) > 1 else "5"
sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
L = importlib.import_module("non-decimated_wavelets_amd._lib")
n = 512
plan = api.Plan([n, n, n], ["db4"] * 3, torch.float32, False, True, "reference", max_level=1).set_variant_from_env()
y = torch.randn(8, n, n, n, device="cuda")
x = torch.empty(n, n, n, device="cuda")
s = torch.cuda.current_stream().cuda_stream
plan.rec(y.data_ptr(), x.data_ptr(), 1, s)
nwg, nw = 1024, 16
buf = torch.zeros(nwg * nw * 2 * 8, dtype=torch.int64, device="cuda")
L.lib().ndwt_plan_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.check(L.lib().ndwt_plan_set_stamps(plan._h, ctypes.c_void_p(buf.data_ptr())))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.rec(y.data_ptr(), x.data_ptr(), 1, s)
e1.record()
torch.cuda.synchronize()
print(f"variant {os.environ['NDWT_VARIANT_INV']}: launch {e0.elapsed_time(e1):.3f} ms")
b = buf.view(nwg, nw, 2, 8).cpu()
names = ["start", "x done", "loads issued", "past barrier", "y done", "z done"]
for wg in (0, 37, 200):
    t = b[wg]
    if int(t.max()) == 0:
        continue
    base = int(t[:, 0, 3][t[:, 0, 3] > 0].min())      # barrier release of plane 100
    print(f"workgroup {wg}: cycles relative to the barrier release of plane 100; columns: " + ", ".join(names))
    for w in range(nw):
        row = []
        for pl in range(2):
            row.append(" ".join(f"{int(t[w, pl, k]) - base:7d}" for k in range(6)))
        print(f"  wave {w:2d} | p100: {row[0]} | p101: {row[1]}")
