#!/usr/bin/env python3
"""Where the float synthesis kernel spends its cycles: per-wave s_memtime sums of the plane loop's phases (diagnostic build
`make -C non-decimated_wavelets_amd/csrc VARIANT=stamps DEFS=-DNDWT_STAMPS`, loaded with NDWT_LIB_VARIANT=stamps).
Read the SHARES, not the run time (the stamps' fences forbid overlaps the product kernel has).
  NDWT_LIB_VARIANT=stamps python tools/stamps_inv.py [wname] [n]"""
import ctypes
import importlib
import os
import sys

import torch

os.environ.setdefault("NDWT_LIB_VARIANT", "stamps")
sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
L = importlib.import_module("non-decimated_wavelets_amd._lib")
if len(sys.argv) > 3:
    os.environ["NDWT_VARIANT_INV"] = sys.argv[3]
wname = sys.argv[1] if len(sys.argv) > 1 else "db4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
plan = api.Plan([n, n, n], [wname] * 3, torch.float32, False, True, "reference", max_level=1).set_variant_from_env()
y = torch.randn(8, n, n, n, device="cuda")
x = torch.empty(n, n, n, device="cuda")
s = torch.cuda.current_stream().cuda_stream
plan.rec(y.data_ptr(), x.data_ptr(), 1, s)
nwg, nwaves = 4096, 16
buf = torch.zeros(nwg * nwaves * 4, dtype=torch.int64, device="cuda")
L.lib().ndwt_plan_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.check(L.lib().ndwt_plan_set_stamps(plan._h, ctypes.c_void_p(buf.data_ptr())))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.rec(y.data_ptr(), x.data_ptr(), 1, s)
e1.record()
torch.cuda.synchronize()
b = buf.view(nwg, nwaves, 4).double()
used = b.sum(dim=(1, 2)) > 0
b = b[used]
tot = b.sum(dim=2)
print(f"{wname} {n}^3: launch {e0.elapsed_time(e1):.3f} ms, {int(used.sum())} workgroups, mean cycles per wave {float(tot.mean()):.0f}")
names = ["wait for loads", "x stage", "load issue + barrier", "y/z stage"]
for k, nm in enumerate(names):
    sh = b[:, :, k] / tot
    print(f"  {nm:22s} share mean {float(sh.mean()):.3f}  min {float(sh.min()):.3f}  max {float(sh.max()):.3f}   "
          f"cycles/plane (wave mean) {float(b[:, :, k].mean()):.0f} total")
# per-wave picture of workgroup 0: waves 0..15
for w in range(b.shape[1]):
    print("  wg0 wave", w, [int(v) for v in b[0, w].tolist()])
