"""Step time of the headline configuration for several separately allocated coefficient arrays (all alive at once, so every one has its own place):
does the allocator's placement change the time?  Also: x / r re-allocated.  python tools/exp_placement2.py [k]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, level = 512, 3
V = n ** 3
nb = api.num_bands(3, level)
plan = api.Plan([n] * 3, ["db4"] * 3, torch.float32, False, True, "reference", max_level=level)
s = torch.cuda.current_stream().cuda_stream
xs = [torch.randn(n, n, n, device="cuda") for _ in range(2)]
rs = [torch.empty(n, n, n, device="cuda") for _ in range(2)]
ys = [torch.empty(nb * V, device="cuda", dtype=torch.float32) for _ in range(k)]


def timed(x, y, r, reps=20):
    for _ in range(5):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s)
        plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.dec(x.data_ptr(), y.data_ptr(), level, s)
        plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(2):
    for i, y in enumerate(ys):
        print(f"round {rnd}  y[{i}] at {y.data_ptr():#x} (mod 4 GiB {y.data_ptr() % (1 << 32):#x})  x0 r0: {timed(xs[0], y, rs[0]):.3f} ms   x1 r1: {timed(xs[1], y, rs[1]):.3f} ms")
print("x", [hex(x.data_ptr()) for x in xs], "r", [hex(r.data_ptr()) for r in rs])
