"""Per-step time of the headline step from a cold start: how many steps the device takes to reach its steady clocks.
usage (GPU box): python tools/ramp_profile.py [steps]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
sizes, level = [512, 512, 512], 3
V = 512 ** 3
plan = api.Plan(sizes, ["db4"] * 3, torch.float32, False, True, "reference", max_level=3, device=0)
x = torch.randn((512, 512, 512), device=dev)
y = torch.empty(api.num_bands(3, level) * V, device=dev)
r = torch.empty_like(x)
s = torch.cuda.current_stream(dev).cuda_stream
torch.cuda.synchronize()
time.sleep(2.0)                                    # let the device fall idle
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    plan.dec(x.data_ptr(), y.data_ptr(), level, s)
    plan.rec(y.data_ptr(), r.data_ptr(), level, s)
    evs[i + 1].record()
torch.cuda.synchronize()
t = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
for a, b in ((0, 1), (1, 5), (5, 10), (10, 25), (25, 50), (50, 100), (100, 200), (200, 300), (300, n)):
    if b <= n:
        print(f"steps {a:4d}..{b:4d}: {sum(t[a:b]) / (b - a):.4f} ms per step")
