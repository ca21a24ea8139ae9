"""What slows the cascaded 2-D synthesis when it alternates with the analysis (cfg2: 180 us in a loop of its own, 220 - 235 us in the bench step)?
rec(y) after dec(x -> y) [the step], after dec(x -> y2) [the analysis writes ANOTHER buffer: rec reads data written long ago], after a plain
640-MiB device memset / copy, and alone.  python tools/exp_rec_after_dec.py"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n, level = 4096, 3
plan = api.Plan([n, n], ["db4"] * 2, torch.float32, False, True, "reference", max_level=level)
nb = api.num_bands(2, level)
x = torch.randn(n, n, device="cuda")
y = torch.empty(nb * n * n, device="cuda")
y2 = torch.empty_like(y)
r = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
plan.dec(x.data_ptr(), y.data_ptr(), level, s)


def timed(before, reps=60):
    t = 0.0
    for i in range(reps + 5):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.rec(y.data_ptr(), r.data_ptr(), level, s)
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            t += e0.elapsed_time(e1)
    return t / reps * 1e3


cases = [("alone", lambda: None),
         ("after dec(x -> y)", lambda: plan.dec(x.data_ptr(), y.data_ptr(), level, s)),
         ("after dec(x -> y2)", lambda: plan.dec(x.data_ptr(), y2.data_ptr(), level, s)),
         ("after y2.zero_() (640 MiB written)", lambda: y2.zero_()),
         ("after y2.copy_(y) (640 MiB read + written)", lambda: y2.copy_(y)),
         ("after x.sum() (64 MiB read)", lambda: x.sum()),
         ("after dec(x -> y) + 200 us idle", lambda: (plan.dec(x.data_ptr(), y.data_ptr(), level, s), torch.cuda._sleep(400000))),
         ("alone", lambda: None)]
for name, fn in cases:
    print(f"rec {name:45s} {timed(fn):7.1f} us")
