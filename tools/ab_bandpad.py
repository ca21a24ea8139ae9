#!/usr/bin/env python3
"""Does the power-of-two band stride (512 MiB between the 8 bands of a 512^3 level) cost anything?  One synthesis /
analysis level through the slab entry points, bands at base + b*(vol + pad) for several pads (interleaved timing)."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = 512
vol = n ** 3
plan = api.Plan([n, n, n], ["db4"] * 3, torch.float32, False, True, "reference", max_level=1).set_variant_from_env()
ab, aa, sb, sa = plan.slab_halo(1)
pads = [0, 1024, 16 * 1024 + 256, 256 * 1024 + 1024, 1024 * 1024 + 4096, 3 * 1024 * 1024 + 17 * 1024]   # elements
big = torch.randn(8 * (vol + max(pads)) + 1024, device="cuda")
out = torch.empty((n + sa + sb) * n * n, device="cuda")
x = torch.randn(vol, device="cuda")
hb = torch.randn(ab * n * n, device="cuda")
ha = torch.randn(aa * n * n, device="cuda")
s = torch.cuda.current_stream().cuda_stream
tot = {(w, p): 0.0 for p in pads for w in ("syn", "ana")}
reps = 10
for k in range(reps + 2):
    for p in pads:
        ptrs = [big.data_ptr() + 4 * b * (vol + p) for b in range(8)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        plan.synthesis_level_slab_ext(ptrs, out.data_ptr(), 1, s)
        ev[1].record()
        plan.analysis_level_slab_split(x.data_ptr(), hb.data_ptr(), ha.data_ptr(), ptrs, 1, s)
        ev[2].record()
        torch.cuda.synchronize()
        if k >= 2:
            tot[("syn", p)] += ev[0].elapsed_time(ev[1])
            tot[("ana", p)] += ev[1].elapsed_time(ev[2])
for w in ("syn", "ana"):
    print(w, {p: round(tot[(w, p)] / reps, 4) for p in pads}, "ms per level, pad in elements")
