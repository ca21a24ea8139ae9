#!/usr/bin/env python3
"""Interleaved A/B of variants (Plan.set_variant) of the fused 2-D path: 3 levels of dec, rec and dec+rec on an n x n fp32 image, timed
round-robin in one process; the reconstruction compared with the first variant's.
python tools/ab_variant2d.py 0,7 [wname] [n] [fwd|inv|both] [f32|f64]   (which direction the variant number applies to; default inv)"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
variants = [int(v) for v in sys.argv[1].split(",")]
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
which = sys.argv[4] if len(sys.argv) > 4 else "inv"
dt = torch.float64 if len(sys.argv) > 5 and sys.argv[5] == "f64" else torch.float32
level = 3
plans = {}
for v in variants:
    plans[v] = api.Plan([n, n], [wname] * 2, dt, False, True, "reference", max_level=level)
    plans[v].set_variant(fwd=v if which in ("fwd", "both") else -1, inv=v if which in ("inv", "both") else -1)
x = torch.randn(n, n, device="cuda", dtype=dt)
y = torch.empty(api.num_bands(2, level), n, n, device="cuda", dtype=dt)
r = {v: torch.empty(n, n, device="cuda", dtype=dt) for v in variants}
s = torch.cuda.current_stream().cuda_stream
for v in variants:
    plans[v].dec(x.data_ptr(), y.data_ptr(), level, s)
    plans[v].rec(y.data_ptr(), r[v].data_ptr(), level, s)
torch.cuda.synchronize()
for v in variants[1:]:
    print(f"variant {v}: max |diff to variant {variants[0]}| = {float((r[v] - r[variants[0]]).abs().max()):.2e} (max |x| {float(r[variants[0]].abs().max()):.2f})")
tot = {v: [0.0, 0.0] for v in variants}
reps = 40
for k in range(reps + 3):
    for v in variants:
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        plans[v].dec(x.data_ptr(), y.data_ptr(), level, s)
        e[1].record()
        plans[v].rec(y.data_ptr(), r[v].data_ptr(), level, s)
        e[2].record()
        torch.cuda.synchronize()
        if k >= 3:
            tot[v][0] += e[0].elapsed_time(e[1])
            tot[v][1] += e[1].elapsed_time(e[2])
print(f"{wname} {n}^2 L{level} ({which})", {v: f"dec {tot[v][0] / reps * 1e3:.1f} rec {tot[v][1] / reps * 1e3:.1f} us" for v in variants})
