#!/usr/bin/env python3
"""Interleaved A/B of fused-kernel variants (Plan.set_variant) on one volume, one level per launch, timed round-robin in one
process; every variant's result is compared with the first one's.
python tools/ab_variant.py fwd|inv 0,6 [wname] [n] [level]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
inv = sys.argv[1] == "inv"
variants = [int(v) for v in sys.argv[2].split(",")]
wname = sys.argv[3] if len(sys.argv) > 3 else "db4"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 512
level = int(sys.argv[5]) if len(sys.argv) > 5 else 3
plans = {v: api.Plan([n, n, n], [wname] * 3, torch.float32, False, True, "reference", max_level=level).set_variant(inv=v) if inv else
         api.Plan([n, n, n], [wname] * 3, torch.float32, False, True, "reference", max_level=level).set_variant(fwd=v) for v in variants}
nb = api.num_bands(3, level)
y = torch.randn((nb, n, n, n), device="cuda")
x = torch.randn((n, n, n), device="cuda")
s = torch.cuda.current_stream().cuda_stream
outs = {}


def run(v):
    if inv:
        plans[v].rec(y.data_ptr(), x.data_ptr(), level, s)
    else:
        plans[v].dec(x.data_ptr(), y.data_ptr(), level, s)


for v in variants:
    run(v)
    torch.cuda.synchronize()
    outs[v] = (x if inv else y).clone() if v == variants[0] else float(((x if inv else y) - outs[variants[0]]).abs().max())
    if inv:
        x.normal_()
for v in variants[1:]:
    print(f"variant {v}: max |diff to variant {variants[0]}| = {outs[v]:.2e}" if not inv else f"variant {v}: (rec overwrites x; parity is ab_inv.py's job)")
tot = {v: 0.0 for v in variants}
reps = 15
for r in range(reps + 2):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(v)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            tot[v] += e0.elapsed_time(e1)
print("inv" if inv else "fwd", wname, n, {v: round(t / reps / level, 4) for v, t in tot.items()}, "ms per level")
