#!/usr/bin/env python3
"""Interleaved A/B of launch geometries for one direction (dec or rec) on the cfg3 volume: the configurations are
timed round-robin inside one process so that clock / temperature drift hits all of them alike.
python tools/ab_fwd.py dec|rec zchunk[,zchunk...] [n3]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
which = sys.argv[1]
zcs = [int(v) for v in sys.argv[2].split(",")]
n3 = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dims, level = [512, 512, n3], 3
plan = api.Plan(dims, ["db4"] * 3, torch.float32, False, True, "reference", max_level=level).set_variant_from_env()
nb = api.num_bands(3, level)
x = torch.randn(n3, 512, 512, device="cuda")
y = torch.empty((nb, n3, 512, 512), device="cuda")
s = torch.cuda.current_stream().cuda_stream
plan.dec(x.data_ptr(), y.data_ptr(), level, s)
tot = {z: 0.0 for z in zcs}
reps = 15
for r in range(reps + 2):
    for z in zcs:
        plan.set_tuning(0, z)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if which == "dec":
            plan.dec(x.data_ptr(), y.data_ptr(), level, s)
        else:
            plan.rec(y.data_ptr(), x.data_ptr(), level, s)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            tot[z] += e0.elapsed_time(e1)
print(which, {z: round(t / reps / level, 4) for z, t in tot.items()}, "ms per level")
