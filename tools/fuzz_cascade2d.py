#!/usr/bin/env python3
"""Randomised parity sweep of the cascaded 2-D kernels (Fwd2C / Inv2C, forced through variant 11 / 12 on small images): random row
lengths (multiples of 4) and heights, mixed db1 .. db4 / db6 wavelets, 2 .. 5 levels, rows per wave forced now and then -- against the
CPU oracle and against one launch per level.   python tools/fuzz_cascade2d.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ndwt_amd as ndwt  # noqa: E402
import ndwt_oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for k in range(cases):
    wn = [f"db{rng.choice([1, 2, 3, 4, 4, 6])}" for _ in range(2)]
    Lp = max(len(orc.wave_filters(w)[0]) for w in wn)
    level = int(rng.integers(2, 6))
    n1 = 4 * int(rng.integers(max(3, (Lp + 3) // 4), 160))
    n2 = int(rng.integers(3 * (Lp - 1) + 1, 160))
    l2 = int(rng.integers(0, 2))
    chunk = int(rng.choice([0, 0, 5, 9, 17, 40]))
    depth = int(rng.choice([11, 12]))
    x = rng.standard_normal((n1, n2))
    xg = torch.from_numpy(np.ascontiguousarray(x.T)).cuda().float().permute(1, 0)
    res = {}
    for name, (vf, vi) in {"cascade": (11, depth), "per-level": (9, 9)}.items():
        w = ndwt.nd_dwt_2D(wn, [n1, n2], "pres_l2_norm", l2, "precision", "single")
        p = w._plan(False, level, xg.device)
        p.set_variant(fwd=vf, inv=vi)
        p.set_tuning(0, chunk if name == "cascade" else 0)
        y = w.dec(xg, level)
        want = orc.spatial_dec(x, wn, level, l2)
        c = rng.standard_normal(want.shape) if name == "cascade" else c
        cg = torch.from_numpy(np.ascontiguousarray(np.transpose(c))).cuda().float().permute(2, 1, 0)
        r = w.rec(cg)
        res[name] = (y, r)
    want_r = orc.spatial_rec(c, wn, l2)
    y, r = res["cascade"]
    e_dec = float(np.abs(y.cpu().numpy() - want).max() / np.abs(want).max())
    e_rec = float(np.abs(r.cpu().numpy() - want_r).max() / max(np.abs(want_r).max(), np.abs(c).max()))
    d_dec = float((y - res["per-level"][0]).abs().max())
    d_rec = float((r - res["per-level"][1]).abs().max() / res["per-level"][1].abs().max())
    ok = e_dec <= 3e-6 and e_rec <= 6e-6 and d_dec == 0.0 and d_rec <= 4e-6
    worst = max(worst, e_dec / 3e-6, e_rec / 6e-6)
    print(f"{'ok  ' if ok else 'FAIL'} {n1}x{n2} {wn} L={level} l2={l2} rows/wave={chunk} inv={depth}: dec {e_dec:.1e} rec {e_rec:.1e} | vs per-level dec {d_dec:.1e} rec {d_rec:.1e}", flush=True)
    if not ok:
        sys.exit(1)
print(f"{cases} cases passed; worst error / tolerance = {worst:.3f}")
