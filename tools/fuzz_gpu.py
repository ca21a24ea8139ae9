#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random dimensions (primes, sizes below one tile, non-multiples of 4), mixed
db1..db10 wavelets, levels, precisions, real / complex, both dilations, against the CPU oracle.  Not part of the test
suite (minutes of oracle time); run it after kernel changes:  python tools/fuzz_gpu.py [cases] [seed] [max_order] [p_atrous]   (max_order <= 4 keeps every
case on the fused kernels)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ndwt_amd as ndwt  # noqa: E402
import ndwt_oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_order = int(sys.argv[3]) if len(sys.argv) > 3 else 10
p_atrous = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
CLS = {1: ndwt.nd_dwt_1D, 2: ndwt.nd_dwt_2D, 3: ndwt.nd_dwt_3D, 4: ndwt.nd_dwt_4D}
TOL = {"double": 1e-12, "single": 3e-6}
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402

worst = 0.0
for case in helpers.fuzz_cases(cases, seed, max_order, p_atrous):
    d, sizes, wn, level, l2 = case["d"], case["sizes"], case["wn"], case["level"], case["l2"]
    precision, cplx, dilation = case["precision"], case["cplx"], case["dilation"]
    rng = np.random.default_rng(case["data_seed"])
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    w = CLS[d](wn if d > 1 else wn[0], sizes, "pres_l2_norm", l2, "precision", precision, "dilation", dilation)
    xt = torch.from_numpy(np.ascontiguousarray(np.transpose(x))).cuda()
    xt = xt.to({("single", False): torch.float32, ("double", False): torch.float64, ("single", True): torch.complex64,
                ("double", True): torch.complex128}[(precision, cplx)])
    xg = xt.permute(*reversed(range(d)))
    y = w.dec(xg, level)
    want = orc.spatial_dec(x, wn, level, l2, dilation)
    e_dec = float(np.abs(y.cpu().numpy() - want).max() / np.abs(want).max())
    r = w.rec(y)
    e_rt = float(np.abs(r.cpu().numpy() - x).max() / np.abs(x).max())
    c = rng.standard_normal(want.shape) + (1j * rng.standard_normal(want.shape) if cplx else 0)
    ct = torch.from_numpy(np.ascontiguousarray(np.transpose(c))).cuda().to(xt.dtype)
    r2 = w.rec(ct.permute(*reversed(range(d + 1))))
    want_r = orc.spatial_rec(c, wn, l2, dilation)
    e_rec = float(np.abs(r2.cpu().numpy() - want_r).max() / max(np.abs(want_r).max(), 1e-30))
    tol = TOL[precision]
    ok = e_dec <= tol and e_rec <= 4 * tol and e_rt <= 20 * tol
    worst = max(worst, e_dec / tol, e_rec / (4 * tol), e_rt / (20 * tol))
    print(f"{'ok  ' if ok else 'FAIL'} d={d} sizes={sizes} wn={wn} L={level} l2={l2} {precision} cplx={int(cplx)} {dilation} "
          f"path={w._plan(cplx, level, xt.device).describe()} dec={e_dec:.2e} rec={e_rec:.2e} rt={e_rt:.2e}", flush=True)
    if not ok:
        sys.exit(1)
print(f"{cases} cases passed; worst error / tolerance = {worst:.3f}")
