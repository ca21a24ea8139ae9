#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/ndwt_oracle.py, FFT-domain 'mat' restatement
of Functions/nd_dwt_{1,2,3,4}D.m; the a-trous vectors come from the signal-domain restatement).

The reference stores no vectors and cannot run here, so these fixtures pin OUR oracle's
outputs: inputs x, coefficients y = dec(x, level), arbitrary coefficients c and r = rec(c).
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ndwt_oracle as orc  # noqa: E402

CASES = {
    "d1_n37_db2_L3": dict(sizes=[37], wname="db2", level=3),
    "d2_12x10_db1db3_L2": dict(sizes=[12, 10], wname=["db1", "db3"], level=2),
    "d3_8x7x6_db1db3db2_L2": dict(sizes=[8, 7, 6], wname=["db1", "db3", "db2"], level=2),
    "d3_10x9x8_db4_L3": dict(sizes=[10, 9, 8], wname="db4", level=3),
    "d4_5x6x4x5_mixed_L2": dict(sizes=[5, 6, 4, 5], wname=["db1", "db3", "db1", "db2"], level=2),
}


def main():
    for name, c in CASES.items():
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        sizes, wname, level = c["sizes"], c["wname"], c["level"]
        d = len(sizes)
        out = {"sizes": np.array(sizes), "level": np.array(level),
               "wname": np.array(wname if isinstance(wname, list) else [wname] * d)}
        for cplx in (0, 1):
            x = rng.standard_normal(sizes)
            cc = rng.standard_normal(sizes + [orc.num_bands(d, level)])
            if cplx:
                x = x + 1j * rng.standard_normal(sizes)
                cc = cc + 1j * rng.standard_normal(cc.shape)
            tag = "c" if cplx else "r"
            out[f"x_{tag}"] = x
            out[f"c_{tag}"] = cc
            for l2 in ((1,) if cplx else (0, 1)):      # variants kept small: real l2 0/1, complex l2 1
                m = orc.NdDwtMat(wname, sizes, l2)
                out[f"y_{tag}_l2{l2}"] = m.dec(x, level)
                out[f"rec_{tag}_l2{l2}"] = m.rec(cc)
            if not cplx:                               # a-trous mode: real, l2 off
                out["y_r_l20_atrous"] = orc.spatial_dec(x, wname, level, 0, "atrous")
                out["rec_r_l20_atrous"] = orc.spatial_rec(cc, wname, 0, "atrous")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
