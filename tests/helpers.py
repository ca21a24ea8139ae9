"""Shared helpers of the test-suite (CPU side)."""
import numpy as np

import ndwt_oracle as orc


def kernel_taps(wname, pres_l2_norm, Lp=None):
    """Kernel-form taps of one axis (what csrc/ndwt_filters.h:make_axis_filter builds), zero-padded to Lp."""
    lo_d, hi_d = orc.wave_filters(wname)
    L = len(lo_d)
    c = 1 / np.sqrt(2.0) if pres_l2_norm else 1.0
    cs = 1 / np.sqrt(2.0) if pres_l2_norm else 0.5
    t = {"ana_lo": c * lo_d[::-1], "ana_hi": c * hi_d[::-1], "syn_lo": cs * lo_d, "syn_hi": cs * hi_d}
    if Lp is not None:
        pad = (Lp - L) // 2
        t = {k: np.concatenate([np.zeros(pad), v, np.zeros(pad)]) for k, v in t.items()}
    return t


def to_kernel_order(x_mat):
    """MATLAB-shaped array [n1,...,nd(,bands)] -> C-contiguous [(bands,) nd, ..., n1] (column-major memory)."""
    return np.ascontiguousarray(np.transpose(x_mat))


def from_kernel_order(a):
    return np.transpose(a)
