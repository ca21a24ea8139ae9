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


def fuzz_cases(n, seed, max_order=10, p_atrous=0.2):
    """`n` random transform configurations from a fixed seed (the generator of tools/fuzz_gpu.py): dimensions (primes,
    sizes below one tile, ragged tiles), mixed db1..db10 wavelets, levels, precisions, real / complex, both dilations.
    Returns dicts; the per-case data seed is part of the dict so that a single case can be replayed."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        d = int(rng.choice([1, 2, 3, 3, 3, 4]))
        budget = {1: 5000, 2: 90, 3: 40, 4: 14}[d]
        orders = [int(rng.integers(1, min(max_order, 10 if d < 4 else 4) + 1)) for _ in range(d)]
        sizes = [int(rng.integers(2 * o, max(2 * o + 2, budget))) for o in orders]
        if rng.random() < 0.3:
            sizes[0] = int(rng.choice([64, 68, 72, 128, 132])) if d <= 3 else sizes[0]      # whole tiles / ragged tiles
        level = int(rng.integers(1, 4))
        dilation = "atrous" if rng.random() < p_atrous else "reference"
        if dilation == "atrous":                                 # dilated filters (plans are built for 3 levels) must fit the axis
            orders = [min(o, 3 if d < 4 else 2) for o in orders]
            sizes = [max(s, 2 * o * 4) for s, o in zip(sizes, orders)]
            if rng.random() < 0.7:                               # axes that divide by 4: dilated levels run fused on sub-lattices
                sizes = [4 * ((s + 3) // 4) for s in sizes]
        out.append({"d": d, "sizes": sizes, "wn": [f"db{o}" for o in orders], "level": level, "dilation": dilation,
                    "l2": int(rng.integers(0, 2)), "precision": "single" if rng.random() < 0.5 else "double",
                    "cplx": bool(rng.random() < 0.4), "data_seed": int(seed) * 100003 + k})
    return out


def fuzz_id(c):
    return (f"{c['d']}d-{'x'.join(map(str, c['sizes']))}-{'.'.join(w[2:] for w in c['wn'])}-L{c['level']}-l2{c['l2']}-"
            f"{c['precision'][0]}{'c' if c['cplx'] else 'r'}-{c['dilation'][0]}")
