"""What the reference itself holds for the hot path, pinned (the reference stores no golden vectors).

  * the tap table -- the literal db1..db10 lists of Functions/wave_filters.m:19-160 -- parsed as text where the reference
    checkout is present (build container; skipped on the GPU box) and compared, to the last bit in fp64, with
    oracle/db_taps.py (spectral factorisation, tools/derive_daubechies.py) and with what the C ABI returns;
  * the db1 closed forms the reference ships as signal-domain code: harr_nddwt_2D.m (tests/test_oracle.py) and the
    4-D one, harr_nddwt_4D.m:262-281,555-579 (multi-level band bookkeeping of :173,:228), against all three oracle
    restatements.  tests/test_gpu_parity.py compares the HIP path with the same closed form.
Parity with MATLAB output stays "unpinned" by the rules (no fixtures exist); these are the pins that do exist.
"""
import os
import re

import numpy as np
import pytest

import ndwt_oracle as orc
from db_taps import DB_H

REF_FILTERS = "/root/reference/Functions/wave_filters.m"


def parse_wave_filters_m(path):
    """{K: [h0, h1, ...]} from the `case {'dbK'} low_d = [ ... ];` blocks, literals converted by float()"""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"case\s*\{'db(\d+)'\}\s*low_d\s*=\s*\[(.*?)\];", txt, re.S):
        K, body = int(m.group(1)), m.group(2)
        body = body.replace("...", " ")
        if "sqrt" in body:                                    # db1: [1/sqrt(2),1/sqrt(2)]
            assert K == 1 and body.replace(" ", "") == "1/sqrt(2),1/sqrt(2)"
            out[K] = [1 / np.sqrt(2.0)] * 2
            continue
        out[K] = [float(v) for v in re.split(r"[,\s]+", body.strip()) if v]
    return out


@pytest.mark.skipif(not os.path.exists(REF_FILTERS), reason="reference checkout not present (GPU box)")
def test_tap_table_equals_reference_literals_to_the_bit():
    ref = parse_wave_filters_m(REF_FILTERS)
    assert sorted(ref) == list(range(1, 11))
    for K in range(1, 11):
        h_ref = np.array(ref[K], dtype=np.float64)
        assert len(h_ref) == 2 * K
        h = np.array([float(v) for v in DB_H[K]], dtype=np.float64)
        if K == 1:
            h = np.array([1 / np.sqrt(2.0)] * 2)
        assert np.array_equal(h, h_ref), (K, np.abs(h - h_ref).max())            # 0 ulp
        # wave_filters.m:164-172 on the parsed literals == the oracle's wave_filters
        hi = h_ref[::-1].copy()
        hi[1::2] = -hi[1::2]
        lo_d, hi_d = orc.wave_filters(f"db{K}")
        assert np.array_equal(lo_d, h_ref[::-1]) and np.array_equal(hi_d, hi[::-1])


@pytest.mark.skipif(not os.path.exists(REF_FILTERS), reason="reference checkout not present (GPU box)")
def test_c_abi_wave_filters_equals_reference_literals_to_the_bit():
    import ndwt_amd as ndwt                                   # loads libndwt_hip.so; ndwt_wave_filters needs no GPU
    ref = parse_wave_filters_m(REF_FILTERS)
    for K in range(1, 11):
        h_ref = np.array(ref[K], dtype=np.float64)
        hi = h_ref[::-1].copy()
        hi[1::2] = -hi[1::2]
        lo_d, hi_d = ndwt._lib.wave_filters(f"db{K}")
        assert np.array_equal(np.array(lo_d), h_ref[::-1]), K
        assert np.array_equal(np.array(hi_d), hi[::-1]), K


@pytest.mark.parametrize("l2", [0, 1])
@pytest.mark.parametrize("level", [1, 2, 3])
def test_haar4d_closed_form_pins_the_oracle(l2, level):
    """harr_nddwt_4D.m closed form == the 'mat' FFT path, the nddwt.c control flow and the signal-domain restatement"""
    rng = np.random.default_rng(70 + level)
    sizes = [6, 5, 4, 7]
    x = rng.standard_normal(sizes)
    want = orc.haar4d_dec(x, level, l2)
    assert want.shape == tuple(sizes) + (16 + 15 * (level - 1),)
    assert np.abs(orc.NdDwtMat("db1", sizes, l2).dec(x, level) - want).max() < 1e-13
    assert np.abs(orc.NdDwtMex("db1", sizes, l2).dec(x, level) - want).max() < 1e-13
    assert np.abs(orc.spatial_dec(x, "db1", level, l2) - want).max() < 1e-13
    c = rng.standard_normal(want.shape)
    r = orc.haar4d_rec(c, l2)
    assert np.abs(orc.NdDwtMat("db1", sizes, l2).rec(c) - r).max() < 1e-13
    assert np.abs(orc.NdDwtMex("db1", sizes, l2).rec(c) - r).max() < 1e-13
    assert np.abs(orc.spatial_rec(c, "db1", l2) - r).max() < 1e-13
    assert np.abs(orc.haar4d_rec(want, l2) - x).max() < 1e-13                    # its own round trip


def _slices_pass(t, axis, diff, scale, rec):
    """one shift-and-add pass written with the file's index ranges (1:end-1 / 2:end plus the wrap row), no roll"""
    t = np.moveaxis(t, axis, 0)
    out = np.empty_like(t)
    if not rec:                                               # harr_nddwt_4D.m:268-269 (sum), :286-287 (difference)
        out[:-1] = scale * (t[:-1] - t[1:]) if diff else scale * (t[:-1] + t[1:])
        out[-1] = scale * (-t[0] + t[-1]) if diff else scale * (t[0] + t[-1])
    else:                                                     # :565-566 (sum), :585-586 (difference)
        out[1:] = scale * (-t[:-1] + t[1:]) if diff else scale * (t[:-1] + t[1:])
        out[0] = scale * (t[0] - t[-1]) if diff else scale * (t[0] + t[-1])
    return np.moveaxis(out, 0, axis)


def test_haar4d_level1_spelled_out_bands():
    """every band of level_1_dec / level_1_rec written with the file's explicit index ranges and wrap rows"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((4, 3, 5, 2))
    s = 0.5
    y = orc.haar4d_level1_dec(x, s)
    c = rng.standard_normal(x.shape + (16,))
    r = np.zeros_like(x)
    for b in range(16):
        t = x
        for a in range(4):
            t = _slices_pass(t, a, (b >> a) & 1, s, rec=False)
        assert np.array_equal(y[..., b], t), b
        t = c[..., b]
        for a in range(4):
            t = _slices_pass(t, a, (b >> a) & 1, s, rec=True)
        r = r + t
    assert np.abs(orc.haar4d_level1_rec(c, s) - r).max() < 1e-15
