"""The C / OpenMP restatement of the oracle (oracle/ndwt_spatial.c) against the numpy restatements: the signal-domain formula, the
FFT-domain class of the reference ('mat' path, nd_dwt_3D.m:142-256), the nddwt.c control flow, the db1 closed form the reference ships
and the committed golden fixtures.  CPU only."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ndwt_oracle as orc  # noqa: E402
import ndwt_spatial as orc_c  # noqa: E402

CASES = [
    # shape,            wavelets,                      level, l2, dilation,    dtype,      complex
    ((37,),             "db3",                         3,     0,  "reference", np.float64, False),
    ((4096,),           "db2",                         3,     0,  "reference", np.float64, True),     # cfg1's shape (example_nd_dwt_1D.m)
    ((12, 20),          ["db2", "db4"],                2,     1,  "reference", np.float64, True),
    ((50, 19),          ["db9", "db7"],                1,     1,  "reference", np.float32, False),
    ((16, 12, 10),      ["db1", "db3", "db2"],         3,     0,  "reference", np.float64, False),
    ((24, 16, 24),      "db2",                         3,     1,  "atrous",    np.float64, False),
    ((20, 24, 28),      "db10",                        2,     0,  "reference", np.float64, False),
    ((40, 33, 21),      "db6",                         2,     1,  "reference", np.float32, False),
    ((8, 6, 4, 11),     ["db4", "db3", "db1", "db2"],  2,     0,  "reference", np.float32, True),
    ((16, 16, 16, 16),  "db2",                         2,     1,  "atrous",    np.float64, True),
]


def _data(shape, dt, cplx, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape) + (1j * rng.standard_normal(shape) if cplx else 0)
    return x.astype((np.complex64 if dt == np.float32 else np.complex128) if cplx else dt)


@pytest.mark.parametrize("shape,wn,level,l2,dil,dt,cplx", CASES)
def test_c_oracle_equals_the_numpy_signal_domain_formula(shape, wn, level, l2, dil, dt, cplx):
    tol = 1e-13 if dt == np.float64 else 2e-6
    x = _data(shape, dt, cplx, 1)
    wide = np.complex128 if cplx else np.float64
    want = orc.spatial_dec(x.astype(wide), wn, level, l2, dil)
    got = orc_c.spatial_dec(x, wn, level, l2, dil)
    assert got.shape == want.shape and got.dtype == x.dtype
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
    c = _data(want.shape, dt, cplx, 2)
    want_r = orc.spatial_rec(c.astype(wide), wn, l2, dil)
    got_r = orc_c.spatial_rec(c, wn, l2, dil)
    assert np.abs(got_r - want_r).max() <= 4 * tol * np.abs(want_r).max()
    assert np.abs(orc_c.spatial_rec(got, wn, l2, dil) - x).max() <= 20 * tol * np.abs(x).max()          # perfect reconstruction


@pytest.mark.parametrize("shape,wn,level,l2", [((18, 14, 22), "db4", 3, 1), ((18, 14, 22), ["db2", "db5", "db3"], 2, 0), ((32, 30), "db6", 3, 0),
                                               ((64,), "db8", 2, 1), ((8, 6, 10, 12), "db2", 2, 1)])
def test_c_oracle_equals_the_fft_domain_restatements(shape, wn, level, l2):
    """Independent algorithms: DFT-domain products of the reference's classes ('mat', and the nddwt.c op sequence) vs periodic correlations in C."""
    for cplx in (False, True):
        x = _data(shape, np.float64, cplx, 3)
        got = orc_c.spatial_dec(x, wn, level, l2)
        for cls in (orc.NdDwtMat, orc.NdDwtMex):
            w = cls(wn, list(shape), l2)
            want = w.dec(x, level)
            assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), cls.__name__
            c = _data(want.shape, np.float64, cplx, 4)
            assert np.abs(orc_c.spatial_rec(c, wn, l2) - w.rec(c)).max() <= 1e-12 * np.abs(c).max(), cls.__name__


def test_c_oracle_reproduces_the_db1_closed_form_of_the_reference():
    """Functions/harr_nddwt_2D.m:263-322 (restated in the numpy oracle as haar2d_level1_dec / rec): shift-and-add form of one db1 level."""
    x = _data((12, 10), np.float64, False, 5)
    for l2 in (0, 1):
        scale = 0.5 if l2 else 1 / np.sqrt(2.0)
        want = orc.haar2d_level1_dec(x, scale)
        got = orc_c.spatial_dec(x, "db1", 1, l2)
        assert np.abs(got - want).max() <= 1e-14


def test_c_oracle_against_the_golden_fixtures():
    """tests/golden/*.npz (made by tests/golden/make_golden.py from the FFT-domain restatement): every stored transform, real and complex."""
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert files
    checked = 0
    for f in files:
        z = np.load(f, allow_pickle=True)
        wn = [str(w) for w in np.atleast_1d(z["wname"])]
        level = int(z["level"])
        for key in z.files:
            if not key.startswith("y_"):
                continue
            kind, tag = key[2], key[4:]                      # y_<r|c>_l2<0|1>[_atrous]
            l2, dil = int(tag[2]), "atrous" if tag.endswith("atrous") else "reference"
            got = orc_c.spatial_dec(z["x_" + kind], wn, level, l2, dil)
            assert np.abs(got - z[key]).max() <= 1e-12 * np.abs(z[key]).max(), (f, key)
            got_r = orc_c.spatial_rec(z["c_" + kind], wn, l2, dil)
            want_r = z["rec_" + key[2:]]
            assert np.abs(got_r - want_r).max() <= 1e-12 * np.abs(want_r).max(), (f, key)
            checked += 1
    assert checked >= 15


def test_c_oracle_thread_count_does_not_change_the_result():
    L = orc_c.load()
    x = _data((24, 20, 16), np.float32, False, 6)
    n0 = L.ndwt_c_max_threads()
    try:
        L.ndwt_c_set_threads(1)
        a = orc_c.dec_planar(x, "db4", 2, 1)
        L.ndwt_c_set_threads(max(2, n0))
        b = orc_c.dec_planar(x, "db4", 2, 1)
    finally:
        L.ndwt_c_set_threads(n0)
    assert np.array_equal(a, b)
