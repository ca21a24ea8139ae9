"""The oracle against itself, the golden fixtures and the reference's known answers (CPU only).

Reference acceptance criteria (Test/nddwt{1,2,3,4}D_test.m:25-27, mex/mex_test.m:28,41): (A) rec(dec(x)) == x,
(B) energy equality with pres_l2_norm, (C) backends agree, (E) real in -> real out.
"""
import glob
import os

import numpy as np
import pytest

import ndwt_oracle as orc

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def test_wave_filters_db2_values():
    lo, hi = orc.wave_filters("db2")   # SURVEY.md 3.4 / wave_filters.m:25-28,164-172
    np.testing.assert_allclose(lo, [-0.12940952255126037, 0.2241438680420134, 0.8365163037378079, 0.48296291314453416], rtol=1e-15)
    np.testing.assert_allclose(hi, [-0.48296291314453416, 0.8365163037378079, -0.2241438680420134, -0.12940952255126037], rtol=1e-15)
    with pytest.raises(ValueError):
        orc.wave_filters("sym4")


@pytest.mark.parametrize("K", range(1, 11))
def test_filters_are_orthonormal(K):
    lo, hi = orc.wave_filters(f"db{K}")
    L = 2 * K
    assert len(lo) == L
    assert abs(lo.sum() - np.sqrt(2)) < 1e-14 and abs(hi.sum()) < 1e-13
    for k in range(K):
        assert abs(np.dot(lo[: L - 2 * k], lo[2 * k:]) - (k == 0)) < 1e-13
        assert abs(np.dot(hi[: L - 2 * k], hi[2 * k:]) - (k == 0)) < 1e-13
        assert abs(np.dot(lo[: L - 2 * k], hi[2 * k:])) < 1e-13


def test_band_bookkeeping():
    assert [orc.num_bands(d, 3) for d in (1, 2, 3, 4)] == [4, 10, 22, 46]
    for d in (1, 2, 3, 4):
        for lev in range(1, 7):
            assert orc.level_from_bands(d, orc.num_bands(d, lev)) == lev


CROSS = [([37], "db2"), ([12, 10], ["db1", "db3"]), ([8, 7, 6], ["db1", "db3", "db2"]), ([6, 7, 6, 5], ["db1", "db3", "db1", "db2"])]


@pytest.mark.parametrize("sizes,wn", CROSS)
@pytest.mark.parametrize("l2", [0, 1])
@pytest.mark.parametrize("cplx", [0, 1])
def test_three_restatements_agree(sizes, wn, l2, cplx):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    for level in (1, 2, 3):
        mat, mex = orc.NdDwtMat(wn, sizes, l2), orc.NdDwtMex(wn, sizes, l2)
        y = mat.dec(x, level)
        assert np.iscomplexobj(y) == bool(cplx)                                  # (E)
        assert np.abs(y - mex.dec(x, level)).max() < 1e-12                       # (C) mat == mex control flow
        assert np.abs(y - orc.spatial_dec(x, wn, level, l2)).max() < 1e-12       # FFT domain == signal domain
        for r in (mat.rec(y), mex.rec(y), orc.spatial_rec(y, wn, l2)):
            assert np.abs(r - x).max() < 1e-12                                   # (A)
        if l2:
            assert abs(np.linalg.norm(y) - np.linalg.norm(x)) < 1e-12 * np.linalg.norm(x)   # (B)
        ya = orc.spatial_dec(x, wn, level, l2, "atrous")
        assert np.abs(orc.spatial_rec(ya, wn, l2, "atrous") - x).max() < 1e-12
        if level == 1:
            assert np.abs(ya - y).max() < 1e-12                                  # level 1 identical in both modes


@pytest.mark.parametrize("l2", [0, 1])
def test_db1_closed_form_of_the_reference(l2):
    # Functions/harr_nddwt_2D.m:263-322 (obj.scale :121-126, /4 in rec :221-223)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((9, 11))
    scale = 0.5 if l2 else 1 / np.sqrt(2)
    m = orc.NdDwtMat("db1", [9, 11], l2)
    y = m.dec(x, 1)
    assert np.abs(y - orc.haar2d_level1_dec(x, scale)).max() < 1e-14
    c = rng.standard_normal((9, 11, 4))
    r = orc.haar2d_level1_rec(c, scale)
    r = r if l2 else r / 4
    assert np.abs(m.rec(c) - r).max() < 1e-14


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    sizes, level, wn = list(g["sizes"]), int(g["level"]), [str(w) for w in g["wname"]]
    wn1 = wn[0] if len(sizes) == 1 else wn
    for tag, l2s in (("r", (0, 1)), ("c", (1,))):
        for l2 in l2s:
            y = orc.spatial_dec(g[f"x_{tag}"], wn1, level, l2)          # the independent restatement
            assert np.abs(y - g[f"y_{tag}_l2{l2}"]).max() < 1e-12
            r = orc.NdDwtMex(wn1, sizes, l2).rec(g[f"c_{tag}"])
            assert np.abs(r - g[f"rec_{tag}_l2{l2}"]).max() < 1e-12
    assert np.abs(orc.spatial_dec(g["x_r"], wn1, level, 0, "atrous") - g["y_r_l20_atrous"]).max() < 1e-13


def test_filter_longer_than_axis_is_rejected():
    with pytest.raises(ValueError, match="Second Dimension of Data is shorter"):
        orc.NdDwtMat(["db1", "db4"], [16, 6])
