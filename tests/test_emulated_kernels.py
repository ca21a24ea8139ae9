"""Host emulation of the fused HIP kernels (same source, clang + ASan/UBSan) against the oracle.

Catches indexing mistakes (global and LDS out-of-bounds, wrong halo/wrap arithmetic, band order) on
the CPU; the -m gpu tests then check the real kernels.  Reference semantics: one level of
Functions/nd_dwt_3D.m:345-374 (level_1_dec / level_1_rec).
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import ndwt_oracle as orc
from helpers import kernel_taps, to_kernel_order

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "emu", "libndwt_emu.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("clang++ of the ROCm toolchain is needed to build the host emulator")
    # with the ASan runtime preloaded (tests/emu/run_emu_tests.sh) use the sanitizer build, else the plain one
    asan = "libclang_rt.asan" in os.environ.get("LD_PRELOAD", "")
    target = "libndwt_emu.so" if asan else "libndwt_emu_plain.so"
    subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(ROOT, "tests", "emu"), target])
    return ctypes.CDLL(os.path.join(ROOT, "tests", "emu", target))


def _run(emu, x_mat_or_bands, wnames, l2, inverse, dtype, vec4, zchunk, small, z_wrap=1, variant=0, cplx=False, shrink=(0.0, 0, 0), dil=1):
    Ls = [len(orc.wave_filters(w)[0]) for w in wnames]
    Lp = max(Ls)
    lo = np.zeros((3, 20))
    hi = np.zeros((3, 20))
    for ax in range(3):
        t = kernel_taps(wnames[ax], l2, Lp)
        lo[ax, :Lp] = t["syn_lo" if inverse else "ana_lo"]
        hi[ax, :Lp] = t["syn_hi" if inverse else "ana_hi"]
    cdt = (np.complex64 if dtype == np.float32 else np.complex128) if cplx else dtype
    src = to_kernel_order(x_mat_or_bands).astype(cdt)
    if inverse:
        n3, n2, n1 = src.shape[1:]
        out = np.full((n3, n2, n1), np.nan, dtype=cdt)
    else:
        n3, n2, n1 = src.shape
        out = np.full((8, n3, n2, n1), np.nan, dtype=cdt)
    if cplx:
        n1 *= 2                       # the kernels see scalars along x
    fn = emu.ndwt_emu3_f32 if dtype == np.float32 else emu.ndwt_emu3_f64
    fn.restype = ctypes.c_int
    rc = fn(int(inverse), Lp, int(vec4), src.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
            n1, n2, n3, 1, zchunk, lo.ctypes.data_as(ctypes.c_void_p), hi.ctypes.data_as(ctypes.c_void_p), z_wrap,
            int(small), int(variant), 2 if cplx else 1, ctypes.c_double(shrink[0]), int(shrink[1]), int(shrink[2]), int(dil))
    assert rc == 0
    return np.transpose(out)


CASES = [
    # sizes (n1,n2,n3),      wavelets,               vec4,  zchunk, small_tile
    ((16, 9, 7), ("db1", "db3", "db2"), True, 0, True),
    ((13, 10, 9), ("db2", "db2", "db2"), False, 4, True),
    ((20, 17, 12), ("db4", "db4", "db4"), True, 5, True),
    ((68, 18, 9), ("db4", "db4", "db4"), True, 0, False),       # production tile shape (db4 only)
    ((70, 19, 10), ("db2", "db1", "db4"), False, 6, False),
    ((134, 20, 7), ("db4", "db4", "db4"), False, 0, False),      # three production tiles along x, the last anchored at the end of the row
    ((130, 18, 6), ("db4", "db4", "db4"), False, 0, False),      # a 2-column last tile; the tile before it reaches past the end too
]


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,zchunk,small", CASES)
@pytest.mark.parametrize("l2", [0, 1])
def test_emulated_fused3_analysis(emu, sizes, wn, vec4, zchunk, small, l2):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(sizes)
    filt = [orc.wave_filters(w) for w in wn]
    want = orc.spatial_level_dec(x, filt, l2)
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        for variant in ((0, 1) if not small and dtype == np.float32 else (0,)):
            got = _run(emu, x, wn, l2, False, dtype, vec4, zchunk, small, variant=variant)
            assert np.isfinite(got).all()
            assert np.abs(got - want).max() <= tol * np.abs(want).max()


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,zchunk,small", CASES)
@pytest.mark.parametrize("l2", [0, 1])
def test_emulated_fused3_synthesis(emu, sizes, wn, vec4, zchunk, small, l2):
    rng = np.random.default_rng(2)
    c = rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    want = orc.spatial_level_rec(c, filt, l2)
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        # variants 1 (production tile, float) and 2 (small tile, every tap length) are the lane-shift synthesis kernel
        # 3: the library's default float synthesis configuration (lane-shift kernel, 64x16 tile, 256 threads)
        # double: 1 = the library's default synthesis configuration (lane-shift kernel, 64x16 tile, 512 threads)
        for variant in ((0, 1, 3) if not small and dtype == np.float32 else (0, 2) if small else (0, 1)):
            got = _run(emu, c, wn, l2, True, dtype, vec4, zchunk, small, variant=variant)
            assert np.isfinite(got).all()
            assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1.0)


CASES_Y = [
    # sizes (n1,n2,n3),      wavelets,               vec4,  zchunk, small_tile   (padding (Lp - len) / 2 even on every axis)
    ((16, 9, 7), ("db1", "db3", "db1"), True, 0, True),
    ((13, 10, 9), ("db2", "db2", "db2"), False, 4, True),
    ((20, 17, 12), ("db4", "db4", "db4"), True, 5, True),
    ((24, 19, 11), ("db6", "db2", "db4"), True, 0, True),        # two-hop lane shifts
    ((68, 39, 9), ("db4", "db4", "db4"), True, 0, False),        # production tile shape (8 and 12 taps): ragged in x and y
    ((70, 60, 10), ("db2", "db2", "db4"), False, 6, False),
    ((134, 36, 7), ("db4", "db4", "db4"), False, 0, False),      # ragged rows on the production tile: end tiles anchored at the row end
    ((131, 34, 6), ("db6", "db6", "db6"), False, 0, False),
    ((24, 21, 13), ("db5", "db5", "db5"), True, 0, True),
    ((21, 14, 12), ("db5", "db3", "db1"), False, 5, True),
    ((24, 23, 19), ("db7", "db7", "db7"), True, 0, True),
    ((28, 20, 21), ("db8", "db8", "db4"), True, 7, True),
    ((20, 26, 20), ("db9", "db9", "db9"), True, 0, True),         # 25 haloed rows: two rounds of rows on the small tile
    ((23, 19, 18), ("db9", "db7", "db5"), False, 6, True),
    ((24, 22, 21), ("db10", "db10", "db10"), True, 0, True),      # 27 haloed rows: three rounds of rows on the small tile
    ((21, 20, 20), ("db10", "db8", "db6"), False, 4, True),
    ((72, 37, 12), ("db6", "db6", "db6"), True, 0, False),       # production tile shape, 12 taps: 15 of the 16 waves hold rows
]


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,zchunk,small", CASES_Y)
@pytest.mark.parametrize("l2", [0, 1])
def test_emulated_pair_packed_synthesis(emu, sizes, wn, vec4, zchunk, small, l2):
    """Inv3Y, the float synthesis default: x pairs on packed FMAs, high-pass taps derived from the low-pass ones by operand
    modifiers, loads only from lanes that hold a row; variant 5 = one register set of band loads, 8 = two with staggered refill"""
    rng = np.random.default_rng(12)
    c = rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    want = orc.spatial_level_rec(c, filt, l2)
    for variant in ((5, 8) if vec4 else (5,)):
        got = _run(emu, c, wn, l2, True, np.float32, vec4, zchunk, small, variant=variant)
        assert np.isfinite(got).all(), variant
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0), variant


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,zchunk,small", [
    ((16, 9, 7), ("db1", "db3", "db1"), True, 0, True),
    ((20, 17, 12), ("db4", "db4", "db4"), True, 5, True),          # chunks of 5: leading zero iterations, a middle chunk, trailing ones
    ((20, 17, 12), ("db4", "db4", "db4"), True, 2, True),          # chunks shorter than the filter: every chunk starts or ends in zeros
    ((24, 19, 4), ("db6", "db2", "db4"), True, 0, True),           # a slab thinner than the filter (cfg5's regime)
    ((21, 14, 12), ("db5", "db3", "db1"), False, 5, True),
    ((68, 39, 9), ("db4", "db4", "db4"), True, 4, False),          # production tile
    ((72, 37, 10), ("db6", "db6", "db6"), True, 0, False),
    ((24, 22, 9), ("db10", "db10", "db10"), True, 6, True),        # pending z sums in LDS
])
def test_emulated_pair_packed_synthesis_of_a_zero_extended_slab(emu, sizes, wn, vec4, zchunk, small):
    """Inv3Y with z_wrap = 3 (the scatter-add synthesis of a slab: the coefficient planes outside the slab read as zero): the march
    skips the iterations before the first plane of the slab and only emits the pending sums after the last one -- same numbers as
    the periodic oracle on the zero-padded slab, and the planes past the slab are never read (they hold NaN here)"""
    rng = np.random.default_rng(21)
    n1, n2, n_in = sizes
    filt = [orc.wave_filters(w) for w in wn]
    L = max(len(f[0]) for f in filt)
    c = rng.standard_normal((n1, n2, n_in, 8))
    pad = L - 1
    cp = np.zeros((n1, n2, n_in + 2 * pad, 8))
    cp[:, :, pad:pad + n_in] = c
    ftz = [filt[0], filt[1], (np.pad(filt[2][0], ((L - len(filt[2][0])) // 2,) * 2), np.pad(filt[2][1], ((L - len(filt[2][1])) // 2,) * 2))]
    full = orc.spatial_level_rec(cp, ftz, 1)                       # periodic, but the support never reaches the wrap
    sa = L // 2 - 1
    want = full[:, :, pad - sa:pad - sa + n_in + L - 1]            # planes [0, sa): owed to the slab before, the last L/2: to the one after
    src = np.full((n1, n2, n_in + L - 1, 8), np.nan)
    src[:, :, :n_in] = c
    for variant in ((5, 8) if vec4 else (5,)):
        got = _run(emu, src, wn, 1, True, np.float32, vec4, zchunk, small, z_wrap=3, variant=variant)
        assert np.isfinite(got).all(), variant
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0), variant


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,zchunk,small", [
    ((16, 9, 7), ("db1", "db3", "db1"), 0, True),
    ((20, 17, 12), ("db4", "db4", "db4"), 5, True),               # partial sums travel one lane
    ((24, 19, 11), ("db6", "db2", "db4"), 0, True),               # ... two lanes (12 taps), mixed tap lengths padded to 12
    ((24, 21, 13), ("db5", "db5", "db5"), 0, True),
    ((24, 23, 19), ("db7", "db7", "db7"), 0, True),
    ((28, 20, 21), ("db8", "db8", "db4"), 7, True),
    ((20, 26, 20), ("db9", "db9", "db9"), 0, True),               # ... three lanes to the right (18 taps)
    ((24, 22, 21), ("db10", "db10", "db10"), 0, True),            # ... three lanes both ways (20 taps)
    ((68, 39, 9), ("db4", "db4", "db4"), 0, False),               # the library's instances on their production tiles: 8 taps,
    ((72, 37, 12), ("db6", "db6", "db6"), 0, False),              # 12 taps with the shared y / z tap pairs,
    ((72, 37, 12), ("db6", "db6", "db4"), 4, False),              # 12 taps, y and z taps apart,
    ((52, 30, 9), ("db10", "db10", "db10"), 0, False),            # 20 taps on the 48 x 28 tile
])
def test_emulated_pair_packed_synthesis_scatter_x_stage(emu, sizes, wn, zchunk, small):
    """Inv3Y<..., XSC>: the x stage in scatter form (partial sums travel between lanes instead of samples; the library's default for
    10 .. 20 taps on rows of whole groups of 4) -- periodic levels and one zero-extended slab per case"""
    rng = np.random.default_rng(33)
    c = rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    for l2 in (0, 1):
        want = orc.spatial_level_rec(c, filt, l2)
        got = _run(emu, c, wn, l2, True, np.float32, True, zchunk, small, variant=10)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0)
    # zero-extended slab (the scatter-add synthesis of the sharded drivers), as in the test above
    n1, n2, n_in = sizes
    L = max(len(f[0]) for f in filt)
    pad = L - 1
    cp = np.zeros((n1, n2, n_in + 2 * pad, 8))
    cp[:, :, pad:pad + n_in] = c
    ftz = [filt[0], filt[1], (np.pad(filt[2][0], ((L - len(filt[2][0])) // 2,) * 2), np.pad(filt[2][1], ((L - len(filt[2][1])) // 2,) * 2))]
    full = orc.spatial_level_rec(cp, ftz, 1)
    sa = L // 2 - 1
    want = full[:, :, pad - sa:pad - sa + n_in + L - 1]
    src = np.full((n1, n2, n_in + L - 1, 8), np.nan)
    src[:, :, :n_in] = c
    got = _run(emu, src, wn, 1, True, np.float32, True, zchunk, small, z_wrap=3, variant=10)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0)


CASES_YC = [
    ((10, 9, 7), ("db1", "db3", "db1"), True, 0),
    ((12, 10, 9), ("db2", "db2", "db2"), False, 4),
    ((14, 17, 12), ("db4", "db4", "db4"), True, 5),
    ((18, 13, 11), ("db5", "db3", "db1"), True, 0),
    ((20, 15, 13), ("db6", "db6", "db6"), True, 6),              # 6 halo groups of two elements each
    ((13, 12, 12), ("db6", "db2", "db4"), False, 0),
    ((67, 35, 8), ("db4", "db4", "db4"), False, 0),              # 134 scalars per row: the last lane of a row straddles the wrap
    ((22, 17, 15), ("db7", "db7", "db7"), True, 0),              # 14 / 16 taps (fused for complex64 since round 3)
    ((24, 18, 17), ("db8", "db6", "db4"), True, 7),
    ((21, 16, 16), ("db8", "db8", "db8"), False, 0),
]


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,zchunk", CASES_YC)
@pytest.mark.parametrize("l2", [0, 1])
def test_emulated_pair_packed_synthesis_complex(emu, sizes, wn, vec4, zchunk, l2):
    """Inv3Y on interleaved complex data (EW = 2): a pair is the (re, im) of one element and takes one tap"""
    rng = np.random.default_rng(13)
    c = rng.standard_normal(tuple(sizes) + (8,)) + 1j * rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    want = orc.spatial_level_rec(c, filt, l2)
    for variant in ((5, 8, 10) if vec4 else (5,)):     # 10: the x stage in scatter form (the library's default from 10 taps on)
        got = _run(emu, c, wn, l2, True, np.float32, vec4, zchunk, True, variant=variant, cplx=True)
        assert np.isfinite(got).all(), variant
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0), variant


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,zchunk", [((70, 36, 9), ("db5", "db5", "db5"), 0), ((50, 35, 8), ("db6", "db6", "db4"), 4), ((52, 33, 10), ("db8", "db8", "db8"), 0)])
def test_emulated_pair_packed_synthesis_complex_scatter_form_production_tiles(emu, sizes, wn, zchunk):
    """Inv3Y<.., EW = 2, XSC> on the library's tiles (64 x 32 for 10 taps, 48 x 32 for 12 .. 16): interleaved complex64 data"""
    rng = np.random.default_rng(14)
    c = rng.standard_normal(tuple(sizes) + (8,)) + 1j * rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    want = orc.spatial_level_rec(c, filt, 1)
    got = _run(emu, c, wn, 1, True, np.float32, True, zchunk, False, variant=10, cplx=True)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0)


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,nlev,ychunk", [
    ((64, 40), ("db4", "db4"), 3, 0),            # one wave, one chunk
    ((64, 57), ("db4", "db4"), 3, 13),           # chunks shorter than the march-in (21 rows): every level starts in another chunk's rows
    ((260, 31), ("db2", "db3"), 2, 8),           # two waves along x (the second mostly outside the image), mixed wavelets
    ((244, 26), ("db4", "db1"), 3, 0),           # 232 columns per wave with 8 taps: a 12-column second tile
    ((64, 45), ("db6", "db6"), 2, 20),           # 12 taps: halo of two lanes per level on the right
    ((32, 30), ("db1", "db1"), 3, 7),
])
@pytest.mark.parametrize("l2", [0, 1])
def test_emulated_cascaded_2d_analysis(emu, sizes, wn, nlev, ychunk, l2):
    """Fwd2C: two or three levels of an image in one march (the approximations between the levels never leave the registers) -- the
    same bands as `nlev` levels of the oracle, in the reference's band order (nd_dwt_2D.m:141-197)"""
    rng = np.random.default_rng(31)
    x = rng.standard_normal(sizes)
    want = orc.spatial_dec(x, list(wn), nlev, l2)                          # (n1, n2, 1 + 3 nlev)
    Lp = max(len(orc.wave_filters(w)[0]) for w in wn)
    lo = np.zeros((3, 20))
    hi = np.zeros((3, 20))
    for ax in range(2):
        t = kernel_taps(wn[ax], l2, Lp)
        lo[ax, :Lp], hi[ax, :Lp] = t["ana_lo"], t["ana_hi"]
    src = to_kernel_order(x).astype(np.float32)
    n2, n1 = src.shape
    out = np.full((1 + 3 * nlev, n2, n1), np.nan, dtype=np.float32)
    emu.ndwt_emu2_cascade_f32.restype = ctypes.c_int
    rc = emu.ndwt_emu2_cascade_f32(Lp, nlev, src.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n1, n2, ychunk,
                                   lo.ctypes.data_as(ctypes.c_void_p), hi.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    got = np.transpose(out)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,nlev,ychunk,depth", [
    ((64, 40), ("db4", "db4"), 3, 0, 1),
    ((64, 57), ("db4", "db4"), 3, 13, 2),        # chunks shorter than the march-in; two rows of band loads in flight
    ((260, 31), ("db2", "db3"), 2, 8, 1),        # two waves along x, mixed wavelets
    ((244, 26), ("db4", "db2"), 3, 0, 1),
    ((64, 45), ("db3", "db3"), 2, 20, 2),
    ((32, 30), ("db1", "db1"), 3, 7, 1),
])
@pytest.mark.parametrize("l2", [0, 1])
def test_emulated_cascaded_2d_synthesis(emu, sizes, wn, nlev, ychunk, depth, l2):
    """Inv2C: two or three synthesis levels of an image in one march -- rec of arbitrary coefficients as the oracle computes it"""
    rng = np.random.default_rng(32)
    c = rng.standard_normal(tuple(sizes) + (1 + 3 * nlev,))
    want = orc.spatial_rec(c, list(wn), l2)
    Lp = max(len(orc.wave_filters(w)[0]) for w in wn)
    lo = np.zeros((3, 20))
    hi = np.zeros((3, 20))
    for ax in range(2):
        t = kernel_taps(wn[ax], l2, Lp)
        lo[ax, :Lp], hi[ax, :Lp] = t["syn_lo"], t["syn_hi"]
    src = to_kernel_order(c).astype(np.float32)
    n2, n1 = src.shape[1:]
    out = np.full((n2, n1), np.nan, dtype=np.float32)
    emu.ndwt_emu2_cascade_inv_f32.restype = ctypes.c_int
    rc = emu.ndwt_emu2_cascade_inv_f32(Lp, nlev, depth, src.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n1, n2, ychunk,
                                       lo.ctypes.data_as(ctypes.c_void_p), hi.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    got = np.transpose(out)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 4e-6 * max(np.abs(want).max(), 1.0)


def _run2(emu, arr, wnames, l2, inverse, dtype, vec4, ychunk, cplx=False, shrink=(0.0, 0, 0), dil=1):
    Ls = [len(orc.wave_filters(w)[0]) for w in wnames]
    Lp = max(Ls)
    lo = np.zeros((3, 20))
    hi = np.zeros((3, 20))
    for ax in range(2):
        t = kernel_taps(wnames[ax], l2, Lp)
        lo[ax, :Lp] = t["syn_lo" if inverse else "ana_lo"]
        hi[ax, :Lp] = t["syn_hi" if inverse else "ana_hi"]
    cdt = (np.complex64 if dtype == np.float32 else np.complex128) if cplx else dtype
    src = to_kernel_order(arr).astype(cdt)
    if inverse:
        n2, n1 = src.shape[1:]
        out = np.full((n2, n1), np.nan, dtype=cdt)
    else:
        n2, n1 = src.shape
        out = np.full((4, n2, n1), np.nan, dtype=cdt)
    if cplx:
        n1 *= 2
    fn = emu.ndwt_emu2_f32 if dtype == np.float32 else emu.ndwt_emu2_f64
    fn.restype = ctypes.c_int
    rc = fn(int(inverse), Lp, int(vec4), src.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n1, n2, ychunk,
            lo.ctypes.data_as(ctypes.c_void_p), hi.ctypes.data_as(ctypes.c_void_p), 1, 2 if cplx else 1, ctypes.c_double(shrink[0]),
            int(shrink[1]), int(shrink[2]), int(dil))
    assert rc == 0
    return np.transpose(out)


CASES2 = [
    # sizes (n1,n2), wavelets, vec4, ychunk
    ((40, 13), ("db1", "db3"), True, 0),
    ((255, 9), ("db4", "db2"), False, 4),      # more than one wave tile along x, odd width
    ((516, 20), ("db4", "db4"), True, 7),      # three wave tiles, several row chunks
    ((36, 30), ("db6", "db5"), True, 11),      # two-lane shifts
    ((30, 12), ("db2", "db6"), False, 0),
    ((301, 9), ("db4", "db4"), False, 0),      # two wave tiles, the second anchored at the end of the row (ndwt_device.h: tile_origin)
    ((250, 8), ("db3", "db4"), False, 3),      # a second tile of 2 columns that cannot be anchored (the row is shorter than a tile + halo)
]


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,ychunk", CASES2)
@pytest.mark.parametrize("l2", [0, 1])
def test_emulated_fused2(emu, sizes, wn, vec4, ychunk, l2):
    """register-only 2-D kernels (lane-shift x filter): one level against nd_dwt_2D.m:312-337 semantics"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(sizes)
    c = rng.standard_normal(tuple(sizes) + (4,))
    filt = [orc.wave_filters(w) for w in wn]
    want_y = orc.spatial_level_dec(x, filt, l2)
    want_r = orc.spatial_level_rec(c, filt, l2)
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        got = _run2(emu, x, wn, l2, False, dtype, vec4, ychunk)
        assert np.isfinite(got).all() and np.abs(got - want_y).max() <= tol * np.abs(want_y).max()
        got = _run2(emu, c, wn, l2, True, dtype, vec4, ychunk)
        assert np.isfinite(got).all() and np.abs(got - want_r).max() <= tol * max(np.abs(want_r).max(), 1.0)


MARCH = [
    # (outer, n, inner), wavelet, chunk
    ((3, 13, 8), "db1", 0),
    ((2, 20, 12), "db4", 7),
    ((1, 37, 1028), "db2", 10),      # more than one 256-thread block along the contiguous run
    ((2, 24, 4), "db6", 0),
    ((1, 45, 16), "db10", 9),
]


@pytest.mark.slow
@pytest.mark.parametrize("shape,wn,chunk", MARCH)
def test_emulated_axis_march(emu, shape, wn, chunk):
    """register-window kernels for one non-contiguous axis (outer axis of 4-D volumes, per-axis path)"""
    outer, n, inner = shape
    rng = np.random.default_rng(5)
    lo_d, hi_d = orc.wave_filters(wn)
    L = len(lo_d)
    t = kernel_taps(wn, 1)
    x = rng.standard_normal(shape)
    a_in, d_in = rng.standard_normal(shape), rng.standard_normal(shape)
    want_lo, want_hi = orc._analysis_axis(x, lo_d, hi_d, 1, 1 / np.sqrt(2.0), 1)
    want_r = orc._synthesis_axis(a_in, d_in, lo_d, hi_d, 1, 1 / np.sqrt(2.0), 1)
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        fn = emu.ndwt_emu_march_f32 if dtype == np.float32 else emu.ndwt_emu_march_f64
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_longlong] * 3 + [ctypes.c_int, ctypes.c_int,
                                                                                                   ctypes.c_void_p, ctypes.c_void_p]
        P = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
        xin = np.ascontiguousarray(x, dtype=dtype)
        lo = np.full(shape, np.nan, dtype=dtype)
        hi = np.full(shape, np.nan, dtype=dtype)
        tl, th = np.ascontiguousarray(t["ana_lo"]), np.ascontiguousarray(t["ana_hi"])
        assert fn(0, L, P(xin), None, P(lo), P(hi), inner, n, outer, chunk, 1, P(tl), P(th)) == 0
        assert np.abs(lo - want_lo).max() <= tol * np.abs(want_lo).max() and np.abs(hi - want_hi).max() <= tol * np.abs(want_hi).max()
        ai, di = np.ascontiguousarray(a_in, dtype=dtype), np.ascontiguousarray(d_in, dtype=dtype)
        r = np.full(shape, np.nan, dtype=dtype)
        tl, th = np.ascontiguousarray(t["syn_lo"]), np.ascontiguousarray(t["syn_hi"])
        assert fn(1, L, P(ai), P(di), P(r), None, inner, n, outer, chunk, 1, P(tl), P(th)) == 0
        assert np.abs(r - want_r).max() <= tol * np.abs(want_r).max()


AXISX = [
    # (outer, n), wavelet, complex?, vec4
    ((3, 40), "db1", False, True),
    ((2, 300), "db4", False, True),       # two wave segments per row
    ((2, 37), "db4", False, False),       # odd length: scalar loads/stores
    ((3, 50), "db4", True, True),         # interleaved complex: taps step over (re, im) pairs, two-lane shifts
    ((1, 131), "db6", True, False),
    ((2, 541), "db4", False, False),      # three wave segments, the last anchored at the end of the row
    ((1, 277), "db1", True, False),
    ((2, 64), "db6", False, True),
]


@pytest.mark.slow
@pytest.mark.parametrize("shape,wn,cplx,vec4", AXISX)
def test_emulated_contiguous_axis(emu, shape, wn, cplx, vec4):
    """lane-shift kernel for the contiguous axis (1-D signals, interleaved complex data)"""
    outer, n = shape
    rng = np.random.default_rng(6)
    lo_d, hi_d = orc.wave_filters(wn)
    L = len(lo_d)
    t = kernel_taps(wn, 1)
    mk = (lambda: rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) if cplx else (lambda: rng.standard_normal(shape))
    x, a_in, d_in = mk(), mk(), mk()
    want_lo, want_hi = orc._analysis_axis(x, lo_d, hi_d, 1, 1 / np.sqrt(2.0), 1)
    want_r = orc._synthesis_axis(a_in, d_in, lo_d, hi_d, 1, 1 / np.sqrt(2.0), 1)
    ew = 2 if cplx else 1
    for dtype, cdtype, tol in ((np.float64, np.complex128, 1e-13), (np.float32, np.complex64, 2e-6)):
        fn = emu.ndwt_emu_axisx_f32 if dtype == np.float32 else emu.ndwt_emu_axisx_f64
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4 + [ctypes.c_longlong] * 2 + [ctypes.c_void_p] * 2
        P = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
        cast = (lambda v: np.ascontiguousarray(v, dtype=cdtype)) if cplx else (lambda v: np.ascontiguousarray(v, dtype=dtype))
        new = lambda: np.full(shape, np.nan, dtype=cdtype if cplx else dtype)
        xin, lo, hi = cast(x), new(), new()
        tl, th = np.ascontiguousarray(t["ana_lo"]), np.ascontiguousarray(t["ana_hi"])
        assert fn(0, L, ew, int(vec4), P(xin), None, P(lo), P(hi), n * ew, outer, P(tl), P(th)) == 0
        assert np.abs(lo - want_lo).max() <= tol * np.abs(want_lo).max() and np.abs(hi - want_hi).max() <= tol * np.abs(want_hi).max()
        ai, di, r = cast(a_in), cast(d_in), new()
        tl, th = np.ascontiguousarray(t["syn_lo"]), np.ascontiguousarray(t["syn_hi"])
        assert fn(1, L, ew, int(vec4), P(ai), P(di), P(r), None, n * ew, outer, P(tl), P(th)) == 0
        assert np.abs(r - want_r).max() <= tol * np.abs(want_r).max()


CPLX3 = [
    ((10, 9, 7), ("db1", "db3", "db2"), False, 0, True),
    ((24, 17, 12), ("db4", "db4", "db4"), True, 5, True),
    ((70, 19, 10), ("db4", "db2", "db4"), True, 6, False),     # production tiles (analysis 64x16, synthesis 64x32)
    ((22, 15, 13), ("db5", "db5", "db5"), True, 0, True),      # 10 / 12 taps over (re, im) pairs: 5- and 6-group halos
    ((26, 14, 12), ("db6", "db3", "db6"), True, 5, True),
]


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,zchunk,small", CPLX3)
def test_emulated_fused3_interleaved_complex(emu, sizes, wn, vec4, zchunk, small):
    """fused 3-D kernels on interleaved complex data (x taps step over (re, im) pairs)"""
    rng = np.random.default_rng(8)
    x = rng.standard_normal(sizes) + 1j * rng.standard_normal(sizes)
    c = rng.standard_normal(tuple(sizes) + (8,)) + 1j * rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    want_y = orc.spatial_level_dec(x, filt, 1)
    want_r = orc.spatial_level_rec(c, filt, 1)
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        got = _run(emu, x, wn, 1, False, dtype, vec4, zchunk, small, cplx=True)
        assert np.isfinite(got).all() and np.abs(got - want_y).max() <= tol * np.abs(want_y).max()
        got = _run(emu, c, wn, 1, True, dtype, vec4, zchunk, small, cplx=True)
        assert np.isfinite(got).all() and np.abs(got - want_r).max() <= tol * np.abs(want_r).max()


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,vec4,ychunk", [((40, 13), ("db1", "db2"), True, 0), ((150, 20), ("db4", "db4"), True, 7),
                                                    ((33, 12), ("db4", "db1"), False, 0)])
def test_emulated_fused2_interleaved_complex(emu, sizes, wn, vec4, ychunk):
    rng = np.random.default_rng(9)
    x = rng.standard_normal(sizes) + 1j * rng.standard_normal(sizes)
    c = rng.standard_normal(tuple(sizes) + (4,)) + 1j * rng.standard_normal(tuple(sizes) + (4,))
    filt = [orc.wave_filters(w) for w in wn]
    want_y = orc.spatial_level_dec(x, filt, 0)
    want_r = orc.spatial_level_rec(c, filt, 0)
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 2e-6)):
        got = _run2(emu, x, wn, 0, False, dtype, vec4, ychunk, cplx=True)
        assert np.isfinite(got).all() and np.abs(got - want_y).max() <= tol * np.abs(want_y).max()
        got = _run2(emu, c, wn, 0, True, dtype, vec4, ychunk, cplx=True)
        assert np.isfinite(got).all() and np.abs(got - want_r).max() <= tol * max(np.abs(want_r).max(), 1.0)


def _np_shrink_bands(c, t, hard, mask):
    m = np.abs(c)
    with np.errstate(invalid="ignore", divide="ignore"):
        g = np.where(m > t, 1.0 if hard else (m - t) / np.where(m > 0, m, 1.0), 0.0)
    out = c * g
    for b in range(c.shape[-1]):
        if not (mask >> b) & 1:
            out[..., b] = c[..., b]
    return out


@pytest.mark.slow
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("hard", [0, 1])
def test_emulated_synthesis_with_fused_shrinkage(emu, cplx, hard):
    """the lane-shift synthesis kernels threshold the bands selected by shrink_mask as they load them (ndwt_denoise)"""
    rng = np.random.default_rng(12)
    # 3-D: production tile (variant 1) and small tile (variant 2)
    for sizes, wn, vec4, zchunk, small, variant in (((68, 18, 9), ("db4", "db2", "db4"), True, 0, False, 1),
                                                    ((20, 9, 7), ("db2", "db3", "db1"), True, 4, True, 2)):
        c = rng.standard_normal(tuple(sizes) + (8,)) + (1j * rng.standard_normal(tuple(sizes) + (8,)) if cplx else 0)
        filt = [orc.wave_filters(w) for w in wn]
        want = orc.spatial_level_rec(_np_shrink_bands(c, 0.5, hard, 0xFE), filt, 1)
        got = _run(emu, c, wn, 1, True, np.float64, vec4, zchunk, small, variant=variant, cplx=cplx, shrink=(0.5, 0xFE, hard))
        assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0)
    # 2-D
    sizes, wn = (152, 20), ("db4", "db2")             # vec4 path: rows are a multiple of 4 scalars
    c = rng.standard_normal(tuple(sizes) + (4,)) + (1j * rng.standard_normal(tuple(sizes) + (4,)) if cplx else 0)
    filt = [orc.wave_filters(w) for w in wn]
    want = orc.spatial_level_rec(_np_shrink_bands(c, 0.5, hard, 0xE), filt, 1)
    got = _run2(emu, c, wn, 1, True, np.float64, True, 7, cplx=cplx, shrink=(0.5, 0xE, hard))
    assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0)


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,dil,dtype", [
    ((24, 12, 8), ("db4", "db2", "db3"), 2, np.float64),
    ((72, 16, 6), ("db2", "db4", "db1"), 2, np.float32),
    ((72, 16, 8), ("db4", "db3", "db2"), 4, np.float32),        # stride 4: the 512-thread tiles, x taps over 4 scalars
])
def test_emulated_dilated_level_on_sublattices(emu, sizes, wn, dil, dtype):
    """an a-trous level (tap stride = dil on every axis) as dil^3 independent stride-1 problems: x through EW = dil,
    the (y, z) sub-lattices as batch items with strided rows and planes"""
    rng = np.random.default_rng(21)
    x = rng.standard_normal(sizes)
    c = rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    tol = 1e-13 if dtype == np.float64 else 2e-6
    want_y = orc.spatial_level_dec(x, filt, 1, dil)
    got = _run(emu, x, wn, 1, False, dtype, True, 0, False, dil=dil)
    assert np.isfinite(got).all() and np.abs(got - want_y).max() <= tol * np.abs(want_y).max()
    want_r = orc.spatial_level_rec(c, filt, 1, dil)
    got = _run(emu, c, wn, 1, True, dtype, True, 0, False, dil=dil)
    assert np.isfinite(got).all() and np.abs(got - want_r).max() <= tol * max(np.abs(want_r).max(), 1.0)


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,dil", [
    ((136, 68, 6), ("db4", "db4", "db4"), 2),          # 68 x 34 per sub-lattice: ragged production tiles
    ((72, 16, 8), ("db2", "db2", "db2"), 2),
    ((264, 104, 8), ("db4", "db2", "db4"), 4),         # 8 taps at stride 4: 64 x 24 tiles, 23 lanes per haloed row
    ((72, 16, 8), ("db2", "db2", "db2"), 4),
    ((64, 12, 16), ("db3", "db1", "db3"), 4),
    ((80, 8, 12), ("db1", "db1", "db1"), 4),
])
def test_emulated_dilated_synthesis_on_the_pair_packed_kernel(emu, sizes, wn, dil):
    """float synthesis of an a-trous level through Inv3Y: tap stride 2 = its interleaved-pair form (the two x sub-lattices are the
    (re, im) halves), tap stride 4 = whole-lane x shifts (EW = 4, one tap for both pairs of a lane); one and two register sets"""
    rng = np.random.default_rng(24)
    c = rng.standard_normal(tuple(sizes) + (8,))
    filt = [orc.wave_filters(w) for w in wn]
    want = orc.spatial_level_rec(c, filt, 1, dil)
    L = max(len(f[0]) for f in filt)
    # (variant 10 at tap stride 4: the x stage in scatter form -- the sums walk from lane to lane -- the library's default for 8 taps)
    for variant in ((5, 8) if L in (2, 8) else (5,)) + ((10,) if dil == 4 and L >= 4 else ()):
        got = _run(emu, c, wn, 1, True, np.float32, True, 0, False, variant=variant, dil=dil)
        assert np.isfinite(got).all(), variant
        assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0), variant


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wn,dtype", [((72, 20), ("db4", "db2"), np.float64), ((264, 14), ("db2", "db4"), np.float32)])
def test_emulated_dilated_2d_level_on_row_sublattices(emu, sizes, wn, dtype):
    """2-D a-trous level with tap stride 2: x through EW = 2, the two row sub-lattices as batch items (row stride 2*n1)"""
    rng = np.random.default_rng(23)
    x = rng.standard_normal(sizes)
    c = rng.standard_normal(tuple(sizes) + (4,))
    filt = [orc.wave_filters(w) for w in wn]
    tol = 1e-13 if dtype == np.float64 else 2e-6
    want_y = orc.spatial_level_dec(x, filt, 0, 2)
    got = _run2(emu, x, wn, 0, False, dtype, True, 0, dil=2)
    assert np.isfinite(got).all() and np.abs(got - want_y).max() <= tol * np.abs(want_y).max()
    want_r = orc.spatial_level_rec(c, filt, 0, 2)
    got = _run2(emu, c, wn, 0, True, dtype, True, 5, dil=2)
    assert np.isfinite(got).all() and np.abs(got - want_r).max() <= tol * max(np.abs(want_r).max(), 1.0)


def _taps3(wname, l2, L):
    t = kernel_taps(wname, l2, L)
    arr = {k: np.zeros((3, 20)) for k in t}
    for k in t:
        arr[k][:, :L] = t[k]
    return arr


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wname,zchunk", [
    ((68, 36, 9), "db4", 0),          # ragged in x and y: 2 x 2 production tiles
    ((72, 33, 20), "db4", 7),         # several z chunks
    ((64, 32, 5), "db3", 0),
    ((12, 10, 6), "db2", 0),          # a volume smaller than the halo: every axis wraps more than once
    ((76, 40, 7), "db1", 3),
])
@pytest.mark.parametrize("l2,hard", [(1, 0), (0, 1)])
def test_emulated_fused_level1_denoise(emu, sizes, wname, zchunk, l2, hard):
    """Den3: level 1 of dec -> shrink -> rec in one launch -- the detail bands are recomputed from x on the haloed tile, thresholded in
    registers and synthesised together with a given approximation band (reference: one level of nd_dwt_3D.m:345-374 each way)"""
    rng = np.random.default_rng(31)
    x = rng.standard_normal(sizes)
    apx = rng.standard_normal(sizes)
    filt = [orc.wave_filters(wname)] * 3
    c = orc.spatial_level_dec(x, filt, l2)
    c = _np_shrink_bands(c, 0.6, hard, 0xFE)
    c[..., 0] = apx
    want = orc.spatial_level_rec(c, filt, l2)
    L = len(filt[0][0])
    t = _taps3(wname, l2, L)
    xs, aps = to_kernel_order(x).astype(np.float32), to_kernel_order(apx).astype(np.float32)
    n3, n2, n1 = xs.shape
    out = np.full((n3, n2, n1), np.nan, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = emu.ndwt_emu_den3_f32(L, p(xs), p(aps), p(out), n1, n2, n3, zchunk, p(t["syn_lo"]), p(t["syn_hi"]), p(t["ana_lo"]), p(t["ana_hi"]),
                               ctypes.c_double(0.6), hard)
    assert rc == 0
    got = np.transpose(out)
    assert np.isfinite(got).all()
    # hard thresholding is discontinuous: a coefficient within rounding of the threshold may fall on the other side in fp32
    bad = np.abs(got - want) > 4e-6 * max(np.abs(want).max(), 1.0)
    assert bad.mean() <= (2e-3 if hard else 0.0), float(np.abs(got - want).max())


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wname,vec4,zchunk", [((68, 36, 9), "db4", True, 0), ((70, 33, 12), "db3", False, 5), ((64, 32, 6), "db1", True, 0)])
def test_emulated_approximation_only_analysis(emu, sizes, wname, vec4, zchunk):
    """Fwd3<.., LOWONLY>: band 0 of one analysis level, the other seven neither computed through nor stored"""
    rng = np.random.default_rng(32)
    x = rng.standard_normal(sizes)
    filt = [orc.wave_filters(wname)] * 3
    want = orc.spatial_level_dec(x, filt, 1)[..., 0]
    L = len(filt[0][0])
    t = _taps3(wname, 1, L)
    xs = to_kernel_order(x).astype(np.float32)
    n3, n2, n1 = xs.shape
    out = np.full((n3, n2, n1), np.nan, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert emu.ndwt_emu_low3_f32(L, int(vec4), p(xs), p(out), n1, n2, n3, zchunk, p(t["ana_lo"]), p(t["ana_hi"])) == 0
    got = np.transpose(out)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wnames,zchunk", [((68, 36, 9), ("db6",) * 3, 0), ((72, 40, 16), ("db5",) * 3, 6), ((64, 32, 5), ("db7",) * 3, 0),
                                                 ((128, 33, 7), ("db6", "db2", "db4"), 3), ((16, 12, 14), ("db6",) * 3, 0),
                                                 ((68, 36, 9), ("db8",) * 3, 0), ((72, 40, 20), ("db8", "db4", "db8"), 7),
                                                 ((68, 20, 22), ("db10",) * 3, 0), ((64, 33, 24), ("db10", "db6", "db8"), 9)])
def test_emulated_analysis_with_pinned_taps(emu, sizes, wnames, zchunk):
    """16 / 20 taps: Fwd3<.., WLDS = 2 / 4> -- that many slots of every thread's z window in LDS, the others in registers, plain taps
    (20 taps: the 512-thread 64 x 16 tile with two columns per thread).
    Fwd3<.., PIN> (float real, 10 / 12 / 14 taps on vec4 data): tap pairs pinned in scalar registers, the high-pass taps taken from
    the low-pass pairs through the operand modifiers of the packed FMA (mirror + alternating signs, which survives an EVEN zero padding
    of a shorter axis' taps) -- all 8 bands against the oracle"""
    rng = np.random.default_rng(34)
    x = rng.standard_normal(sizes)
    filt = [orc.wave_filters(w) for w in wnames]
    want = orc.spatial_level_dec(x, filt, 1)
    L = max(len(f[0]) for f in filt)
    t = {k: np.zeros((3, 20)) for k in ("ana_lo", "ana_hi")}
    for ax, w in enumerate(wnames):
        assert ((L - len(filt[ax][0])) // 2) % 2 == 0
        ka = kernel_taps(w, 1, L)
        for k in t:
            t[k][ax, :L] = ka[k]
    xs = to_kernel_order(x).astype(np.float32)
    n3, n2, n1 = xs.shape
    out = np.full((8, n3, n2, n1), np.nan, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert emu.ndwt_emu_pin3_f32(L, p(xs), p(out), n1, n2, n3, zchunk, p(t["ana_lo"]), p(t["ana_hi"])) == 0
    got = np.transpose(out)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wname,zchunk", [((68, 36, 6, 5), "db4", 0), ((64, 32, 9, 3), "db2", 4), ((72, 33, 5, 9), "db1", 0)])
def test_emulated_4d_analysis_with_folded_t_axis(emu, sizes, wname, zchunk):
    """Fwd3<.., TPRE>: a 4-D analysis level in two launches (one per t-band) whose raw planes are the t-filtered combination of L
    frames -- no pass of its own over the data for the t axis (reference: level_1_dec of nd_dwt_4D.m:394-467)"""
    rng = np.random.default_rng(33)
    x = rng.standard_normal(sizes)
    filt = [orc.wave_filters(wname)] * 4
    want = orc.spatial_level_dec(x, filt, 1)
    L = len(filt[0][0])
    t = _taps3(wname, 1, L)
    tl, th = np.ascontiguousarray(t["ana_lo"][0]), np.ascontiguousarray(t["ana_hi"][0])
    xs = to_kernel_order(x).astype(np.float32)
    n4, n3, n2, n1 = xs.shape
    out = np.full((16, n4, n3, n2, n1), np.nan, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert emu.ndwt_emu_tpre_f32(L, p(xs), p(out), n1, n2, n3, n4, zchunk, p(t["ana_lo"]), p(t["ana_hi"]), p(tl), p(th)) == 0
    got = np.transpose(out)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.slow
@pytest.mark.parametrize("sizes,wname,ychunk,shrink", [((256, 21), "db4", 0, None), ((500, 13), "db4", 5, (0.4, 0xE, 0)), ((252, 9), "db3", 0, None),
                                                        ((248, 30), "db6", 11, (0.3, 0xE, 1)), ((40, 6), "db1", 2, None), ((260, 4), "db2", 0, None)])
@pytest.mark.parametrize("depth", [2, 4, 14])
def test_emulated_fused2_synthesis_with_rows_in_flight(emu, sizes, wname, ychunk, shrink, depth):
    """Inv2P: 2 or 4 rows of band loads in flight per wave, the row loop unrolled in groups of L (rotation and slot of every row compile-time
    constants), including chunks shorter than a group and the thresholding of the row about to be consumed.  depth 14 = 4 rows in flight
    in the packed form (pairs of adjacent x outputs per packed FMA, (t[k], t[k-1]) tap pairs; 4 / 8 / 12 taps)"""
    if depth == 14 and len(orc.wave_filters(wname)[0]) not in (4, 8, 12):
        pytest.skip("packed form: 4, 8 and 12 taps")
    rng = np.random.default_rng(34)
    c = rng.standard_normal(tuple(sizes) + (4,))
    filt = [orc.wave_filters(wname)] * 2
    cs = _np_shrink_bands(c, shrink[0], shrink[2], shrink[1]) if shrink else c
    want = orc.spatial_level_rec(cs, filt, 1)
    L = len(filt[0][0])
    t = _taps3(wname, 1, L)
    src = to_kernel_order(c).astype(np.float32)
    n2, n1 = src.shape[1:]
    out = np.full((n2, n1), np.nan, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    sh = shrink or (0.0, 0, 0)
    assert emu.ndwt_emu2_inv2p_f32(L, depth, p(src), p(out), n1, n2, ychunk, p(t["syn_lo"]), p(t["syn_hi"]), ctypes.c_double(sh[0]), sh[1], sh[2]) == 0
    got = np.transpose(out)
    assert np.isfinite(got).all()
    bad = np.abs(got - want) > 4e-6 * max(np.abs(want).max(), 1.0)
    assert bad.mean() <= (2e-3 if shrink and shrink[2] else 0.0), float(np.abs(got - want).max())
