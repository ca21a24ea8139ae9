"""GPU parity tests: the HIP engine (through the C ABI and the nd_dwt_{1,2,3,4}D mirror classes) against the
CPU oracle, the golden fixtures and -- at full size -- size-independent properties.

Tolerances (BASELINE.md section 2 / SURVEY.md 8c): fp64 <= 1e-12 relative to max|c|; fp32 <= 2e-6 relative to
max|c|; fwd+inv round trip < 1e-6 as a relative l2 norm.
"""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

import helpers
import ndwt_amd as ndwt
import ndwt_oracle as orc

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
TOL = {"double": 1e-12, "single": 2e-6}


def _cls(d):
    return {1: ndwt.nd_dwt_1D, 2: ndwt.nd_dwt_2D, 3: ndwt.nd_dwt_3D, 4: ndwt.nd_dwt_4D}[d]


def _relerr(got, want):
    return float(np.abs(np.asarray(got) - want).max() / max(np.abs(want).max(), 1e-300))


def _colmajor_gpu(a, dtype):
    """numpy MATLAB-shaped array -> GPU tensor of the same shape, column-major memory."""
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(a))).cuda()
    if a.dtype.kind == "c":
        t = t.to(torch.complex64 if dtype == "single" else torch.complex128)
    else:
        t = t.to(torch.float32 if dtype == "single" else torch.float64)
    return t.permute(*reversed(range(t.dim())))


def test_extension_is_loaded_and_reports_fused_path():
    assert os.path.exists(ndwt.LIB_PATH)
    w = ndwt.nd_dwt_3D("db4", [64, 64, 64], precision="single")
    x = torch.randn(64, 64, 64, device="cuda")
    w.dec(x, 1)
    assert list(w._plans.values())[0].describe() == "fused3d"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("precision", ["double", "single"])
@pytest.mark.parametrize("generic", [False, True])
def test_golden_fixtures(path, precision, generic):
    g = np.load(path)
    sizes, level, wn = [int(s) for s in g["sizes"]], int(g["level"]), [str(w) for w in g["wname"]]
    d = len(sizes)
    wname = wn[0] if d == 1 else wn
    for tag, l2s in (("r", (0, 1)), ("c", (1,))):
        for l2 in l2s:
            w = _cls(d)(wname, sizes, "pres_l2_norm", l2, "precision", precision)
            x = _colmajor_gpu(g[f"x_{tag}"], precision)
            y = w.dec(x, level)
            for p in w._plans.values():
                p.set_path(generic)
            y = w.dec(x, level)
            assert tuple(y.shape) == tuple(sizes) + (orc.num_bands(d, level),)
            assert y.is_complex() == (tag == "c")
            assert _relerr(y.cpu().numpy(), g[f"y_{tag}_l2{l2}"]) <= TOL[precision]
            r = w.rec(_colmajor_gpu(g[f"c_{tag}"], precision))
            assert _relerr(r.cpu().numpy(), g[f"rec_{tag}_l2{l2}"]) <= TOL[precision]
    # a-trous mode (per-axis kernels with runtime tap stride)
    w = _cls(d)(wname, sizes, "precision", precision, "dilation", "atrous")
    y = w.dec(_colmajor_gpu(g["x_r"], precision), level)
    assert _relerr(y.cpu().numpy(), g["y_r_l20_atrous"]) <= TOL[precision]
    r = w.rec(_colmajor_gpu(g["c_r"], precision))
    assert _relerr(r.cpu().numpy(), g["rec_r_l20_atrous"]) <= TOL[precision]


# the reference's own test configurations (Test/nddwt{1,2,3,4}D_test.m:5-11): complex randn input, dec -> rec
REF_TESTS = [
    (1, [54321], "db1", 4, 0),
    (2, [264, 264], ["db1", "db3"], 1, 1),
    (3, [164, 64, 40], ["db1", "db3", "db1"], 1, 1),
    (4, [64, 64, 20, 10], ["db1", "db3", "db1", "db1"], 1, 1),
]


@pytest.mark.parametrize("d,sizes,wname,level,l2", REF_TESTS, ids=["1D", "2D", "3D", "4D"])
@pytest.mark.parametrize("precision", ["double", "single"])
def test_reference_test_scripts(d, sizes, wname, level, l2, precision):
    rng = np.random.default_rng(10 + d)
    x = rng.standard_normal(sizes) + 1j * rng.standard_normal(sizes)
    w = _cls(d)(wname, sizes[0] if d == 1 else sizes, "pres_l2_norm", l2, "precision", precision)
    xg = _colmajor_gpu(x, precision)
    y = w.dec(xg, level)
    r = w.rec(y)
    want = orc.NdDwtMat(wname, sizes, l2).dec(x, level)
    assert _relerr(y.cpu().numpy(), want) <= TOL[precision]                                    # (C) backends agree
    assert y.dtype == (torch.complex64 if precision == "single" else torch.complex128)         # (D) class follows precision
    xn = np.linalg.norm(x)
    if l2:
        assert abs(float(torch.linalg.vector_norm(y)) - xn) <= (1e-5 if precision == "single" else 1e-11) * xn   # (B)
    err = float(torch.linalg.vector_norm(r - xg)) / xn
    assert err < (1e-6 if precision == "single" else 1e-13)                                    # (A)


# the shapes of the reference's cross-backend scripts (mex/mex_test.m:11,48,84,120: 1-D n = 10000; 2-D 129 x 131; 3-D 131 x 128 x 30;
# 4-D 128 x 68 x 8 x 8 db3; complex randn input) and of Test/nddwt1D_test.m:5 -- rows that are not whole groups of 4 scalars
MEX_TESTS = [
    (1, [10000], "db1", 3, None),
    (1, [54321], "db2", 4, None),
    (2, [129, 131], "db1", 2, "fused2d"),
    (2, [129, 131], "db4", 2, "fused2d"),
    (3, [131, 128, 30], "db1", 2, "fused3d"),
    (3, [131, 128, 30], "db3", 2, "fused3d"),
    (4, [128, 68, 8, 8], "db3", 1, "axis+fused3d"),
]


@pytest.mark.parametrize("d,sizes,wname,level,path", MEX_TESTS)
@pytest.mark.parametrize("precision,cplx", [("double", True), ("single", True), ("single", False)])
def test_reference_mex_test_shapes(d, sizes, wname, level, path, precision, cplx):
    """odd row lengths (complex: an odd number of elements) run the fused kernels' VEC4 = false instances -- one access per lane
    wherever its 4 scalars are contiguous, end-of-row tiles anchored at the row end -- and agree with the oracle like any other shape"""
    rng = np.random.default_rng(60 + d)
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    w = _cls(d)(wname, sizes[0] if d == 1 else sizes, "pres_l2_norm", 1, "precision", precision)
    xg = _colmajor_gpu(x, precision)
    y = w.dec(xg, level)
    if path is not None:
        assert list(w._plans.values())[0].describe() == path
    want = orc.NdDwtMat(wname, sizes, 1).dec(x, level)
    assert _relerr(y.cpu().numpy(), want) <= TOL[precision]
    c = rng.standard_normal(want.shape) + (1j * rng.standard_normal(want.shape) if cplx else 0)     # rec of arbitrary coefficients
    r = w.rec(_colmajor_gpu(c, precision))
    want_r = orc.NdDwtMat(wname, sizes, 1).rec(c)
    assert _relerr(r.cpu().numpy(), want_r) <= 4 * TOL[precision]


RANDOM = [
    # d, sizes, wavelets, level
    (1, [97], "db5", 3),
    (1, [4096], "db2", 3),                       # BASELINE config 1 shape (fp64 db2 3 levels)
    (2, [70, 45], ["db2", "db4"], 3),
    (2, [128, 64], ["db4", "db4"], 2),
    (3, [37, 29, 23], ["db3", "db1", "db2"], 2),
    (3, [72, 40, 33], ["db4", "db4", "db4"], 3),
    (3, [64, 32, 48], ["db6", "db6", "db6"], 2),
    (3, [40, 40, 40], ["db8", "db2", "db2"], 2),  # longer than the fused instantiations -> per-axis path
    (4, [12, 9, 10, 11], ["db2", "db1", "db3", "db2"], 2),
    (4, [32, 32, 16, 16], ["db4", "db4", "db4", "db4"], 2),   # example_nd_dwt_4D.m:5 size
    # the long filters (tap lengths 14, 18, 20) on every axis position, and db5 (10 taps)
    (1, [333], "db7", 3),
    (1, [4096], "db10", 2),
    (2, [68, 40], ["db9", "db7"], 2),
    (2, [132, 24], ["db5", "db10"], 2),
    (3, [40, 36, 32], ["db7", "db7", "db7"], 2),
    (3, [72, 24, 40], ["db9", "db10", "db5"], 2),
    (3, [64, 40, 44], ["db2", "db3", "db10"], 2),
    (4, [20, 18, 16, 14], ["db1", "db2", "db1", "db7"], 2),
]


@pytest.mark.parametrize("d,sizes,wname,level", RANDOM)
@pytest.mark.parametrize("precision", ["double", "single"])
@pytest.mark.parametrize("l2", [0, 1])
def test_random_shapes_against_oracle(d, sizes, wname, level, precision, l2):
    rng = np.random.default_rng(20 + d)
    x = rng.standard_normal(sizes)
    w = _cls(d)(wname, sizes[0] if d == 1 else sizes, "pres_l2_norm", l2, "precision", precision)
    y = w.dec(_colmajor_gpu(x, precision), level)
    assert not y.is_complex()                                                                  # (E) real in -> real out
    want = orc.spatial_dec(x, wname, level, l2)
    assert _relerr(y.cpu().numpy(), want) <= TOL[precision]
    c = rng.standard_normal(list(sizes) + [orc.num_bands(d, level)])
    r = w.rec(_colmajor_gpu(c, precision))
    assert _relerr(r.cpu().numpy(), orc.spatial_rec(c, wname, l2)) <= TOL[precision]


FUZZ = helpers.fuzz_cases(64, seed=20261004)


@pytest.mark.parametrize("case", FUZZ, ids=[helpers.fuzz_id(c) for c in FUZZ])
def test_fuzz_subset_against_oracle(case):
    """fixed-seed subset of the randomised sweep of tools/fuzz_gpu.py: dec, rec of arbitrary coefficients, round trip"""
    d, sizes, wn, level, l2 = case["d"], case["sizes"], case["wn"], case["level"], case["l2"]
    precision, cplx, dilation = case["precision"], case["cplx"], case["dilation"]
    rng = np.random.default_rng(case["data_seed"])
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    w = _cls(d)(wn if d > 1 else wn[0], sizes, "pres_l2_norm", l2, "precision", precision, "dilation", dilation)
    xg = _colmajor_gpu(x, precision)
    y = w.dec(xg, level)
    want = orc.spatial_dec(x, wn, level, l2, dilation)
    tol = 3e-6 if precision == "single" else 1e-12             # the sweep's tolerance: worst case seen 0.14 of it
    assert _relerr(y.cpu().numpy(), want) <= tol
    assert _relerr(w.rec(y).cpu().numpy(), x) <= 20 * tol
    c = rng.standard_normal(want.shape) + (1j * rng.standard_normal(want.shape) if cplx else 0)
    r2 = w.rec(_colmajor_gpu(c, precision))
    assert _relerr(r2.cpu().numpy(), orc.spatial_rec(c, wn, l2, dilation)) <= 4 * tol


@pytest.mark.parametrize("precision", ["double", "single"])
@pytest.mark.parametrize("l2", [0, 1])
def test_haar4d_closed_form_of_the_reference(precision, l2):
    """the db1 known answer the reference ships as signal-domain code (Functions/harr_nddwt_4D.m:262-281,555-579,
    multi-level bookkeeping :173,:228), against the HIP path: 4-D, 1..3 levels, both pres_l2_norm"""
    rng = np.random.default_rng(77)
    sizes = [20, 12, 10, 9]
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_4D("db1", sizes, "pres_l2_norm", l2, "precision", precision)
    for level in (1, 2, 3):
        want = orc.haar4d_dec(x, level, l2)
        y = w.dec(_colmajor_gpu(x, precision), level)
        assert _relerr(y.cpu().numpy(), want) <= TOL[precision]
        c = rng.standard_normal(want.shape)
        assert _relerr(w.rec(_colmajor_gpu(c, precision)).cpu().numpy(), orc.haar4d_rec(c, l2)) <= 4 * TOL[precision]


def test_fused_and_per_axis_kernels_agree_with_uneven_chunks():
    """same plan, fused vs per-axis path, several workgroup chunkings of the marched axis"""
    sizes = [100, 52, 37]
    x = torch.randn(*reversed(sizes), device="cuda", dtype=torch.float64).permute(2, 1, 0)
    w = ndwt.nd_dwt_3D(["db4", "db2", "db3"], sizes, "pres_l2_norm", 1)
    ref = None
    for generic, zc in ((True, 0), (False, 0), (False, 5), (False, 37), (False, 16)):
        w.dec(x, 1)
        p = list(w._plans.values())[0]
        p.set_path(generic)
        p.set_tuning(0, zc)
        y = w.dec(x, 3)
        r = w.rec(y)
        if ref is None:
            ref = y
        assert float((y - ref).abs().max()) < 1e-12
        assert float((r - x).abs().max()) < 1e-12


def test_host_offload_compute_and_numpy_io():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((33, 20, 18))
    w = ndwt.nd_dwt_3D("db2", [33, 20, 18], "compute", "hip_off")
    y = w.dec(x, 2)
    assert isinstance(y, np.ndarray) and y.shape == (33, 20, 18, 15)
    assert _relerr(y, orc.spatial_dec(x, "db2", 2, 0)) < 1e-12
    assert _relerr(w.rec(y), x) < 1e-12
    # row vectors are transposed like nd_dwt_1D.m:151-153
    w1 = ndwt.nd_dwt_1D("db3", 50, "compute", "gpu_off")
    xr = rng.standard_normal((1, 50))
    assert _relerr(w1.dec(xr, 2), orc.spatial_dec(xr.reshape(-1), "db3", 2, 0)) < 1e-12


def test_host_pointer_entry_points_of_the_c_abi():
    import ctypes
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    p = api.Plan([24, 10, 12], ["db2", "db1", "db3"], torch.float64, max_level=2)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((24, 10, 12))
    xk = np.ascontiguousarray(x.T)
    yk = np.empty((15, 12, 10, 24))
    ndwt._lib.check(ndwt.lib().ndwt_dec_host(p._h, xk.ctypes.data_as(ctypes.c_void_p), yk.ctypes.data_as(ctypes.c_void_p), 2))
    assert _relerr(yk.T, orc.spatial_dec(x, ["db2", "db1", "db3"], 2, 0)) < 1e-12
    xr = np.empty_like(xk)
    ndwt._lib.check(ndwt.lib().ndwt_rec_host(p._h, yk.ctypes.data_as(ctypes.c_void_p), xr.ctypes.data_as(ctypes.c_void_p), 2))
    assert _relerr(xr, xk) < 1e-12
    with pytest.raises(ndwt.NdwtError):
        p.dec(0, 0, 1)                                     # null pointers are an error, not a crash
    with pytest.raises(ndwt.NdwtError, match="max_level"):
        p.dec(1, 1, 5)


@pytest.mark.parametrize("d,sizes,wn,level,precision,cplx", [
    (3, [72, 40, 33], "db4", 3, "single", False),
    (2, [130, 64], ["db2", "db3"], 2, "double", False),
    (3, [24, 20, 16], "db2", 2, "single", True),
    (1, [4096], "db2", 3, "double", False),
])
def test_device_resident_coefficient_handles(d, sizes, wn, level, precision, cplx):
    """ndwt_coef_*: what the gateway's dec_keep / rec_handle / shrink / fetch / release commands call.  dec with the
    coefficients left on the device = the host-pointer dec; rec of the handle inverts it; shrink on the handle = shrink of the array;
    put / get round-trip the reference's packed layout; the plan's staging is reused across calls"""
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    rng = np.random.default_rng(23)
    wl = [wn] * d if isinstance(wn, str) else wn
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    rdt = np.float32 if precision == "single" else np.float64
    cdt = (np.complex64 if precision == "single" else np.complex128) if cplx else rdt
    xk = np.ascontiguousarray(np.transpose(x)).astype(cdt)
    plan = api.Plan(sizes, wl, torch.float32 if precision == "single" else torch.float64, cplx, True, "reference", max_level=level)
    nb = api.num_bands(d, level)
    want = np.ascontiguousarray(np.transpose(orc.spatial_dec(x, wl, level, 1)))
    c = api.Coefficients.dec(plan, xk, level)
    info = c.info()
    assert info["level"] == level and info["bands"] == nb and info["band_pitch"] >= xk.size and info["dev_ptr"]
    y = c.get(np.empty((nb,) + xk.shape, dtype=cdt))
    assert _relerr(y, want) <= TOL[precision]
    r = c.rec(np.empty_like(xk))
    assert _relerr(r, xk) <= 20 * TOL[precision]
    # the host-pointer dec gives the same array (and reuses the plan's staging: two calls, same buffers)
    y2 = np.empty_like(y)
    for _ in range(2):
        api.L.check(api.L.lib().ndwt_dec_host(plan._h, xk.ctypes.data_as(ctypes.c_void_p), y2.ctypes.data_as(ctypes.c_void_p), level))
    assert np.array_equal(y, y2)
    # shrink on the handle, then rec = the denoising step of the host form
    thr = 0.4
    c.shrink(thr)
    den = c.rec(np.empty_like(xk))
    want_den = np.empty_like(xk)
    plan.denoise_host(xk.ctypes.data_as(ctypes.c_void_p), want_den.ctypes.data_as(ctypes.c_void_p), level, thr)
    assert _relerr(den, want_den) <= 20 * TOL[precision]
    # refill the same handle from another signal; upload arbitrary coefficients
    c2 = api.Coefficients.dec(plan, 2 * xk, level, reuse=c)
    assert c2 is c and _relerr(c.get(np.empty_like(y)), 2 * want) <= TOL[precision]
    cc = rng.standard_normal(y.shape).astype(rdt).astype(cdt)
    u = api.Coefficients.put(plan, cc, level)
    assert np.array_equal(u.get(np.empty_like(cc)), cc)
    want_rec = np.ascontiguousarray(np.transpose(orc.spatial_rec(np.transpose(cc), wl, 1)))
    assert np.abs(u.rec(np.empty_like(xk)) - want_rec).max() <= 2 * TOL[precision] * max(np.abs(want_rec).max(), np.abs(cc).max())
    other = api.Plan(sizes, wl, torch.float32 if precision == "single" else torch.float64, cplx, True, "reference", max_level=level)
    with pytest.raises(ndwt.NdwtError, match="another plan"):
        api.L.check(api.L.lib().ndwt_coef_rec_host(other._h, u._h, xk.ctypes.data_as(ctypes.c_void_p)))
    assert api.L.lib().ndwt_plan_destroy(plan._h) != 0 and "still alive" in api.L.lib().ndwt_last_error().decode()   # handles first
    u.release()
    c.release()
    plan.release_staging()
    api.L.check(api.L.lib().ndwt_dec_host(plan._h, xk.ctypes.data_as(ctypes.c_void_p), y2.ctypes.data_as(ctypes.c_void_p), level))
    assert np.array_equal(y, y2)


def test_split_complex_entry_points_match_the_interleaved_transform():
    """separate re / im arrays (the mxGetPr / mxGetPi layout of nd_dwt_mex.c:55-58) through ndwt_*_split[_host]"""
    import ctypes
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    dims, wn, lev = [20, 12, 9], ["db2", "db3", "db1"], 2
    rng = np.random.default_rng(16)
    x = rng.standard_normal(dims) + 1j * rng.standard_normal(dims)
    want = orc.spatial_dec(x, wn, lev, 0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    p = api.Plan(dims, wn, torch.float64, max_level=lev)
    xr, xi = np.ascontiguousarray(x.real.T), np.ascontiguousarray(x.imag.T)
    yr, yi = np.empty((15, 9, 12, 20)), np.empty((15, 9, 12, 20))
    p.dec_split_host(vp(xr), vp(xi), vp(yr), vp(yi), lev)
    assert _relerr(yr.T + 1j * yi.T, want) < 1e-12
    br, bi = np.empty_like(xr), np.empty_like(xi)
    p.rec_split_host(vp(yr), vp(yi), vp(br), vp(bi), lev)
    assert _relerr(br + 1j * bi, xr + 1j * xi) < 1e-12
    # device pointers; real data = NULL imaginary parts
    tx, ty = torch.from_numpy(xr).cuda(), torch.empty((15, 9, 12, 20), dtype=torch.float64, device="cuda")
    p.dec_split(tx.data_ptr(), None, ty.data_ptr(), None, lev)
    torch.cuda.synchronize()
    assert _relerr(ty.cpu().numpy(), yr) < 1e-14
    with pytest.raises(ndwt.NdwtError, match="both"):
        p.dec_split(tx.data_ptr(), tx.data_ptr(), ty.data_ptr(), None, lev)
    pc = api.Plan(dims, wn, torch.float64, True, max_level=lev)
    with pytest.raises(ndwt.NdwtError, match="NDWT_REAL"):
        pc.dec_split(tx.data_ptr(), None, ty.data_ptr(), None, lev)


@pytest.mark.parametrize("sizes,wn,precision", [
    ([24, 20, 16], ["db4", "db2", "db3"], "single"),          # every axis divides by 4: levels 2 and 3 run fused on sub-lattices
    ([72, 40, 32], ["db2", "db2", "db4"], "single"),
    ([24, 20, 16], ["db3", "db1", "db2"], "double"),          # double: level 2 fused (stride 2), level 3 per axis
    ([24, 18, 16], ["db2", "db2", "db2"], "single"),          # 18 % 4 != 0: level 3 per axis
])
def test_atrous_levels_on_sublattices(sizes, wn, precision):
    """dilated (a-trous) levels: stride-2 / stride-4 levels run as independent stride-1 problems on the sub-lattices
    (x taps stepping over 2 / 4 interleaved scalars, (y, z) sub-lattices as batch items with strided rows and planes)"""
    rng = np.random.default_rng(41)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", precision, "dilation", "atrous")
    tol = TOL[precision]
    xg = _colmajor_gpu(x, precision)
    for level in (2, 3):
        y = w.dec(xg, level)
        want = orc.spatial_dec(x, wn, level, 1, "atrous")
        assert _relerr(y.cpu().numpy(), want) <= tol
        assert _relerr(w.rec(y).cpu().numpy(), x) <= 20 * tol
        c = rng.standard_normal(want.shape)
        got = w.rec(_colmajor_gpu(c, precision)).cpu().numpy()
        assert _relerr(got, orc.spatial_rec(c, wn, 1, "atrous")) <= 4 * tol
    # same numbers as the per-axis kernels
    wg = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", precision, "dilation", "atrous")
    plan = wg._plan(False, 3, xg.device)
    plan.set_path(True)
    assert _relerr(wg.dec(xg, 3).cpu().numpy(), w.dec(xg, 3).cpu().numpy()) <= tol


def test_atrous_level3_rec_on_buffers_that_are_not_16_byte_aligned():
    """float a-trous synthesis at tap stride 4 has a whole-lane-shift kernel for 16-byte-aligned data only; a coefficient or output
    pointer off by one element must fall back to Inv3S<.., EW = 4> and give the same numbers (ndwt_rec states no alignment contract)"""
    rng = np.random.default_rng(43)
    sizes, wn, level = [24, 20, 16], ["db4", "db2", "db3"], 3
    c = rng.standard_normal(sizes + [ndwt.num_bands(3, level)])
    want = orc.spatial_rec(c, wn, 1, "atrous")
    ck = torch.from_numpy(np.ascontiguousarray(np.transpose(c))).float()            # kernel order (bands, n3, n2, n1)
    plan = ndwt.Plan(sizes, wn, torch.float32, False, True, "atrous", max_level=level)
    vol = int(np.prod(sizes))
    for off_in, off_out in ((0, 0), (1, 0), (0, 1), (1, 1)):
        ybuf = torch.zeros(ck.numel() + 4, dtype=torch.float32, device="cuda")
        ybuf[off_in:off_in + ck.numel()] = ck.reshape(-1).cuda()
        xbuf = torch.zeros(vol + 4, dtype=torch.float32, device="cuda")
        plan.rec(ybuf.data_ptr() + 4 * off_in, xbuf.data_ptr() + 4 * off_out, level, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = np.transpose(xbuf[off_out:off_out + vol].reshape(sizes[::-1]).cpu().numpy())
        assert _relerr(got, want) <= 4 * TOL["single"], (off_in, off_out)
        assert float(xbuf[:off_out].abs().sum()) == 0 and float(xbuf[off_out + vol:].abs().sum()) == 0   # nothing outside the output


@pytest.mark.parametrize("sizes,wn,precision", [
    ([72, 40], ["db4", "db2"], "single"),        # both axes divide by 4: levels 2 and 3 fused on sub-lattices
    ([264, 36], ["db3", "db4"], "single"),       # two wave tiles along x
    ([72, 40], ["db2", "db3"], "double"),        # double: stride 2 fused, stride 4 per axis
    ([72, 42], ["db2", "db2"], "single"),        # 42 % 4 != 0: level 3 per axis
])
def test_atrous_levels_on_sublattices_2d(sizes, wn, precision):
    rng = np.random.default_rng(43)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_2D(wn, sizes, "pres_l2_norm", 0, "precision", precision, "dilation", "atrous")
    tol = TOL[precision]
    xg = _colmajor_gpu(x, precision)
    for level in (2, 3):
        y = w.dec(xg, level)
        want = orc.spatial_dec(x, wn, level, 0, "atrous")
        assert _relerr(y.cpu().numpy(), want) <= tol
        assert _relerr(w.rec(y).cpu().numpy(), x) <= 20 * tol
        c = rng.standard_normal(want.shape)
        got = w.rec(_colmajor_gpu(c, precision)).cpu().numpy()
        assert _relerr(got, orc.spatial_rec(c, wn, 0, "atrous")) <= 4 * tol


def _np_shrink(c, t, hard):
    m = np.abs(c)
    with np.errstate(invalid="ignore", divide="ignore"):
        g = np.where(m > t, 1.0 if hard else (m - t) / np.where(m > 0, m, 1.0), 0.0)
    out = c * g
    out[..., 0] = c[..., 0]                                   # the coarsest approximation band is kept
    return out


@pytest.mark.parametrize("d,sizes,wn,cplx,precision", [
    (3, [24, 18, 12], ["db2", "db4", "db1"], False, "double"),
    (3, [20, 16, 9], ["db3", "db2", "db2"], True, "single"),
    (2, [37, 21], ["db4", "db1"], True, "double"),           # odd sizes: scalar path of the shrink kernel
    (1, [130], "db3", False, "single"),
])
def test_shrink_and_denoise_against_numpy(d, sizes, wn, cplx, precision):
    """soft / hard thresholding of the detail bands and the one-call dec -> shrink -> rec (C-ABI extension for solvers)"""
    rng = np.random.default_rng(31)
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    w = _cls(d)(wn, sizes, "pres_l2_norm", 1, "precision", precision)
    tol = TOL[precision]
    wl = [wn] * d if isinstance(wn, str) else wn
    y_ref = orc.spatial_dec(x, wl, 2, 1)
    xg = _colmajor_gpu(x, precision)
    y = w.dec(xg, 2)
    for mode in ("soft", "hard"):
        want = _np_shrink(y_ref, 0.4, mode == "hard")
        got = w.shrink(y, 0.4, mode).cpu().numpy()
        # elements within rounding of the threshold may fall on either side in single precision
        near = np.abs(np.abs(y_ref) - 0.4) < 1e-4
        assert np.abs(np.where(near, 0, got - want)).max() <= 10 * tol * np.abs(y_ref).max()
        den = w.denoise(xg, 2, 0.4, mode).cpu().numpy()
        want_x = orc.spatial_rec(np.where(near, got, want), wl, 1)
        assert np.abs(den - want_x).max() <= 20 * tol * max(np.abs(want_x).max(), 1.0)
    assert torch.equal(y, w.dec(xg, 2))                       # shrink() returned a copy
    assert _relerr(w.denoise(xg, 2, 0.0).cpu().numpy(), x) < 20 * tol      # threshold 0 = identity
    with pytest.raises(ndwt.NdwtError, match="threshold"):
        w.denoise(xg, 2, -1.0)


@pytest.mark.parametrize("d,sizes,wn,cplx,precision,pitch", [
    (3, [64, 32, 16], "db4", False, "single", "auto"),            # fused 3-D float (pair-packed synthesis)
    (3, [24, 18, 12], ["db2", "db4", "db1"], False, "double", "auto"),
    (3, [20, 16, 9], ["db3", "db2", "db2"], True, "single", 20 * 16 * 9 + 3),   # odd pitch: bands off the 16-byte grid, scalar kernels
    (2, [64, 48], "db4", True, "double", "auto"),
    (2, [37, 21], ["db4", "db1"], False, "single", 37 * 21 + 5),
    (1, [130], "db3", False, "double", "auto"),
    (4, [16, 12, 10, 8], "db2", False, "single", "auto"),          # 16 bands; the t-axis temporaries are skewed too
    (3, [32, 32, 32], ["db9", "db8", "db8"], False, "single", "auto"),   # fused analysis, per-axis synthesis (odd tap padding)
])
def test_pitched_coefficient_layout_gives_the_packed_results(d, sizes, wn, cplx, precision, pitch):
    """ndwt_dec_pitched / ndwt_rec_pitched / ndwt_shrink_pitched (include/ndwt.h): band b at b * band_pitch elements.  The
    values are those of the packed (reference) layout bit for bit; the classes hand out and take pitched tensors transparently."""
    rng = np.random.default_rng(77)
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    xg = _colmajor_gpu(x, precision)
    wp = _cls(d)(wn, sizes, "pres_l2_norm", 1, "precision", precision)
    wq = _cls(d)(wn, sizes, "pres_l2_norm", 1, "precision", precision, "band_pitch", pitch)
    level = 2
    yp, yq = wp.dec(xg, level), wq.dec(xg, level)
    vol = int(np.prod(sizes))
    assert yq.shape == yp.shape and yq.stride()[-1] > vol and yp.stride()[-1] == vol
    assert torch.equal(yq, yp)                                         # same kernels, same values
    wl = [wn] * d if isinstance(wn, str) else wn
    assert _relerr(yq.cpu().numpy(), orc.spatial_dec(x, wl, level, 1)) <= TOL[precision]
    xr = wq.rec(yq)                                                    # pitched tensor used where it lies
    assert torch.equal(xr, wp.rec(yp)) and torch.equal(xr, wp.rec(yq)) and torch.equal(xr, wq.rec(yq.contiguous()))
    assert _relerr(xr.cpu().numpy(), x) < 20 * TOL[precision]
    for mode in ("soft", "hard"):
        sq = wq.shrink(yq, 0.4, mode)
        assert sq.stride() == yq.stride() and sq.data_ptr() != yq.data_ptr()
        assert torch.equal(sq, wp.shrink(yp, 0.4, mode))
    assert torch.equal(yq, yp)                                         # shrink() returned a copy
    den, want = wq.denoise(xg, level, 0.4), wp.rec(wp.shrink(yp, 0.4))   # (denoise may shrink inside the synthesis kernel: rounding)
    assert float((den - want).abs().max()) <= 20 * TOL[precision] * max(float(want.abs().max()), 1.0)
    with pytest.raises(ValueError, match="smaller than a band"):
        _cls(d)(wn, sizes, "precision", precision, "band_pitch", vol - 1).dec(xg, 1)
    plan = list(wp._plans.values())[0]
    with pytest.raises(ndwt.NdwtError, match="smaller than a band"):
        plan.rec(yp.data_ptr(), xr.data_ptr(), level, 0, band_pitch=vol - 1)


@pytest.mark.parametrize("sizes,wn,path", [
    ([64, 40, 36], "db7", "fused3d"),
    ([68, 41, 30], "db8", "fused3d"),
    ([64, 48, 32], "db9", "fused3d"),                                  # 18 taps: pair-packed synthesis on its 64 x 24 tile
    ([70, 37, 33], "db9", "fused3d"),                                  # ... scalar accesses (n1 not a multiple of 4), ragged tiles
    ([64, 40, 36], ["db9", "db7", "db5"], "fused3d"),                  # mixed wavelets, even padding to 18 taps on every axis
    ([64, 40, 36], ["db9", "db8", "db8"], "fused3d analysis, axis synthesis"),   # odd padding: no derived high-pass taps
    ([64, 40, 36], ["db8", "db7", "db7"], "fused3d"),                  # odd padding, 16 taps: the lane-shift kernel
    ([64, 40, 36], "db10", "fused3d"),                                 # 20 taps: 48 x 28 synthesis tile
    ([100, 37, 33], "db10", "fused3d"),                                # ... ragged 48-wide tiles, rows and planes
    ([62, 40, 36], ["db10", "db8", "db6"], "fused3d"),
    ([64, 40, 36], ["db10", "db9", "db9"], "fused3d analysis, axis synthesis"),
])
def test_long_filters_float(sizes, wn, path):
    """db7 .. db10 on real float data: which kernels serve them, and parity with the oracle for dec, rec and the round trip"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", "single")
    xg = _colmajor_gpu(x, "single")
    y = w.dec(xg, 2)
    assert list(w._plans.values())[0].describe() == path
    wl = [wn] * 3 if isinstance(wn, str) else wn
    assert _relerr(y.cpu().numpy(), orc.spatial_dec(x, wl, 2, 1)) <= TOL["single"]
    c = rng.standard_normal(sizes + [15])
    got = w.rec(_colmajor_gpu(c, "single")).cpu().numpy()
    want = orc.spatial_rec(c, wl, 1)
    assert np.abs(got - want).max() <= TOL["single"] * max(np.abs(want).max(), np.abs(c).max())
    assert _relerr(w.rec(y).cpu().numpy(), x) < 1e-5


@pytest.mark.parametrize("sizes,wn", [
    ([64, 40, 36], "db4"),                                             # 8 taps: the gather form is the default, the scatter form on request (10)
    ([128, 64, 40], "db5"), ([64, 40, 36], "db6"), ([72, 36, 30], ["db6", "db6", "db4"]),   # 10 / 12 taps, with and without the shared y / z tap pairs
    ([64, 40, 36], "db8"), ([64, 48, 32], "db9"), ([96, 56, 30], "db10"),                   # sums that travel 2 and 3 lanes; the 64 x 24 and 48 x 28 tiles
])
def test_scatter_and_gather_form_of_the_synthesis_x_stage(sizes, wn):
    """Inv3Y's x stage in scatter form (partial sums travel between lanes; default for 10 .. 20 taps on rows of whole groups of 4) and in
    gather form (variant_inv 11; variant_inv 10 = scatter form for 8 taps too): both against the oracle, and against each other to rounding"""
    rng = np.random.default_rng(41)
    wl = [wn] * 3 if isinstance(wn, str) else wn
    c = rng.standard_normal(sizes + [15])
    cg = _colmajor_gpu(c, "single")
    want = orc.spatial_rec(c, wl, 1)
    scale = max(np.abs(want).max(), np.abs(c).max())
    got = {}
    for variant in (10, 11):
        w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", "single")
        w._plan(False, 2, cg.device).set_variant(inv=variant)
        got[variant] = w.rec(cg).cpu().numpy()
        assert list(w._plans.values())[0].describe() == "fused3d"
        assert np.abs(got[variant] - want).max() <= TOL["single"] * scale, variant
    assert np.abs(got[10] - got[11]).max() <= 1e-6 * scale
    assert np.abs(got[10] - got[11]).max() > 0                         # two different kernels did run


@pytest.mark.parametrize("sizes,wn,path", [
    ([64, 40, 36], "db7", "fused3d"),                                  # 14 taps: 64 x 8 tiles, 512 threads
    ([68, 41, 30], "db8", "fused3d"),                                  # 16 taps, ragged tiles
    ([70, 37, 33], "db8", "fused3d"),                                  # ... rows that are not whole groups of 4 elements
    ([64, 40, 36], ["db8", "db3", "db7"], "fused3d"),                  # mixed wavelets padded to 16 taps (odd and even padding)
    ([64, 40, 36], "db9", "axis"),                                     # 18 / 20 taps: per-axis path (the fused form spills 400+ registers)
])
def test_long_filters_double_14_16_taps(sizes, wn, path):
    """db7 / db8 on real double data (the reference mex path's precision, Test/nddwt3D_test.m:11 wavelets): fused, parity with the oracle"""
    rng = np.random.default_rng(7)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", "double")
    xg = _colmajor_gpu(x, "double")
    y = w.dec(xg, 2)
    assert list(w._plans.values())[0].describe() == path
    wl = [wn] * 3 if isinstance(wn, str) else wn
    assert _relerr(y.cpu().numpy(), orc.spatial_dec(x, wl, 2, 1)) <= TOL["double"]
    c = rng.standard_normal(sizes + [15])
    got = w.rec(_colmajor_gpu(c, "double")).cpu().numpy()
    want = orc.spatial_rec(c, wl, 1)
    assert np.abs(got - want).max() <= TOL["double"] * max(np.abs(want).max(), np.abs(c).max())
    assert _relerr(w.rec(y).cpu().numpy(), x) < 1e-13


@pytest.mark.parametrize("sizes,wn,precision,path", [
    ([32, 24, 20], "db5", "single", "fused3d"),
    ([34, 21, 19], "db6", "single", "fused3d"),
    ([32, 24, 20], ["db6", "db2", "db5"], "single", "fused3d"),
    ([32, 24, 20], "db7", "single", "fused3d"),                        # complex64 with 14 / 16 taps: 64x16 analysis tile, 48-wide pair-packed synthesis
    ([50, 37, 20], "db8", "single", "fused3d"),
    ([33, 24, 20], "db8", "single", "fused3d"),                        # ... odd number of complex elements per row
    ([32, 24, 20], ["db8", "db7", "db8"], "single", "fused3d analysis, axis synthesis"),   # odd padding: no derived high-pass taps
    ([32, 24, 20], "db9", "single", "axis"),
    ([32, 24, 20], "db5", "double", "fused3d"),                        # complex128: fused up to 10 taps,
    ([34, 21, 19], "db5", "double", "fused3d"),
    ([32, 24, 20], "db6", "double", "fused3d analysis, axis synthesis"),   # ... 12 taps in the analysis only
    ([32, 24, 20], "db7", "double", "axis"),
])
def test_long_filters_complex(sizes, wn, precision, path):
    """interleaved complex data with 10 / 12 taps: fused in single precision (x taps step over (re, im) pairs)"""
    rng = np.random.default_rng(6)
    x = rng.standard_normal(sizes) + 1j * rng.standard_normal(sizes)
    w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", precision)
    xg = _colmajor_gpu(x, precision)
    y = w.dec(xg, 2)
    assert list(w._plans.values())[0].describe() == path
    wl = [wn] * 3 if isinstance(wn, str) else wn
    assert _relerr(y.cpu().numpy(), orc.spatial_dec(x, wl, 2, 1)) <= TOL[precision]
    c = rng.standard_normal(sizes + [15]) + 1j * rng.standard_normal(sizes + [15])
    got = w.rec(_colmajor_gpu(c, precision)).cpu().numpy()
    want = orc.spatial_rec(c, wl, 1)
    assert np.abs(got - want).max() <= TOL[precision] * max(np.abs(want).max(), np.abs(c).max())
    assert _relerr(w.rec(y).cpu().numpy(), x) < 20 * TOL[precision]


@pytest.mark.parametrize("sizes,wn,cplx", [
    ([512, 130, 12], "db4", False),                                    # 8 x 5 tall tiles, ragged in y, one short z chunk
    ([260, 250, 9], ["db3", "db4", "db2"], False),                     # ragged in x and y
    ([256, 130, 10], "db4", True),                                     # interleaved complex: 512 scalars per row
    ([132, 250, 9], ["db3", "db2", "db4"], True),
])
def test_tall_analysis_tile_against_oracle(sizes, wn, cplx):
    """float analysis with 6 / 8 taps takes the 64 x 32 tile with 1024 threads once the volume has 32 such tiles (real and
    interleaved complex data); smaller volumes keep the 64 x 16 tile -- both against the oracle"""
    rng = np.random.default_rng(9)
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", "single")
    xg = _colmajor_gpu(x, "single")
    y = w.dec(xg, 2)
    wl = [wn] * 3 if isinstance(wn, str) else wn
    assert _relerr(y.cpu().numpy(), orc.spatial_dec(x, wl, 2, 1)) <= TOL["single"]
    assert _relerr(w.rec(y).cpu().numpy(), x) < 1e-5


@pytest.mark.parametrize("sizes,wn", [
    ([64, 40, 36], "db5"),
    ([68, 41, 30], "db6"),                                             # 64 x 8 tiles with 512 threads in both directions
    ([70, 37, 33], ["db6", "db5", "db3"]),                             # scalar accesses, mixed wavelets
])
def test_long_filters_double_10_12_taps(sizes, wn):
    """fp64 with 10 / 12 taps on the fused kernels (spill-free 64 x 8 tiles)"""
    rng = np.random.default_rng(7)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", 1, "precision", "double")
    xg = _colmajor_gpu(x, "double")
    y = w.dec(xg, 2)
    assert list(w._plans.values())[0].describe() == "fused3d"
    wl = [wn] * 3 if isinstance(wn, str) else wn
    assert _relerr(y.cpu().numpy(), orc.spatial_dec(x, wl, 2, 1)) <= TOL["double"]
    c = rng.standard_normal(sizes + [15])
    got = w.rec(_colmajor_gpu(c, "double")).cpu().numpy()
    want = orc.spatial_rec(c, wl, 1)
    assert np.abs(got - want).max() <= TOL["double"] * max(np.abs(want).max(), np.abs(c).max())
    assert _relerr(w.rec(y).cpu().numpy(), x) < 20 * TOL["double"]


def test_properties_linearity_shift_adjoint():
    torch.manual_seed(0)
    sizes = [48, 36, 40]
    w = ndwt.nd_dwt_3D("db4", sizes, "pres_l2_norm", 1)
    mk = lambda: torch.randn(40, 36, 48, device="cuda", dtype=torch.float64).permute(2, 1, 0)
    a, b = mk(), mk()
    ya, yb = w.dec(a, 2), w.dec(b, 2)
    assert float((w.dec(2.5 * a - b, 2) - (2.5 * ya - yb)).abs().max()) < 1e-12              # linearity
    sh = (5, -3, 7)
    ys = w.dec(torch.roll(a, sh, dims=(0, 1, 2)), 2)
    assert float((ys - torch.roll(ya, sh, dims=(0, 1, 2))).abs().max()) < 1e-12              # periodic shift equivariance
    c = torch.randn(15, 40, 36, 48, device="cuda", dtype=torch.float64).permute(3, 2, 1, 0)
    lhs = float((ya * c).sum())
    rhs = float((a * w.rec(c)).sum())
    assert abs(lhs - rhs) < 1e-9 * max(abs(lhs), 1.0)                                          # <dec x, c> = <x, rec c> in l2 mode


@pytest.mark.parametrize("wname,level,sizes", [("db4", 3, [512, 512, 512]), ("db6", 4, [512, 512, 128])],
                         ids=["cfg3_512cube_db4_L3", "db6_L4_512x512x128"])
def test_full_size_fp32_roundtrip_and_energy(wname, level, sizes):
    """BASELINE config 3 at full size: properties only (the oracle cannot run this in seconds)."""
    torch.manual_seed(1)
    n1, n2, n3 = sizes
    x = torch.randn(n3, n2, n1, device="cuda", dtype=torch.float32).permute(2, 1, 0)
    w = ndwt.nd_dwt_3D(wname, sizes, "pres_l2_norm", 1, "precision", "single")
    y = w.dec(x, level)
    assert y.shape[-1] == 8 + 7 * (level - 1)
    nx = float(torch.linalg.vector_norm(x.double()))
    ny = float(torch.sqrt(sum(torch.linalg.vector_norm(y[..., b].double()) ** 2 for b in range(y.shape[-1]))))
    assert abs(ny - nx) < 2e-6 * nx                                                           # energy (B)
    r = w.rec(y)
    del y
    rel = float(torch.linalg.vector_norm((r - x).double())) / nx
    assert rel < 1e-6, rel                                                                     # round trip (A), relative l2
    # spot-check one z-column of the finest LLH.. bands against the oracle restricted to a sub-volume is not
    # possible (periodic), so compare against the per-axis kernels on a smaller crop with the same tiles
    xs = x[:128, :64, :72].contiguous().permute(2, 1, 0).contiguous().permute(2, 1, 0)
    ws = ndwt.nd_dwt_3D(wname, [128, 64, 72], "pres_l2_norm", 1, "precision", "single")
    yf = ws.dec(xs, 2)
    for p in ws._plans.values():
        p.set_path(True)
    yg = ws.dec(xs, 2)
    assert float((yf - yg).abs().max()) < 2e-6 * float(yg.abs().max())


def test_slab_entry_points_reproduce_the_periodic_transform():
    """two z-slabs with explicit halos == the periodic single-device level (fused and per-axis paths)"""
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    n1, n2, n3 = 40, 24, 32
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-6)):
        for generic in (False, True):
            x = torch.randn(n3, n2, n1, device="cuda", dtype=dtype)
            full = api.Plan([n1, n2, n3], ["db3"] * 3, dtype, max_level=1)
            full.set_path(generic)
            y = torch.empty(8, n3, n2, n1, device="cuda", dtype=dtype)
            full.dec(x.data_ptr(), y.data_ptr(), 1)
            half = api.Plan([n1, n2, n3 // 2], ["db3"] * 3, dtype, max_level=1)
            half.set_path(generic)
            ab, aa, sb, sa = half.slab_halo(1)
            assert (ab, aa, sb, sa) == (2, 3, 3, 2)
            r = torch.empty(n3, n2, n1, device="cuda", dtype=dtype)
            for k in range(2):
                z0 = k * n3 // 2
                idx = torch.arange(z0 - ab, z0 + n3 // 2 + aa, device="cuda") % n3
                xin = x[idx].contiguous()
                outs = torch.empty(8, n3 // 2, n2, n1, device="cuda", dtype=dtype)
                half.analysis_level_slab(xin.data_ptr(), [outs[b].data_ptr() for b in range(8)], 1)
                assert float((outs - y[:, z0:z0 + n3 // 2]).abs().max()) <= tol * float(y.abs().max())
                idx = torch.arange(z0 - sb, z0 + n3 // 2 + sa, device="cuda") % n3
                yin = y[:, idx].contiguous()
                rk = torch.empty(n3 // 2, n2, n1, device="cuda", dtype=dtype)
                half.synthesis_level_slab([yin[b].data_ptr() for b in range(8)], rk.data_ptr(), 1)
                r[z0:z0 + n3 // 2] = rk
            torch.cuda.synchronize()
            assert float((r - x).abs().max()) <= 10 * tol * float(x.abs().max())


def test_copy_free_slab_entry_points_and_sharded_driver_on_one_gpu():
    """split-halo analysis + zero-extended synthesis (the multi-GPU fast path) == the periodic transform"""
    import importlib
    api = importlib.import_module("non-decimated_wavelets_amd.api")
    sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
    n1, n2, n3 = 72, 40, 48
    # plans on the per-axis path (double with db9 here) refuse the copy-free entry points instead of doing something else
    p64 = api.Plan([n1, n2, 24], ["db9"] * 3, torch.float64, max_level=1)
    assert p64.describe() == "axis"
    dummy = torch.zeros(8, 24 + 17, n2, n1, device="cuda", dtype=torch.float64)
    with pytest.raises(ndwt.NdwtError):
        p64.synthesis_level_slab_ext([dummy[b].data_ptr() for b in range(8)], dummy.data_ptr(), 1)
    for dtype, tol in ((torch.float32, 2e-6), (torch.float64, 1e-12)):
        x = torch.randn(n3, n2, n1, device="cuda", dtype=dtype)
        full = api.Plan([n1, n2, n3], ["db4"] * 3, dtype, pres_l2_norm=True, max_level=1)
        y = torch.empty(8, n3, n2, n1, device="cuda", dtype=dtype)
        full.dec(x.data_ptr(), y.data_ptr(), 1)
        parts = [(0, 20), (20, 48)]                                  # uneven slabs
        r_acc = torch.zeros(n3, n2, n1, device="cuda", dtype=dtype)
        for z0, z1 in parts:
            nl = z1 - z0
            slab = api.Plan([n1, n2, nl], ["db4"] * 3, dtype, pres_l2_norm=True, max_level=1)
            ab, aa, sb, sa = slab.slab_halo(1)
            hb = x[torch.arange(z0 - ab, z0, device="cuda") % n3].contiguous()
            ha = x[torch.arange(z1, z1 + aa, device="cuda") % n3].contiguous()
            loc = x[z0:z1].contiguous()
            outs = torch.empty(8, nl, n2, n1, device="cuda", dtype=dtype)
            slab.analysis_level_slab_split(loc.data_ptr(), hb.data_ptr(), ha.data_ptr(), [outs[b].data_ptr() for b in range(8)], 1)
            assert float((outs - y[:, z0:z1]).abs().max()) <= tol * float(y.abs().max())
            ext = torch.empty(sa + nl + sb, n2, n1, device="cuda", dtype=dtype)
            yl = y[:, z0:z1].contiguous()
            slab.synthesis_level_slab_ext([yl[b].data_ptr() for b in range(8)], ext.data_ptr(), 1)
            idx = torch.arange(z0 - sa, z1 + sb, device="cuda") % n3
            r_acc.index_add_(0, idx, ext)                              # what the scatter-add exchange does
            # runs of planes of the same two calls (what the exchange overlaps with): bit-identical pieces
            o2 = torch.full_like(outs, float("nan"))
            for a0, a1, before, after in ((ab, nl - aa, loc[:ab], loc[nl - aa:]), (0, ab, hb, loc[ab:ab + aa]),
                                          (nl - aa, nl, loc[nl - aa - ab:nl - aa], ha)):
                slab.analysis_level_slab_part(loc[a0:a1].data_ptr(), before.data_ptr(), after.data_ptr(),
                                              [o2[b, a0:a1].data_ptr() for b in range(8)], a1 - a0, 1)
            assert torch.equal(o2, outs)
            e2 = torch.full_like(ext, float("nan"))
            for e0, e1 in ((0, sa), (sa + nl, sa + nl + sb), (sa, sa + nl)):
                slab.synthesis_level_slab_part([yl[b].data_ptr() for b in range(8)], nl, e0, e1 - e0, e2[e0:e1].data_ptr(), 1)
            assert float((e2 - ext).abs().max()) <= 1e-6 * float(ext.abs().max())
            # both ends in one launch: [halo | slab | halo] contiguous input, two runs of max(ab, aa) planes
            m = max(ab, aa)
            buf = torch.cat([hb, loc, ha], 0)
            o3 = torch.full_like(outs, float("nan"))
            slab.analysis_level_slab_runs(buf.data_ptr(), [o3[b].data_ptr() for b in range(8)], m, 2, nl - m, 1)
            assert torch.equal(o3[:, :m], outs[:, :m]) and torch.equal(o3[:, nl - m:], outs[:, nl - m:])
            assert bool(torch.isnan(o3[:, m:nl - m]).all())
            e3 = torch.full((2, sb) + tuple(ext.shape[1:]), float("nan"), device="cuda", dtype=dtype)
            slab.synthesis_level_slab_runs([yl[b].data_ptr() for b in range(8)], nl, 0, nl + sa, 2, sb, e3.data_ptr(), 1)
            assert float((e3[0, :sa] - ext[:sa]).abs().max()) <= 1e-6 * float(ext.abs().max())
            assert float((e3[1] - ext[sa + nl:]).abs().max()) <= 1e-6 * float(ext.abs().max())
            with pytest.raises(ndwt.NdwtError, match="zero-extended"):
                slab.synthesis_level_slab_part([yl[b].data_ptr() for b in range(8)], nl, nl, 8, e2.data_ptr(), 1)
        torch.cuda.synchronize()
        assert float((r_acc - x).abs().max()) <= 20 * tol * float(x.abs().max())
    # the driver itself with world size 1 (halo planes come from the own slab)
    xs = torch.randn(n3, n2, n1, device="cuda")
    w = ndwt.nd_dwt_3D("db4", [n1, n2, n3], "pres_l2_norm", 1, "precision", "single")
    yref = w.dec(xs.permute(2, 1, 0), 3).permute(3, 2, 1, 0)
    for overlap in (True, False):
        eng = sh.ShardedNdDwt(["db4"] * 3, [n1, n2, n3], pres_l2_norm=True, precision="single", device=torch.device("cuda", 0),
                              overlap=overlap)
        assert eng.scheme == "scatter" and eng.overlap == overlap
        ysh = eng.dec(xs, 3)
        assert float((ysh - yref).abs().max()) <= 2e-6 * float(yref.abs().max())
        assert float((eng.rec(ysh) - xs).abs().max()) <= 1e-5


def test_fused2d_kernels_at_baseline_config2_shape():
    """BASELINE config 2 (2-D fp32 4096x4096 db4, 3 levels): fused 2-D kernels vs the per-axis kernels + round trip"""
    torch.manual_seed(2)
    n1 = n2 = 4096
    x = torch.randn(n2, n1, device="cuda", dtype=torch.float32).permute(1, 0)
    w = ndwt.nd_dwt_2D("db4", [n1, n2], "pres_l2_norm", 1, "precision", "single")
    y = w.dec(x, 3)
    p = list(w._plans.values())[0]
    assert p.describe() == "fused2d"
    assert y.shape == (n1, n2, 10)
    r = w.rec(y)
    nx = float(torch.linalg.vector_norm(x.double()))
    assert float(torch.linalg.vector_norm((r - x).double())) / nx < 1e-6
    assert abs(float(torch.linalg.vector_norm(y.double())) - nx) < 2e-6 * nx
    p.set_path(True)
    yg = w.dec(x, 3)
    assert float((y - yg).abs().max()) <= 2e-6 * float(yg.abs().max())


@pytest.mark.parametrize("sizes,wn,level", [
    ([256, 96], "db4", 3),                       # two waves along x (224 columns each), three levels in one launch
    ([512, 130], ["db2", "db3"], 2),             # mixed wavelets padded to 6 taps, two levels
    ([64, 200], "db6", 4),                       # 12 taps: two launches of two levels
    ([636, 52], ["db3", "db6"], 5),              # ... two launches of two levels and a single one: the scratch volumes alternate per LAUNCH
    ([1024, 64], "db1", 5),                      # 2 taps, five levels: a launch of three and one of two
    ([260, 80], "db3", 4),                       # four levels: three in one launch, the fourth on the one-level kernel
    ([232, 77], "db4", 3),
])
def test_cascaded_2d_analysis_against_oracle(sizes, wn, level):
    """dec of a float image with the levels cascaded inside one march (Fwd2C; the default beyond 2048^2, forced here through
    variant_fwd 11): the oracle's coefficients, bit-identical to one launch per level, and rec() inverts it"""
    rng = np.random.default_rng(17)
    x = rng.standard_normal(sizes)
    wl = [wn] * 2 if isinstance(wn, str) else wn
    want = orc.spatial_dec(x, wl, level, 1)
    xg = _colmajor_gpu(x, "single")
    res = {}
    for variant in (11, 9):                      # cascaded / one launch per level
        w = ndwt.nd_dwt_2D(wn, sizes, "pres_l2_norm", 1, "precision", "single")
        w._plan(False, level, xg.device).set_variant(fwd=variant)
        res[variant] = w.dec(xg, level)
        assert _relerr(res[variant].cpu().numpy(), want) <= TOL["single"], variant
    assert float((res[11] - res[9]).abs().max()) == 0.0
    assert _relerr(w.rec(res[11]).cpu().numpy(), x) < 1e-5


@pytest.mark.parametrize("sizes,wn,level,depth", [
    ([256, 96], "db4", 3, 1),
    ([512, 130], ["db2", "db3"], 2, 2),          # two rows of band loads in flight per level
    ([1024, 64], "db1", 5, 1),                   # five levels: a launch of three and one of two
    ([260, 80], "db3", 4, 1),                    # four levels: three in one launch, the finest on the one-level kernel
    ([232, 77], "db4", 3, 2),
])
def test_cascaded_2d_synthesis_against_oracle(sizes, wn, level, depth):
    """rec of arbitrary float coefficients with the levels cascaded inside one march (Inv2C; the default beyond 2048^2, forced here through
    variant_inv 11 / 12): the oracle's reconstruction, and the same bits as one launch per level"""
    rng = np.random.default_rng(18)
    wl = [wn] * 2 if isinstance(wn, str) else wn
    c = rng.standard_normal(sizes + [ndwt.num_bands(2, level)])
    want = orc.spatial_rec(c, wl, 1)
    cg = _colmajor_gpu(c, "single")
    res = {}
    for variant in (11 if depth == 1 else 12, 9):
        w = ndwt.nd_dwt_2D(wn, sizes, "pres_l2_norm", 1, "precision", "single")
        w._plan(False, level, cg.device).set_variant(inv=variant)
        res[variant] = w.rec(cg)
        assert np.abs(res[variant].cpu().numpy() - want).max() <= 2 * TOL["single"] * max(np.abs(want).max(), np.abs(c).max()), variant
    # ndwt_denoise on the cascaded kernels (thresholding fused into Inv2C's loads) against one launch per level
    xg = _colmajor_gpu(rng.standard_normal(sizes), "single")
    den = {}
    for variant in (11 if depth == 1 else 12, 9):
        w = ndwt.nd_dwt_2D(wn, sizes, "pres_l2_norm", 1, "precision", "single")
        w._plan(False, level, xg.device).set_variant(fwd=11 if variant != 9 else 9, inv=variant)
        den[variant] = {m: w.denoise(xg, level, 0.3, m) for m in ("soft", "hard")}
    dv = list(den.values())
    for m in ("soft", "hard"):
        assert float((dv[0][m] - dv[1][m]).abs().max()) <= 4 * TOL["single"] * float(dv[1][m].abs().max()), m
    a, b = list(res.values())
    # (the one-level kernels run 4 / 8 / 12 taps as packed FMAs like the cascade -- the same bits -- and other tap lengths as scalar ones)
    Lp = max(len(ndwt.wave_filters(v)[0]) for v in wl)
    assert float((a - b).abs().max()) <= (0.0 if Lp in (4, 8) else 2 * TOL["single"] * float(b.abs().max()))


def test_cascaded_2d_analysis_is_the_default_at_cfg2_size():
    """4096^2 db4, 3 levels (BASELINE config 2): one launch for the three levels (ndwt_plan_get_profile counts launches), same bits"""
    torch.manual_seed(3)
    n = 4096
    x = torch.randn(n, n, device="cuda")
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    nb = api.num_bands(2, 3)
    s = torch.cuda.current_stream().cuda_stream
    ys = []
    for variant, launches in ((0, 1), (9, 3)):
        p = api.Plan([n, n], ["db4"] * 2, torch.float32, False, True, "reference", max_level=3).set_variant(fwd=variant)
        y = torch.empty((nb, n, n), device="cuda")
        p.set_profiling(True)
        p.dec(x.data_ptr(), y.data_ptr(), 3, s)
        torch.cuda.synchronize()
        assert p.get_profile(0)[1] == launches
        ys.append(y)
    assert float((ys[0] - ys[1]).abs().max()) == 0.0
    rs = []
    for variant, launches in ((0, 1), (9, 3)):                # ... and the synthesis side
        p = api.Plan([n, n], ["db4"] * 2, torch.float32, False, True, "reference", max_level=3).set_variant(inv=variant)
        r = torch.empty(n, n, device="cuda")
        p.set_profiling(True)
        p.rec(ys[0].data_ptr(), r.data_ptr(), 3, s)
        torch.cuda.synchronize()
        assert p.get_profile(1)[1] == launches
        rs.append(r)
    assert float((rs[0] - rs[1]).abs().max()) == 0.0
    assert float((rs[0] - x).norm() / x.norm()) < 1e-6


def _two_rank_worker(rank, world, port, q):
    """one of `world` processes that share cuda:0; slabs are exchanged over gloo (host staged)"""
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
        n1, n2, n3, level = 72, 40, 50, 3                    # 50 planes over 3 ranks: uneven slabs
        dev = torch.device("cuda", 0)
        torch.manual_seed(5)
        xs = torch.randn(n3, n2, n1, device=dev)               # same volume on every rank
        errs = []
        for wname in ("db4", "db6", "db1"):                    # db6: cfg4's wavelet (halo 5 + 6 planes); db1: no plane before the slab
            w = ndwt.nd_dwt_3D(wname, [n1, n2, n3], "pres_l2_norm", 1, "precision", "single")
            yref = w.dec(xs.permute(2, 1, 0), level).permute(3, 2, 1, 0)
            for overlap in (True, False):
                eng = sh.ShardedNdDwt([wname] * 3, [n1, n2, n3], pres_l2_norm=True, precision="single", device=dev, overlap=overlap)
                assert eng.scheme == "scatter" and eng.overlap == overlap and eng._host_stage
                yl = eng.dec(xs[eng.z0:eng.z1].contiguous(), level)
                e_dec = float((yl - yref[:, eng.z0:eng.z1]).abs().max() / yref.abs().max())
                xl = eng.rec(yl)
                e_rec = float((xl - xs[eng.z0:eng.z1]).abs().max())
                errs.append((e_dec, e_rec))
        q.put((rank, errs))
    finally:
        dist.destroy_process_group()


def _t_sharded_worker(rank, world, port, q):
    """4-D volume sharded on t (cfg5's decomposition): scatter-add synthesis through the zero-extended 4-D entry point"""
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
        sizes, level = [24, 20, 12, 18], 2                     # t = 18 frames over 2 ranks: 9 each, halo 3 + 4
        dev = torch.device("cuda", 0)
        torch.manual_seed(9)
        xs = torch.randn(*reversed(sizes), device=dev)
        w = ndwt.nd_dwt_4D("db4", sizes, "pres_l2_norm", 1, "precision", "single")
        yref = w.dec(xs.permute(3, 2, 1, 0), level).permute(4, 3, 2, 1, 0)
        eng = sh.ShardedNdDwt(["db4"] * 4, sizes, pres_l2_norm=True, precision="single", device=dev)
        assert eng.scheme == "scatter" and not eng.overlap
        yl = eng.dec(xs[eng.z0:eng.z1].contiguous(), level)
        e_dec = float((yl - yref[:, eng.z0:eng.z1]).abs().max() / yref.abs().max())
        e_rec = float((eng.rec(yl) - xs[eng.z0:eng.z1]).abs().max())
        c = torch.randn_like(yref)
        want = w.rec(c.permute(4, 3, 2, 1, 0)).permute(3, 2, 1, 0)
        e_rec2 = float((eng.rec(c[:, eng.z0:eng.z1].contiguous()) - want[eng.z0:eng.z1]).abs().max() / want.abs().max())
        q.put((rank, [(e_dec, e_rec), (e_rec2, 0.0)]))
    finally:
        dist.destroy_process_group()


def _thin_slab_worker(rank, world, port, q):
    """cfg5's regime on the product engine: slabs THINNER than the filter (4 planes per rank, db4 = 8 taps: halo 3 + 4 >= slab, so
    the exchange is multi-hop and the plans are slab plans), 4-D sharded on t and 3-D sharded on z; and the a-trous dilation
    through the gather scheme of the HIP engine"""
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
        dev = torch.device("cuda", 0)
        errs = []
        for sizes, wn, level, dilation in (([24, 20, 12, 16], "db4", 2, "reference"),       # 4-D, t = 16 over 4 ranks
                                            ([40, 24, 16], "db4", 3, "reference"),           # 3-D, z = 16 over 4 ranks
                                            ([24, 20, 16], "db2", 2, "atrous"),              # dilated taps: gather scheme, per-axis kernels
                                            ([40, 24, 18], "db6", 2, "reference")):          # uneven 4/5/4/5 planes, 12 taps
            d = len(sizes)
            torch.manual_seed(21)
            xs = torch.randn(*reversed(sizes), device=dev)
            w = _cls(d)(wn, sizes, "pres_l2_norm", 1, "precision", "single", "dilation", dilation)
            perm = tuple(reversed(range(d)))
            yref = w.dec(xs.permute(*perm), level).permute(*reversed(range(d + 1)))
            eng = sh.ShardedNdDwt([wn] * d, sizes, pres_l2_norm=True, precision="single", dilation=dilation, device=dev)
            assert eng.n_local < len(ndwt.wave_filters(wn)[0]) or dilation == "atrous" or wn == "db6"
            yl = eng.dec(xs[eng.z0:eng.z1].contiguous(), level)
            e_dec = float((yl - yref[:, eng.z0:eng.z1]).abs().max() / yref.abs().max())
            e_rec = float((eng.rec(yl) - xs[eng.z0:eng.z1]).abs().max())
            yl2 = eng.dec(xs[eng.z0:eng.z1].contiguous(), level)           # second call: the cached scratch buffers are reused
            e_again = float((yl2 - yl).abs().max())
            errs.append((max(e_dec, e_again), e_rec))
        q.put((rank, errs))
    finally:
        dist.destroy_process_group()


def test_thin_slabs_and_atrous_on_the_hip_engine_over_gloo():
    for rank, errs in _run_ranks(_thin_slab_worker, 4):
        for e_a, e_b in errs:
            assert e_a <= 4e-6 and e_b <= 2e-5, (rank, errs)


def test_slab_plan_checks_the_whole_axis_not_the_slab():
    """ndwt_plan_create_slab: the reference's filter-length check (nd_dwt_3D.m:277-286) applies to the sharded axis of the whole
    volume; a thin slab plan serves the slab entry points only"""
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    with pytest.raises(ndwt.NdwtError, match="Fourth Dimension of Data is shorter"):
        api.Plan([32, 32, 16, 4], ["db4"] * 4, torch.float32)
    p = api.Plan([32, 32, 16, 4], ["db4"] * 4, torch.float32, global_outer=32)
    assert p.slab_halo(1) == (3, 4, 4, 3)
    with pytest.raises(ndwt.NdwtError, match="slab"):
        p.dec(1, 1, 1)
    with pytest.raises(ndwt.NdwtError, match="Fourth Dimension of Data is shorter"):
        api.Plan([32, 32, 16, 4], ["db4"] * 4, torch.float32, global_outer=6)
    with pytest.raises(ndwt.NdwtError, match="global_outer"):
        api.Plan([32, 32, 16, 4], ["db4"] * 4, torch.float32, global_outer=3)


@pytest.mark.parametrize("dims,wn,level,precision,cplx,dilation,nslab,exact", [
    ([72, 40, 50], ["db4", "db4", "db4"], 3, "single", False, "reference", 3, True),      # fused 3-D slabs, uneven 16 / 17 / 17 planes
    ([40, 24, 18], ["db2", "db1", "db6"], 2, "double", False, "reference", 4, True),      # thinner than the 12-tap filter: multi-hop halos
    ([24, 20, 12, 16], ["db4"] * 4, 2, "single", False, "reference", 4, True),            # 4-D, 4 frames per slab under 8 taps (cfg5's regime)
    # the slab kernels of these three differ from the single-device ones (outer filter shorter than the padded length; dilated
    # sub-lattice kernels; 2-D): same numbers to rounding, not bit for bit
    ([36, 30, 16], ["db3", "db2", "db2"], 2, "double", True, "reference", 2, False),      # interleaved complex
    ([24, 20, 16], ["db2", "db2", "db2"], 3, "single", False, "atrous", 2, False),        # dilated taps: halo of 8 planes at level 3
    ([70, 45], ["db2", "db4"], 3, "double", False, "reference", 3, False),                # 2-D, sharded on its second axis
])
def test_single_process_multi_device_plan(dims, wn, level, precision, cplx, dilation, nslab, exact):
    """ndwt_mplan_*: one host thread, several slabs (here all on the one GPU: a device may be listed more than once), halo
    planes moved between slabs by device-to-device copies ordered with events.  Gather exchange in both directions: where the
    slab and the single-device transform run the same kernels the results are equal BIT FOR BIT; always the oracle's within the
    usual tolerance."""
    import ctypes
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    d = len(dims)
    rng = np.random.default_rng(88)
    x = rng.standard_normal(dims) + (1j * rng.standard_normal(dims) if cplx else 0)
    rdt = np.float32 if precision == "single" else np.float64
    cdt = (np.complex64 if precision == "single" else np.complex128) if cplx else rdt
    xk = np.ascontiguousarray(x.T).astype(cdt)
    tdt = torch.float32 if precision == "single" else torch.float64
    mp = api.MultiPlan(dims, wn, tdt, [0] * nslab, cplx, True, dilation, max_level=level)
    sl = mp.slabs()
    assert len(sl) == nslab and sl[0][1] == 0 and sum(s[2] for s in sl) == dims[-1]
    scatter = "scatter-add" in mp.describe()              # fused 3-D plans at tap stride 1; everything else gathers anyway
    assert scatter == (d == 3 and exact) and "peer access between all devices" in mp.describe()
    yk = mp.dec(xk, level)
    want = orc.spatial_dec(x, wn, level, 1, dilation)
    assert _relerr(yk.T, want) <= TOL[precision]
    # the single-device host entry points on the same data
    p1 = api.Plan(dims, wn, tdt, cplx, True, dilation, max_level=level)
    y1 = np.empty_like(yk)
    ndwt._lib.check(ndwt.lib().ndwt_dec_host(p1._h, xk.ctypes.data_as(ctypes.c_void_p), y1.ctypes.data_as(ctypes.c_void_p), level))
    assert np.array_equal(yk, y1) if exact else _relerr(yk, y1) <= TOL[precision]     # analysis: halo planes gathered, bit-identical
    c = (rng.standard_normal(yk.shape) + (1j * rng.standard_normal(yk.shape) if cplx else 0)).astype(cdt)
    r1 = np.empty(xk.shape, dtype=cdt)
    ndwt._lib.check(ndwt.lib().ndwt_rec_host(p1._h, c.ctypes.data_as(ctypes.c_void_p), r1.ctypes.data_as(ctypes.c_void_p), level))
    for scheme in ("scatter", "gather"):
        mp.set_exchange(scheme)
        r = mp.rec(c)
        # gather reproduces one device bit for bit; the scatter-add synthesis sums the same products in another order
        bitwise = exact and (scheme == "gather" or not scatter)
        assert np.array_equal(r, r1) if bitwise else _relerr(r, r1) <= 4 * TOL[precision], scheme
        assert _relerr(r.T, orc.spatial_rec(np.transpose(c), wn, 1, dilation)) <= 4 * TOL[precision]
        assert np.array_equal(r, mp.rec(c))                                           # deterministic: a fixed order of summation
        assert _relerr(mp.rec(yk), xk) <= 20 * TOL[precision]
    mp.set_exchange("scatter")
    # the overlapped schedule (copy streams; interior planes, then the ends / partial sums first, then the own planes) against the
    # plain one: the same kernels on the same planes and the same order of summation -- the same bits (2: partial sums staged
    # through the receive buffers, the path between different devices)
    fast = "slabs read in place" in mp.describe()        # fused 3-D slab plans: the only ones with an overlapped schedule
    assert ("overlapped" in mp.describe()) == fast
    ref = {}
    for ov in (0, 1, 2):
        mp.set_overlap(ov)
        yo, ro = mp.dec(xk, level), mp.rec(c)
        if ov == 0:
            assert "exchange, then compute" in mp.describe()
            ref = {"y": yo, "r": ro}
        assert np.array_equal(yo, ref["y"]) and np.array_equal(ro, ref["r"]), ov
        assert np.array_equal(yo, yk)
    mp.set_overlap(1)
    # one host thread per slab queues that slab's work (default) / the calling thread queues everything: the same work on the same streams
    for th in (False, True, False, True):
        mp.set_threads(th)
        for _ in range(2):
            assert np.array_equal(mp.dec(xk, level), ref["y"]) and np.array_equal(mp.rec(c), ref["r"]), th
    # device-resident form: one tensor per slab, read and written in place
    dev = torch.device("cuda", 0)
    xs = [torch.from_numpy(xk[z0:z0 + n]).to(dev) for _, z0, n in sl]
    ys = mp.dec_device(xs, level)
    assert all(np.array_equal(yd.cpu().numpy(), yk[:, z0:z0 + n]) for yd, (_, z0, n) in zip(ys, sl))
    assert all(torch.equal(xd, torch.from_numpy(xk[z0:z0 + n]).to(dev)) for xd, (_, z0, n) in zip(xs, sl))     # inputs untouched
    cs = [torch.from_numpy(np.ascontiguousarray(c[:, z0:z0 + n])).to(dev) for _, z0, n in sl]
    rs = mp.rec_device(cs)
    r = mp.rec(c)
    assert all(np.array_equal(rd.cpu().numpy(), r[z0:z0 + n]) for rd, (_, z0, n) in zip(rs, sl))
    with pytest.raises(ndwt.NdwtError, match="max_level"):
        mp.dec(xk, level + 1)
    # the same through the class API: the 'devices' option of the drop-in classes (host arrays, compute = 'hip_off')
    w = _cls(d)(wn, dims, "pres_l2_norm", 1, "precision", precision, "dilation", dilation, "compute", "hip_off", "devices", [0] * nslab)
    yc = w.dec(x, level)
    assert isinstance(yc, np.ndarray) and yc.shape == tuple(dims) + (orc.num_bands(d, level),)
    assert np.array_equal(yc, yk.T)
    assert _relerr(w.rec(yc), x) <= 20 * TOL[precision]


def _run_ranks(worker, world):
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _rccl_self_worker(rank, world, port, q):
    """ONE rank in an `nccl` (= RCCL) group: the sharded driver with its self-segments routed through grouped send / receive (test hook
    `_self_p2p`) instead of local copies, so that the RCCL branch runs on one GPU: device buffers sent as they are, halo planes received
    in place in the margins of the [halo | slab | halo] scratch, `scatter_recv` buffers reused across calls, every work.wait()."""
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
        errs, info = [], []
        xs3 = None
        for sizes, wn, level in (([72, 40, 50], "db4", 3), ([136, 64, 40], "db6", 2), ([24, 20, 12, 18], "db4", 2)):
            d = len(sizes)
            torch.manual_seed(11)
            xs = torch.randn(*reversed(sizes), device=dev)
            xs3 = xs if xs3 is None else xs3
            w = _cls(d)(wn, sizes, "pres_l2_norm", 1, "precision", "single")
            yref = w.dec(xs.permute(*reversed(range(d))), level).permute(*reversed(range(d + 1)))
            c = torch.randn_like(yref)
            want = w.rec(c.permute(*reversed(range(d + 1)))).permute(*reversed(range(d)))
            cases = ((True, False, "torch"), (True, True, "torch"), (False, False, "torch"), (False, False, "rccl"), (True, True, "rccl")) if d == 3 else \
                    ((False, False, "torch"), (False, False, "rccl"))
            for overlap, two, transport in cases:             # transport "rccl": RCCL calls on the transform's own stream (ndwt_comm_*)
                try:
                    eng = sh.ShardedNdDwt([wn] * d, sizes, pres_l2_norm=True, precision="single", device=dev, overlap=overlap,
                                          two_streams=two, transport=transport, _self_p2p=True)
                    assert eng.scheme == "scatter" and not eng._host_stage and eng.world == 1
                    for rep in range(3):                            # later calls reuse the cached scratch / receive buffers
                        yl = eng.dec(xs, level)
                        e_dec = float((yl - yref).abs().max() / yref.abs().max())
                        e_rec = float((eng.rec(yl) - xs).abs().max())
                        e_rec2 = float((eng.rec(c) - want).abs().max() / want.abs().max())
                        errs.append((max(e_dec, e_rec2), e_rec))
                except Exception as exc:                            # e.g. RCCL refusing a send to self: reported, not hidden
                    info.append(f"{sizes} {wn} overlap={overlap}: {type(exc).__name__}: {exc}"[:300])
        # tune(): every schedule of both transports measured, one kept; the result still equals the single-device transform
        eng = sh.ShardedNdDwt(["db4"] * 3, [72, 40, 50], pres_l2_norm=True, precision="single", device=dev, _self_p2p=True)
        rec = eng.tune(xs3, 3, steps=2)
        if not all(k in rec for k in ("ms_one_piece", "ms_overlap", "ms_overlap_two_streams", "ms_rccl_one_piece", "ms_rccl_overlap_two_streams")):
            info.append(f"tune() did not measure every schedule: {rec}")
        e_t = float((eng.rec(eng.dec(xs3, 3)) - xs3).abs().max())
        errs.append((0.0, e_t))
        torch.cuda.synchronize(dev)
        q.put((rank, errs, info))
    finally:
        dist.destroy_process_group()


def test_rccl_branch_of_the_sharded_driver_on_one_gpu():
    """the `nccl` branch of sharded.py executed for real: a 1-rank RCCL group, segments to self through P2POp"""
    (rank, errs, info), = _run_ranks(_rccl_self_worker, 1)
    assert not info, info
    assert len(errs) == 37
    for e_a, e_b in errs:
        assert e_a <= 4e-6 and e_b <= 2e-5, errs


def test_t_sharded_4d_scatter_over_gloo():
    for rank, errs in _run_ranks(_t_sharded_worker, 2):
        for e_a, e_b in errs:
            assert e_a <= 4e-6 and e_b <= 1e-5, (rank, errs)


def test_three_ranks_share_one_gpu_over_gloo():
    """the product slab engine (HIP kernels, run-of-planes entry points, in-place halo margins) under a real
    multi-process exchange: 3 processes on cuda:0, gloo with host staging standing in for RCCL"""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, errs in res:
        for e_dec, e_rec in errs:
            assert e_dec <= 2e-6 and e_rec <= 1e-5, (rank, errs)


@pytest.mark.parametrize("sizes,wn,level", [
    ([64, 32, 16], "db4", 3),            # one tile
    ([72, 40, 33], "db4", 2),            # ragged tiles, odd plane count
    ([132, 70, 20], "db3", 3),
    ([128, 64, 64], "db2", 1),           # a single level: the approximation goes straight from the band-0 analysis to Den3
    ([68, 36, 9], "db1", 2),
    ([16, 12, 10], "db4", 2),            # smaller than the tile and its halo: every axis wraps
])
@pytest.mark.parametrize("l2", [0, 1])
def test_denoise_with_fused_level1_against_numpy(sizes, wn, level, l2):
    """ndwt_denoise where the finest level never materialises its detail bands (Den3: recomputed from x, thresholded in registers)
    against dec -> shrink -> rec of the oracle and against the library's own materialising path"""
    rng = np.random.default_rng(41)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_3D(wn, sizes, "pres_l2_norm", l2, "precision", "single")
    xg = _colmajor_gpu(x, "single")
    wl = [wn] * 3
    y_ref = orc.spatial_dec(x, wl, level, l2)
    thr = 0.35 * (8.0 ** 0.5 if not l2 else 1.0)
    w.denoise(xg, level, thr)                                 # (creates the plan)
    plan = list(w._plans.values())[0]
    for mode in ("soft", "hard"):
        plan.set_fused_level1(2)                              # 8 taps too (off by default there: not faster)
        got = w.denoise(xg, level, thr, mode)
        plan.set_fused_level1(0)
        mat = w.denoise(xg, level, thr, mode)
        plan.set_fused_level1(1)
        want = orc.spatial_rec(_np_shrink(y_ref, thr, mode == "hard"), wl, l2)
        scale = max(np.abs(want).max(), 1.0)
        # hard thresholding is discontinuous: coefficients within rounding of the threshold may fall on either side in fp32
        frac = 2e-3 if mode == "hard" else 0.0
        for name, a in (("numpy", want), ("materialised", mat.cpu().numpy())):
            bad = np.abs(got.cpu().numpy() - a) > 2e-5 * scale
            assert bad.mean() <= frac, (name, mode, float(np.abs(got.cpu().numpy() - a).max()))
    assert _relerr(w.denoise(xg, level, 0.0).cpu().numpy(), x) < 20 * TOL["single"]      # threshold 0 = identity


def test_one_plan_serves_several_torch_streams():
    """the classes keep ONE plan per (data kind, device); a call on another torch stream first waits for the work the plan's previous
    stream still has queued (the plan's scratch is never shared by two streams in flight, and no scratch accumulates per stream)"""
    sizes = [96, 64, 48]
    rng = np.random.default_rng(5)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_3D("db4", sizes, "pres_l2_norm", 1, "precision", "single")
    xg = _colmajor_gpu(x, "single")
    want = w.dec(xg, 3)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = []
    for rep in range(4):
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                y = w.dec(xg, 3)
                outs.append((y, w.rec(y)))
    torch.cuda.synchronize()
    assert len(w._plans) == 1
    for y, r in outs:
        assert torch.equal(y, want)
        assert _relerr(r.cpu().numpy(), x) < 20 * TOL["single"]


@pytest.mark.parametrize("d,sizes,wn,precision", [(3, [64, 40, 36], "db4", "single"), (2, [260, 64], "db3", "single"), (1, [4096], "db2", "double"),
                                                  (3, [48, 40, 36], "db2", "double"), (4, [24, 20, 16, 12], "db2", "single")])
def test_dec_rec_and_denoise_are_graph_capturable(d, sizes, wn, precision):
    """a long-lived plan's dec / rec / denoise enqueue kernels and nothing else (no allocation, copy from the host or synchronisation once
    the plan is warm), so a solver iteration can be captured into a HIP graph and replayed; the replay reproduces the eager results on
    new input data"""
    dt = torch.float32 if precision == "single" else torch.float64
    plan = ndwt.Plan(sizes, [wn] * d, dt, False, True, "reference", max_level=2)
    nb = ndwt.num_bands(d, 2)
    shp = tuple(reversed(sizes))
    x = torch.randn(*shp, device="cuda", dtype=dt)
    y = torch.empty((nb,) + shp, device="cuda", dtype=dt)
    r, den = torch.empty_like(x), torch.empty_like(x)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):                                  # warm: scratch of the plan allocated outside the capture
            plan.dec(x.data_ptr(), y.data_ptr(), 2, st.cuda_stream)
            plan.rec(y.data_ptr(), r.data_ptr(), 2, st.cuda_stream)
            plan.denoise(x.data_ptr(), den.data_ptr(), 2, 0.3, 0, st.cuda_stream)
    st.synchronize()
    y0, r0, d0 = y.clone(), r.clone(), den.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        plan.dec(x.data_ptr(), y.data_ptr(), 2, st.cuda_stream)
        plan.rec(y.data_ptr(), r.data_ptr(), 2, st.cuda_stream)
        plan.denoise(x.data_ptr(), den.data_ptr(), 2, 0.3, 0, st.cuda_stream)
    y.zero_(); r.zero_(); den.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, y0) and torch.equal(r, r0) and torch.equal(den, d0)
    x2 = torch.randn_like(x)
    x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    yg, rg, dg = y.clone(), r.clone(), den.clone()
    plan.dec(x.data_ptr(), y.data_ptr(), 2, 0)
    plan.rec(y.data_ptr(), r.data_ptr(), 2, 0)
    plan.denoise(x.data_ptr(), den.data_ptr(), 2, 0.3, 0, 0)
    torch.cuda.synchronize()
    assert torch.equal(y, yg) and torch.equal(r, rg) and torch.equal(den, dg)


@pytest.mark.parametrize("sizes,wn,path", [([260, 96], "db7", "fused2d"), ([512, 70], "db8", "fused2d"), ([250, 65], "db9", "fused2d"),
                                           ([264, 40], "db10", "fused2d"), ([236, 33], ["db10", "db7"], "fused2d"), ([128, 48], ["db2", "db9"], "fused2d")])
def test_long_filters_2d_float(sizes, wn, path):
    """db7 .. db10 on real float images: the register-only fused 2-D kernels on the 256-register budget (no spills), against the oracle"""
    rng = np.random.default_rng(12)
    wl = [wn] * 2 if isinstance(wn, str) else wn
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_2D(wn, sizes, "pres_l2_norm", 1, "precision", "single")
    xg = _colmajor_gpu(x, "single")
    y = w.dec(xg, 2)
    assert list(w._plans.values())[0].describe() == path
    assert _relerr(y.cpu().numpy(), orc.spatial_dec(x, wl, 2, 1)) <= TOL["single"]
    c = rng.standard_normal(sizes + [7])
    got = w.rec(_colmajor_gpu(c, "single")).cpu().numpy()
    want = orc.spatial_rec(c, wl, 1)
    assert np.abs(got - want).max() <= TOL["single"] * max(np.abs(want).max(), np.abs(c).max())
    assert _relerr(w.rec(y).cpu().numpy(), x) < 1e-5


@pytest.mark.parametrize("sizes,wn,precision,cplx,path", [
    ([130, 48], "db5", "single", True, "fused2d"), ([128, 40], "db6", "single", True, "fused2d"), ([131, 36], "db7", "single", True, "fused2d"),
    ([132, 33], ["db8", "db5"], "single", True, "fused2d"), ([128, 40], "db9", "single", True, "axis"),
    ([260, 40], "db7", "double", False, "fused2d"), ([250, 33], ["db8", "db2"], "double", False, "fused2d"), ([128, 40], "db9", "double", False, "axis"),
    ([128, 40], "db5", "double", True, "axis"),
])
def test_long_filters_2d_complex64_and_double(sizes, wn, precision, cplx, path):
    """2-D fused kernels beyond db6 / db4: interleaved complex64 with 10 .. 16 taps and real double with 14 / 16 taps (256-register budget)"""
    rng = np.random.default_rng(14)
    wl = [wn] * 2 if isinstance(wn, str) else wn
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    w = ndwt.nd_dwt_2D(wn, sizes, "pres_l2_norm", 1, "precision", precision)
    xg = _colmajor_gpu(x, precision)
    y = w.dec(xg, 2)
    assert list(w._plans.values())[0].describe() == path
    assert _relerr(y.cpu().numpy(), orc.spatial_dec(x, wl, 2, 1)) <= TOL[precision]
    c = rng.standard_normal(sizes + [7]) + (1j * rng.standard_normal(sizes + [7]) if cplx else 0)
    got = w.rec(_colmajor_gpu(c, precision)).cpu().numpy()
    want = orc.spatial_rec(c, wl, 1)
    assert np.abs(got - want).max() <= TOL[precision] * max(np.abs(want).max(), np.abs(c).max())
    assert _relerr(w.rec(y).cpu().numpy(), x) < (1e-13 if precision == "double" else 1e-5)


@pytest.mark.parametrize("sizes,wn,precision", [([260, 96], "db4", "double"), ([512, 70], ["db2", "db3"], "double"), ([248, 64], "db1", "double"),
                                                ([264, 130], "db6", "single"), ([256, 64], ["db2", "db2"], "single")])
def test_2d_synthesis_with_rows_in_flight(sizes, wn, precision):
    """Inv2P on the device: double (2 / 4 rows of band loads in flight, up to 8 taps) and the packed float form (4 / 8 / 12 taps) against
    the oracle, and denoise (thresholding fused into those loads) against the separate pass"""
    rng = np.random.default_rng(11)
    wl = [wn] * 2 if isinstance(wn, str) else wn
    c = rng.standard_normal(sizes + [7])
    w = ndwt.nd_dwt_2D(wn, sizes, "pres_l2_norm", 1, "precision", precision)
    got = w.rec(_colmajor_gpu(c, precision)).cpu().numpy()
    want = orc.spatial_rec(c, wl, 1)
    assert np.abs(got - want).max() <= TOL[precision] * max(np.abs(want).max(), np.abs(c).max())
    x = rng.standard_normal(sizes)
    xg = _colmajor_gpu(x, precision)
    y = w.dec(xg, 2)
    assert _relerr(w.rec(y).cpu().numpy(), x) < (1e-13 if precision == "double" else 1e-5)
    den = w.denoise(xg, 2, 0.4, "soft")
    ref = w.rec(w.shrink(y.clone(), 0.4, "soft"))
    assert float((den - ref).abs().max()) <= 20 * TOL[precision] * max(float(ref.abs().max()), 1.0)


def test_sharded_driver_band_pitch_option():
    """ShardedNdDwt(band_pitch='packed') returns a contiguous coefficient slab (for callers that pass it to collectives or take raw
    pointers); the default 'auto' returns the pitched view; the values are the same"""
    sh = __import__("importlib").import_module("non-decimated_wavelets_amd.sharded")
    dev = torch.device("cuda", 0)
    x = torch.randn(24, 40, 72, device=dev)
    a = sh.ShardedNdDwt("db4", [72, 40, 24], pres_l2_norm=True, precision="single", device=dev)
    b = sh.ShardedNdDwt("db4", [72, 40, 24], pres_l2_norm=True, precision="single", device=dev, band_pitch="packed")
    ya, yb = a.dec(x, 2), b.dec(x, 2)
    assert yb.is_contiguous() and not ya.is_contiguous() and torch.equal(ya, yb)
    assert torch.equal(a.rec(ya), b.rec(yb)) and yb.view(-1).numel() == yb.numel()
    with pytest.raises(ValueError, match="band_pitch"):
        sh.ShardedNdDwt("db4", [72, 40, 24], device=dev, band_pitch="odd")


def test_4d_analysis_with_folded_t_axis_variant():
    """A/B variant 7 of the 4-D analysis (the t axis folded into the fused launches, 17 instead of 21 volume transfers per level; measured
    slower -- more loads through the vector-memory pipe -- and therefore not the default): the same coefficients as the default path"""
    sizes = [64, 32, 12, 9]
    rng = np.random.default_rng(71)
    x = rng.standard_normal(sizes)
    w = ndwt.nd_dwt_4D("db4", sizes, "pres_l2_norm", 1, "precision", "single")
    xg = _colmajor_gpu(x, "single")
    y0 = w.dec(xg, 2)
    list(w._plans.values())[0].set_variant(fwd=7)
    y7 = w.dec(xg, 2)
    list(w._plans.values())[0].set_variant(fwd=0)
    want = orc.spatial_dec(x, ["db4"] * 4, 2, 1)
    assert _relerr(y7.cpu().numpy(), want) <= TOL["single"] and _relerr(y0.cpu().numpy(), want) <= TOL["single"]
    assert float((y7 - y0).abs().max()) <= 2e-6 * float(y0.abs().max())


def test_denoise_fused_level1_random_shapes_and_in_place():
    """ndwt_denoise with level 1 in one launch (Den3) against the materialising path of the same plan over random shapes (ragged tiles,
    volumes smaller than the halo, 1 .. 3 levels, db1 .. db4); x == out takes the materialising path and gives the same values"""
    api = __import__("importlib").import_module("non-decimated_wavelets_amd.api")
    rng = np.random.default_rng(2024)
    s = torch.cuda.current_stream().cuda_stream
    for case in range(14):
        K = int(rng.integers(1, 5))
        n1 = 4 * int(rng.integers(max(1, (2 * K + 3) // 4), 40))
        n2, n3 = int(rng.integers(2 * K, 70)), int(rng.integers(2 * K, 40))
        level = int(rng.integers(1, 4))
        hard = bool(rng.integers(0, 2))
        plan = api.Plan([n1, n2, n3], [f"db{K}"] * 3, torch.float32, False, bool(rng.integers(0, 2)), "reference", max_level=3)
        x = torch.randn(n3, n2, n1, device="cuda")
        fused, mat, inplace = torch.empty_like(x), torch.empty_like(x), x.clone()
        plan.set_fused_level1(2)
        plan.denoise(x.data_ptr(), fused.data_ptr(), level, 0.3, hard, s)
        plan.denoise(inplace.data_ptr(), inplace.data_ptr(), level, 0.3, hard, s)      # not fusable in place
        plan.set_fused_level1(0)
        plan.denoise(x.data_ptr(), mat.data_ptr(), level, 0.3, hard, s)
        torch.cuda.synchronize()
        assert torch.equal(inplace, mat), (case, n1, n2, n3, K, level)
        scale = max(float(mat.abs().max()), 1.0)
        bad = (fused - mat).abs() > 2e-5 * scale
        assert float(bad.float().mean()) <= (2e-3 if hard else 0.0), (case, n1, n2, n3, K, level, float((fused - mat).abs().max()))
