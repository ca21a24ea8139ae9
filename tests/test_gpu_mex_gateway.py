"""The MATLAB gateway (matlab/nd_dwt_hip_mex.c) EXECUTED: compiled against the declaration stubs, linked with a mock mx / mex runtime
(tests/mexmock/mock_mx.c) and libndwt_hip.so, and driven from here the way MATLAB would call it -- y = nd_dwt_hip_mex(x, wnames, dir,
level, l2, ...) and the string commands (dec_keep / rec_handle / shrink / fetch / release / denoise).  MATLAB itself is not on the image:
this pins the gateway's own logic (argument parsing, 1-D column vectors, the band-count guard of nd_dwt_mex.c:124-127, plan cache and
eviction, split and interleaved complex storage, the handle registry), not MATLAB's behaviour.  Reference call sites: nd_dwt_3D.m:161,225.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import ndwt_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = {np.float64: 1e-12, np.float32: 2e-6}


@pytest.fixture(scope="module", params=["split", "interleaved"])
def gw(request):
    """the gateway built for the split-complex mex API (mxGetData / mxGetImagData, what the reference's own gateway uses) and for -R2018a"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "mexmock"), f"libmexmock_{request.param}.so"])
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "mexmock", f"libmexmock_{request.param}.so"))
    vp = ctypes.c_void_p
    lib.mock_numeric.restype = vp
    lib.mock_numeric.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, vp, vp]
    lib.mock_uint64.restype = vp
    lib.mock_uint64.argtypes = [ctypes.c_uint64]
    lib.mock_string.restype = vp
    lib.mock_string.argtypes = [ctypes.c_char_p]
    lib.mock_cell.restype = vp
    lib.mock_cell.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.mock_free.argtypes = [vp]
    lib.mock_call.argtypes = [ctypes.c_int, ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(vp)]
    lib.mock_last_error.restype = ctypes.c_char_p
    for f in ("mock_ndim", "mock_is_single", "mock_is_complex"):
        getattr(lib, f).argtypes = [vp]
    lib.mock_dim.argtypes = [vp, ctypes.c_int]
    lib.mock_dim.restype = ctypes.c_int64
    lib.mock_real.argtypes = [vp]
    lib.mock_real.restype = vp
    lib.mock_imag.argtypes = [vp]
    lib.mock_imag.restype = vp
    g = Gateway(lib)
    yield g
    lib.mock_clear()                                       # `clear mex`: mexAtExit releases handles, then plans


class MexError(RuntimeError):
    pass


class Gateway:
    def __init__(self, lib):
        self.lib = lib
        self.interleaved = bool(lib.mock_interleaved())

    # numpy (MATLAB shape) -> mock mxArray (column-major memory) and back
    def arr(self, a):
        a = np.asarray(a)
        dims = (ctypes.c_int64 * a.ndim)(*a.shape)
        single = a.dtype in (np.float32, np.complex64)
        f = np.asfortranarray(a)
        if np.iscomplexobj(a):
            if self.interleaved:
                return self.lib.mock_numeric(a.ndim, dims, int(single), 1, f.ctypes.data_as(ctypes.c_void_p), None)
            re, im = np.asfortranarray(f.real), np.asfortranarray(f.imag)
            return self.lib.mock_numeric(a.ndim, dims, int(single), 1, re.ctypes.data_as(ctypes.c_void_p), im.ctypes.data_as(ctypes.c_void_p))
        return self.lib.mock_numeric(a.ndim, dims, int(single), 0, f.ctypes.data_as(ctypes.c_void_p), None)

    def scalar(self, v):
        return self.arr(np.array([[float(v)]]))

    def names(self, wn):
        if isinstance(wn, str):
            return self.lib.mock_string(wn.encode())
        items = (ctypes.c_void_p * len(wn))(*[self.lib.mock_string(w.encode()) for w in wn])
        return self.lib.mock_cell(len(wn), items)

    def to_numpy(self, h):
        nd = self.lib.mock_ndim(h)
        shape = [int(self.lib.mock_dim(h, i)) for i in range(nd)]
        rdt = np.float32 if self.lib.mock_is_single(h) else np.float64
        n = int(np.prod(shape))
        if self.lib.mock_is_complex(h):
            if self.interleaved:
                cdt = np.complex64 if rdt == np.float32 else np.complex128
                flat = np.frombuffer((ctypes.c_char * (n * np.dtype(cdt).itemsize)).from_address(self.lib.mock_real(h)), dtype=cdt).copy()
            else:
                re = np.frombuffer((ctypes.c_char * (n * np.dtype(rdt).itemsize)).from_address(self.lib.mock_real(h)), dtype=rdt)
                im = np.frombuffer((ctypes.c_char * (n * np.dtype(rdt).itemsize)).from_address(self.lib.mock_imag(h)), dtype=rdt)
                flat = re + 1j * im
        else:
            flat = np.frombuffer((ctypes.c_char * (n * np.dtype(rdt).itemsize)).from_address(self.lib.mock_real(h)), dtype=rdt).copy()
        return flat.reshape(shape, order="F")

    def call(self, *args, nlhs=1, raw=False):
        hs = []
        for a in args:
            if isinstance(a, (str, list, tuple)) and not isinstance(a, np.ndarray):
                hs.append(self.names(a))
            elif isinstance(a, np.ndarray):
                hs.append(self.arr(a))
            elif isinstance(a, Handle):
                hs.append(self.lib.mock_uint64(a.value))
            else:
                hs.append(self.scalar(a))
        out = (ctypes.c_void_p * 1)(None)
        rc = self.lib.mock_call(nlhs, out, len(hs), (ctypes.c_void_p * len(hs))(*hs))
        for h in hs:
            self.lib.mock_free(h)
        if rc:
            raise MexError(self.lib.mock_last_error().decode())
        if not out[0]:
            return None
        if raw:
            return out[0]
        res = self.to_numpy(out[0])
        self.lib.mock_free(out[0])
        return res


class Handle:
    def __init__(self, value):
        self.value = int(value)


def _relerr(got, want):
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))


@pytest.mark.parametrize("sizes,wn,level,dt,cplx", [
    ([72, 40, 33], "db4", 3, np.float32, False),
    ([164, 64, 40], ["db1", "db3", "db1"], 1, np.float64, True),      # Test/nddwt3D_test.m:5-11 (complex input, mixed wavelets)
    ([129, 131], ["db2", "db3"], 2, np.float64, True),                # mex/mex_test.m:48
    ([24, 20, 6, 5], "db2", 2, np.float32, False),
])
def test_gateway_dec_and_rec_like_the_reference_call(gw, sizes, wn, level, dt, cplx):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)
    x = x.astype((np.complex64 if dt == np.float32 else np.complex128) if cplx else dt)
    wl = [wn] * len(sizes) if isinstance(wn, str) else wn
    y = gw.call(x, wn if isinstance(wn, str) else list(wn), 0, level, 1)
    want = orc.spatial_dec(x.astype(np.complex128 if cplx else np.float64), wl, level, 1)
    assert y.shape == want.shape and y.dtype == x.dtype                 # output class follows the input (Test/nddwt3D_test.m:25)
    assert _relerr(y, want) <= TOL[dt]
    r = gw.call(y, wn if isinstance(wn, str) else list(wn), 1, level, 1)
    assert r.shape == tuple(sizes) and _relerr(r, x) <= 20 * TOL[dt]
    # the band-count guard of nd_dwt_mex.c:124-127, with the reference's identifier and text
    with pytest.raises(MexError, match="MATLAB:FFT2mx:invalidNumInputs: FIlter size and image size not consistant"):
        gw.call(y[..., :-1].copy(), wn if isinstance(wn, str) else list(wn), 1, level, 1)


def test_gateway_one_dimensional_column_vector_and_argument_errors(gw):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((4096, 1))                                  # a column vector is 1-D (nd_dwt_mex.c:68-70)
    y = gw.call(x, "db2", 0, 3, 0)
    want = orc.spatial_dec(x[:, 0], ["db2"], 3, 0)
    assert y.shape == (4096, 4) and _relerr(y, want) <= 1e-12
    assert _relerr(gw.call(y, "db2", 1, 3, 0)[:, 0], x[:, 0]) <= 1e-11
    with pytest.raises(MexError, match="Five Inputs Required"):
        gw.call(x, "db2", 0, 3)
    with pytest.raises(MexError, match="Too many output arguments"):
        gw.call(x, "db2", 0, 3, 0, nlhs=2)
    with pytest.raises(MexError, match="Unknown Wavelet Name"):
        gw.call(x, "haar", 0, 3, 0)
    with pytest.raises(MexError, match="level must be at least 1"):
        gw.call(x, "db2", 0, 0, 0)


def test_gateway_devices_option_shards_over_the_multi_device_plan(gw):
    """`devices` with several ordinals (here device 0 twice: two slabs on the one GPU) goes through ndwt_mdec_host / ndwt_mrec_host;
    `exchange` = 'gather' reproduces the single-device reconstruction bit for bit, the default scatter-add to rounding"""
    rng = np.random.default_rng(9)
    sizes, wn, level = [72, 40, 34], "db4", 2
    x = rng.standard_normal(sizes).astype(np.float32)
    y1 = gw.call(x, wn, 0, level, 1)
    dev = np.array([[0.0, 0.0]])
    y2 = gw.call(x, wn, 0, level, 1, "reference", dev)
    assert np.array_equal(y1, y2)
    c = rng.standard_normal(y1.shape).astype(np.float32)
    r1 = gw.call(c, wn, 1, level, 1)
    assert np.array_equal(gw.call(c, wn, 1, level, 1, "reference", dev, "gather"), r1)
    assert _relerr(gw.call(c, wn, 1, level, 1, "reference", dev, "scatter"), r1) <= 4e-6
    assert _relerr(gw.call(c, wn, 1, level, 1, "reference", dev), r1) <= 4e-6
    # a single ordinal selects that device for the ordinary plan
    assert np.array_equal(gw.call(x, wn, 0, level, 1, "reference", np.array([[0.0]])), y1)


def test_gateway_plan_cache_serves_many_configurations(gw):
    """more configurations than cache slots (8): least-recently-used plans are evicted and rebuilt, results stay right"""
    rng = np.random.default_rng(7)
    for rep in range(2):
        for k in range(10):
            n = 16 + 4 * k
            x = rng.standard_normal((n, 12)).astype(np.float32)
            y = gw.call(x, ["db2", "db1"], 0, 2, 1)
            assert _relerr(y, orc.spatial_dec(x.astype(np.float64), ["db2", "db1"], 2, 1)) <= 2e-6


@pytest.mark.parametrize("dt,cplx", [(np.float32, False), (np.float64, False), (np.float32, True)])
def test_gateway_device_resident_handle_commands(gw, dt, cplx):
    if cplx and not gw.interleaved:
        with pytest.raises(MexError, match="interleaved-complex API"):
            gw.call("dec_keep", (np.ones((8, 8, 8)) * (1 + 1j)).astype(np.complex64), "db1", 1, 1)
        return
    rng = np.random.default_rng(8)
    sizes, wn, level = [40, 24, 20], ["db2", "db4", "db1"], 2
    x = (rng.standard_normal(sizes) + (1j * rng.standard_normal(sizes) if cplx else 0)).astype(np.complex64 if cplx else dt)
    raw = gw.call("dec_keep", x, wn, level, 1, raw=True)                # (a uint64 scalar: read from the raw mock array)
    h = Handle(ctypes.c_uint64.from_address(gw.lib.mock_real(raw)).value)
    gw.lib.mock_free(raw)
    want = orc.spatial_dec(x.astype(np.complex128 if cplx else np.float64), wn, level, 1)
    y = gw.call("fetch", h)
    assert y.shape == want.shape and _relerr(y, want) <= TOL[dt]
    assert _relerr(gw.call("rec_handle", h), x) <= 20 * TOL[dt]
    assert gw.call("shrink", h, 0.5, "soft", nlhs=0) is None
    mag = np.abs(want)
    shr = np.where(mag > 0.5, want * (1 - 0.5 / np.maximum(mag, 1e-300)), 0)
    shr[..., 0] = want[..., 0]
    got = gw.call("fetch", h)
    near = np.abs(mag - 0.5) < 1e-4
    assert np.abs(np.where(near, 0, got - shr)).max() <= 10 * TOL[dt] * np.abs(want).max()
    den = gw.call("denoise", x, wn, level, 1, 0.5, "soft")
    assert _relerr(den, gw.call("rec_handle", h)) <= 20 * TOL[dt]
    gw.call("release", h, nlhs=0)
    with pytest.raises(MexError, match="stale or unknown coefficient handle"):
        gw.call("rec_handle", h)
    with pytest.raises(MexError, match="unknown command"):
        gw.call("frobnicate", h)
    # a plan with live handles is not evicted: fill the cache with other configurations, the handle still works
    raw = gw.call("dec_keep", x, wn, level, 1, raw=True)
    h2 = Handle(ctypes.c_uint64.from_address(gw.lib.mock_real(raw)).value)
    gw.lib.mock_free(raw)
    for k in range(9):
        gw.call(rng.standard_normal((16 + 4 * k, 12)).astype(np.float32), ["db1", "db1"], 0, 1, 0)
    assert _relerr(gw.call("rec_handle", h2), x) <= 20 * TOL[dt]
    gw.call("release_all", nlhs=0)
