"""ctypes front end of oracle/ndwt_spatial.c (CPU oracle #2, C / OpenMP) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this.  Same array conventions as ndwt_oracle.py: x has the MATLAB
shape `sizes` = [n1, .., nd] (numpy axis k takes wavelet k), coefficients have shape [n1, .., nd, bands].  `dec_planar` / `rec_planar` keep
the C library's band-planar layout [bands, n1, .., nd] (no transposed copy: what the baseline timing uses).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

try:
    from .ndwt_oracle import level_from_bands, num_bands, wave_filters
except ImportError:                                    # imported as a plain module (tests put oracle/ on sys.path)
    from ndwt_oracle import level_from_bands, num_bands, wave_filters

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(native: bool = False, out_dir: str | None = None) -> str:
    """Compile ndwt_spatial.c (gcc); returns the path of the shared library.  native=True: -march=native into `out_dir` (bench timing)."""
    src = os.path.join(_HERE, "ndwt_spatial.c")
    if not native:
        subprocess.run(["make", "-s", "-C", _HERE], check=True)
        return os.path.join(_HERE, "_build", "libndwt_spatial.so")
    out = os.path.join(out_dir or os.path.join(_HERE, "_build"), "libndwt_spatial_native.so")
    subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", src, "-o", out, "-lm"], check=True)
    return out


def load(path: str | None = None):
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    p = path or os.path.join(_HERE, "_build", "libndwt_spatial.so")
    if not os.path.exists(p):
        p = build()
    L = ctypes.CDLL(p)
    vp, ip, i = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int
    for suf in ("f32", "f64"):
        for d in ("dec", "rec"):
            f = getattr(L, f"ndwt_c_{d}_{suf}")
            f.restype = i
            f.argtypes = [vp, vp, i, ctypes.POINTER(ctypes.c_longlong), i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ip,
                          i, i, i, i]
    L.ndwt_c_max_threads.restype = i
    L.ndwt_c_set_threads.argtypes = [i]
    if path is None:
        _LIB = L
    return L


def _taps(wname, d):
    if isinstance(wname, str):
        wname = [wname] * d
    filt = [wave_filters(w) for w in wname]
    maxlen = max(len(f[0]) for f in filt)
    lo = np.zeros((d, maxlen))
    hi = np.zeros((d, maxlen))
    for k, (l, h) in enumerate(filt):
        lo[k, :len(l)] = l
        hi[k, :len(h)] = h
    return lo, hi, np.array([len(f[0]) for f in filt], dtype=np.int32), maxlen


def _call(name, a, b, shape, ncomp, wname, level, l2, dilation, lib):
    L = lib or load()
    d = len(shape)
    lo, hi, ln, maxlen = _taps(wname, d)
    shp = (ctypes.c_longlong * d)(*shape)
    f = getattr(L, f"ndwt_c_{name}_{'f32' if a.dtype == np.float32 else 'f64'}")
    rc = f(a.ctypes.data, b.ctypes.data, d, shp, ncomp, lo.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
           hi.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ln.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), maxlen, level, int(bool(l2)),
           int(dilation == "atrous"))
    if rc:
        raise RuntimeError(f"ndwt_c_{name}: status {rc}")


def _as_real(x):
    x = np.ascontiguousarray(x)
    if np.iscomplexobj(x):
        return x.view(x.real.dtype).reshape(x.shape + (2,)), 2
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    return x, 1


def dec_planar(x, wname, level, pres_l2_norm=0, dilation="reference", out=None, lib=None):
    """x [n1, .., nd] (real or complex, float32 / float64) -> coefficients [bands, n1, .., nd]."""
    shape = tuple(x.shape)
    xr, ncomp = _as_real(x)
    nb = num_bands(len(shape), level)
    y = out if out is not None else np.empty((nb,) + xr.shape, dtype=xr.dtype)
    _call("dec", xr, y, shape, ncomp, wname, level, pres_l2_norm, dilation, lib)
    if out is not None:
        return out
    return y.view(x.dtype).reshape((nb,) + shape) if ncomp == 2 else y


def rec_planar(y, wname, pres_l2_norm=0, dilation="reference", level=None, out=None, lib=None):
    """coefficients [bands, n1, .., nd] -> x [n1, .., nd]."""
    shape = tuple(y.shape[1:])
    level = level or level_from_bands(len(shape), y.shape[0])
    yr, ncomp = _as_real(y)
    x = out if out is not None else np.empty(yr.shape[1:], dtype=yr.dtype)
    _call("rec", yr, x, shape, ncomp, wname, level, pres_l2_norm, dilation, lib)
    if out is not None:
        return out
    return x.view(y.dtype).reshape(shape) if ncomp == 2 else x


def spatial_dec(x, wname, level, pres_l2_norm=0, dilation="reference"):
    """The signature and layout of ndwt_oracle.spatial_dec: [n1, .., nd] -> [n1, .., nd, bands]."""
    return np.moveaxis(dec_planar(np.asarray(x), wname, level, pres_l2_norm, dilation), 0, -1)


def spatial_rec(c, wname, pres_l2_norm=0, dilation="reference", level=None):
    return rec_planar(np.moveaxis(np.asarray(c), -1, 0), wname, pres_l2_norm, dilation, level)
