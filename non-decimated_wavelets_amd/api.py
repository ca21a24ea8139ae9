"""Host-side mirror of the reference's class API for the NDWT hot path.

Reference: Functions/nd_dwt_1D.m, nd_dwt_2D.m, nd_dwt_3D.m, nd_dwt_4D.m -- constructor
`nd_dwt_3D(wname, sizes, 'pres_l2_norm', b, 'compute', c, 'precision', p)`, `dec(x, level)`, `rec(y)`
and the public properties (nd_dwt_3D.m:67-75).  Everything is computed by the HIP library behind
include/ndwt.h; PyTorch only provides device memory and the current stream.

Shapes follow MATLAB: x is `sizes` = [n1, ..., nd]; coefficients are [n1, ..., nd, bands] with the
band order of the reference (nd_dwt_3D.m:45-52).  Memory is column-major like MATLAB's: the tensors
returned are permuted views of contiguous (bands, nd, ..., n1) buffers, and column-major inputs are
consumed without a copy.

`compute` values: 'hip' -- torch tensors on the GPU in and out (the analogue of the reference's
'gpu'); 'hip_off' -- host arrays (numpy / CPU tensors) in and out, staged through the GPU (the
analogue of 'gpu_off').  The reference's own names are accepted as aliases for the data placement
they imply ('gpu' -> 'hip'; 'gpu_off', 'mat', 'mex' -> 'hip_off'); the arithmetic is always the HIP
engine's.  There is no CPU compute path.
"""
from __future__ import annotations

import ctypes
import warnings

import numpy as np
import torch

from . import _lib as L

_ORD = ["First", "Second", "Third", "Fourth"]


class Plan:
    """Thin RAII wrapper over ndwt_plan (include/ndwt.h)."""

    def __init__(self, dims, wnames, dtype, complex_interleaved=False, pres_l2_norm=False, dilation="reference",
                 max_level=1, device=0, global_outer=None):
        """global_outer: length of the outermost axis of the WHOLE volume when `dims` describe one slab of it (multi-GPU):
        the reference's filter-length check then applies to the whole axis, a slab may be thinner than the filter"""
        self.dims = [int(d) for d in dims]
        self.ndim = len(self.dims)
        self.wnames = list(wnames)
        self.dtype = dtype
        self.max_level = int(max_level)
        self.device = int(device)
        self._h = ctypes.c_void_p(None)
        dims_c = (ctypes.c_int64 * self.ndim)(*self.dims)
        names_c = (ctypes.c_char_p * self.ndim)(*[w.encode() for w in self.wnames])
        dt = L.NDWT_F32 if dtype in (torch.float32, np.float32, "single") else L.NDWT_F64
        dil = {"reference": L.NDWT_DILATION_REFERENCE, "atrous": L.NDWT_DILATION_ATROUS}[dilation]
        cplx = L.NDWT_COMPLEX_INTERLEAVED if complex_interleaved else L.NDWT_REAL
        if global_outer is None:
            L.check(L.lib().ndwt_plan_create(ctypes.byref(self._h), self.ndim, dims_c, names_c, dt, cplx,
                                             int(bool(pres_l2_norm)), dil, self.max_level, self.device))
        else:
            L.check(L.lib().ndwt_plan_create_slab(ctypes.byref(self._h), self.ndim, dims_c, int(global_outer), names_c, dt, cplx,
                                                  int(bool(pres_l2_norm)), dil, self.max_level, self.device))

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                L.lib().ndwt_plan_destroy(self._h)
                self._h = ctypes.c_void_p(None)
        except Exception:
            pass

    def set_path(self, generic: bool):
        L.check(L.lib().ndwt_plan_set_path(self._h, L.NDWT_PATH_GENERIC if generic else L.NDWT_PATH_AUTO))

    def set_tuning(self, target_blocks=0, force_zchunk=0):
        L.check(L.lib().ndwt_plan_set_tuning(self._h, int(target_blocks), int(force_zchunk)))

    def set_variant(self, fwd=-1, inv=-1, zchunk_fwd=-1, zchunk_inv=-1, fp64_fused=-1):
        """tuning hook of tools/ (A/B runs of kernel variants; same values): negative = unchanged"""
        L.check(L.lib().ndwt_plan_set_variant(self._h, int(fwd), int(inv), int(zchunk_fwd), int(zchunk_inv), int(fp64_fused)))
        return self

    def set_fused_level1(self, mode):
        """tuning hook: 0 / False = denoise() keeps the level-1 detail bands in memory; 1 / True (default) = level 1 in one launch where
        that is faster (tap lengths <= 6); 2 = wherever the kernel exists (8 taps too)"""
        L.check(L.lib().ndwt_plan_set_fused_level1(self._h, int(mode)))
        return self

    def set_variant_from_env(self):
        """tools/ only: NDWT_VARIANT_FWD / NDWT_VARIANT_INV / NDWT_ZCHUNK_FWD / NDWT_ZCHUNK_INV / NDWT_FP64_FUSED of the caller's
        environment, applied through set_variant (the library itself never reads the environment)"""
        import os
        g = lambda k: int(os.environ[k]) if os.environ.get(k) not in (None, "") else -1
        return self.set_variant(g("NDWT_VARIANT_FWD"), g("NDWT_VARIANT_INV"), g("NDWT_ZCHUNK_FWD"), g("NDWT_ZCHUNK_INV"), g("NDWT_FP64_FUSED"))

    def set_profiling(self, on: bool):
        L.check(L.lib().ndwt_plan_set_profiling(self._h, int(bool(on))))

    def get_profile(self, kind: int):
        """(total_ms, launches) of kernel kind 0 fused analysis, 1 fused synthesis, 2 axis analysis, 3 axis synthesis"""
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        L.check(L.lib().ndwt_plan_get_profile(self._h, int(kind), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def describe(self) -> str:
        buf = ctypes.create_string_buffer(64)
        L.check(L.lib().ndwt_plan_describe(self._h, buf, 64))
        return buf.value.decode()

    def band_pitch(self) -> int:
        """recommended band pitch of a pitched coefficient buffer, in elements (include/ndwt.h: ndwt_band_pitch)"""
        return int(L.lib().ndwt_band_pitch(self._h))

    def dec(self, x_ptr, y_ptr, level, stream=0, band_pitch=0):
        """band_pitch: elements between consecutive bands of y (0 = packed, the reference layout)"""
        L.check(L.lib().ndwt_dec_pitched(self._h, x_ptr, y_ptr, int(band_pitch), int(level), ctypes.c_void_p(stream)))

    def rec(self, y_ptr, x_ptr, level, stream=0, band_pitch=0):
        L.check(L.lib().ndwt_rec_pitched(self._h, y_ptr, int(band_pitch), x_ptr, int(level), ctypes.c_void_p(stream)))

    def shrink(self, y_ptr, level, threshold, hard=False, stream=0, band_pitch=0):
        """in-place soft (default) / hard thresholding of the detail bands"""
        L.check(L.lib().ndwt_shrink_pitched(self._h, y_ptr, int(band_pitch), int(level), float(threshold), int(bool(hard)), ctypes.c_void_p(stream)))

    def denoise(self, x_ptr, out_ptr, level, threshold, hard=False, stream=0):
        """dec -> shrink -> rec with the coefficients in a scratch array owned by the plan"""
        L.check(L.lib().ndwt_denoise(self._h, x_ptr, out_ptr, int(level), float(threshold), int(bool(hard)), ctypes.c_void_p(stream)))

    def denoise_host(self, x_ptr, out_ptr, level, threshold, hard=False):
        L.check(L.lib().ndwt_denoise_host(self._h, x_ptr, out_ptr, int(level), float(threshold), int(bool(hard))))

    def dec_split(self, x_re, x_im, y_re, y_im, level, stream=0):
        """split complex (separate re / im device arrays, the mxGetPr / mxGetPi layout); x_im, y_im may be None"""
        L.check(L.lib().ndwt_dec_split(self._h, x_re, x_im, y_re, y_im, int(level), ctypes.c_void_p(stream)))

    def rec_split(self, y_re, y_im, x_re, x_im, level, stream=0):
        L.check(L.lib().ndwt_rec_split(self._h, y_re, y_im, x_re, x_im, int(level), ctypes.c_void_p(stream)))

    def dec_split_host(self, x_re, x_im, y_re, y_im, level):
        L.check(L.lib().ndwt_dec_split_host(self._h, x_re, x_im, y_re, y_im, int(level)))

    def rec_split_host(self, y_re, y_im, x_re, x_im, level):
        L.check(L.lib().ndwt_rec_split_host(self._h, y_re, y_im, x_re, x_im, int(level)))

    def slab_segments(self, add, dsts, srcs, counts, stream=0):
        """up to 8 runs of elements copied (add=False) or added (dst += src) in one launch (include/ndwt.h: ndwt_slab_segments)"""
        n = len(dsts)
        L.check(L.lib().ndwt_slab_segments(self._h, 1 if add else 0, n, (ctypes.c_void_p * n)(*dsts), (ctypes.c_void_p * n)(*srcs),
                                           (ctypes.c_int64 * n)(*counts), ctypes.c_void_p(stream)))

    def release_staging(self):
        L.check(L.lib().ndwt_plan_release_staging(self._h))

    def slab_halo(self, stride=1):
        v = [ctypes.c_int64(0) for _ in range(4)]
        L.check(L.lib().ndwt_slab_halo(self._h, int(stride), *[ctypes.byref(t) for t in v]))
        return tuple(t.value for t in v)   # (ana_before, ana_after, syn_before, syn_after)

    def analysis_level_slab(self, in_ptr, out_ptrs, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(out_ptrs))(*out_ptrs)
        L.check(L.lib().ndwt_analysis_level_slab(self._h, in_ptr, arr, int(stride), ctypes.c_void_p(stream)))

    def analysis_level_slab_split(self, in_ptr, halo_before_ptr, halo_after_ptr, out_ptrs, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(out_ptrs))(*out_ptrs)
        L.check(L.lib().ndwt_analysis_level_slab_split(self._h, in_ptr, halo_before_ptr, halo_after_ptr, arr, int(stride),
                                                       ctypes.c_void_p(stream)))

    def synthesis_level_slab_ext(self, in_ptrs, out_ext_ptr, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(in_ptrs))(*in_ptrs)
        L.check(L.lib().ndwt_synthesis_level_slab_ext(self._h, arr, out_ext_ptr, int(stride), ctypes.c_void_p(stream)))

    def analysis_level_slab_part(self, in_ptr, halo_before_ptr, halo_after_ptr, out_ptrs, n_planes, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(out_ptrs))(*out_ptrs)
        L.check(L.lib().ndwt_analysis_level_slab_part(self._h, in_ptr, halo_before_ptr, halo_after_ptr, arr, int(stride),
                                                      int(n_planes), ctypes.c_void_p(stream)))

    def synthesis_level_slab_part(self, in_ptrs, n_in, e0, n_out, out_ptr, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(in_ptrs))(*in_ptrs)
        L.check(L.lib().ndwt_synthesis_level_slab_part(self._h, arr, int(n_in), int(e0), int(n_out), out_ptr, int(stride),
                                                       ctypes.c_void_p(stream)))

    def analysis_level_slab_runs(self, in_ptr, out_ptrs, n_planes, n_runs, run_stride, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(out_ptrs))(*out_ptrs)
        L.check(L.lib().ndwt_analysis_level_slab_runs(self._h, in_ptr, arr, int(stride), int(n_planes), int(n_runs), int(run_stride),
                                                      ctypes.c_void_p(stream)))

    def synthesis_level_slab_runs(self, in_ptrs, n_in, e0, e_stride, n_runs, n_out, out_ptr, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(in_ptrs))(*in_ptrs)
        L.check(L.lib().ndwt_synthesis_level_slab_runs(self._h, arr, int(n_in), int(e0), int(e_stride), int(n_runs), int(n_out),
                                                       out_ptr, int(stride), ctypes.c_void_p(stream)))

    def synthesis_level_slab(self, in_ptrs, out_ptr, stride=1, stream=0):
        arr = (ctypes.c_void_p * len(in_ptrs))(*in_ptrs)
        L.check(L.lib().ndwt_synthesis_level_slab(self._h, arr, out_ptr, int(stride), ctypes.c_void_p(stream)))


class Coefficients:
    """Device-resident coefficients behind an opaque handle (include/ndwt.h: ndwt_coef_*): what the MATLAB gateway's `dec_keep` /
    `rec_handle` commands hold between calls -- only the signal crosses PCIe (the reference's gateway moves the whole coefficient
    array through host memory per call, nd_dwt_3D.m:161,225).  Host arrays are numpy, kernel order ((bands,) nd, ..., n1)."""

    def __init__(self, plan: Plan, handle: ctypes.c_void_p):
        self.plan, self._h = plan, handle

    @classmethod
    def dec(cls, plan: Plan, x: np.ndarray, level: int, reuse: "Coefficients | None" = None):
        h = reuse._h if reuse is not None else ctypes.c_void_p(None)
        x = np.ascontiguousarray(x)
        L.check(L.lib().ndwt_coef_dec_host(plan._h, x.ctypes.data_as(ctypes.c_void_p), int(level), ctypes.byref(h)))
        return reuse if reuse is not None else cls(plan, h)

    @classmethod
    def put(cls, plan: Plan, y: np.ndarray, level: int):
        h = ctypes.c_void_p(None)
        y = np.ascontiguousarray(y)
        L.check(L.lib().ndwt_coef_put_host(plan._h, int(level), y.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)))
        return cls(plan, h)

    def info(self):
        lev, nb, pitch, ptr = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_void_p(None)
        L.check(L.lib().ndwt_coef_info(self._h, ctypes.byref(lev), ctypes.byref(nb), ctypes.byref(pitch), ctypes.byref(ptr)))
        return {"level": lev.value, "bands": nb.value, "band_pitch": pitch.value, "dev_ptr": ptr.value}

    def rec(self, out: np.ndarray):
        assert out.flags.c_contiguous
        L.check(L.lib().ndwt_coef_rec_host(self.plan._h, self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def shrink(self, threshold, hard=False):
        L.check(L.lib().ndwt_coef_shrink(self.plan._h, self._h, float(threshold), int(bool(hard))))
        return self

    def get(self, out: np.ndarray):
        assert out.flags.c_contiguous
        L.check(L.lib().ndwt_coef_get_host(self.plan._h, self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def release(self):
        if self._h is not None and self._h.value:
            L.lib().ndwt_coef_release(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class MultiPlan:
    """Single-process multi-device plan (include/ndwt.h, ndwt_mplan_*): ONE host thread drives the listed devices; the volume is
    sharded in slabs on its outermost axis (a device listed twice holds two slabs).  Host arrays in kernel order
    ((bands,) nd, ..., n1, contiguous) in and out -- the path behind the MATLAB gateway, where the host is one process."""

    def __init__(self, dims, wnames, dtype, devices, complex_interleaved=False, pres_l2_norm=False, dilation="reference", max_level=3):
        self.dims = [int(d) for d in dims]
        self.ndim = len(self.dims)
        self.np_dtype = np.float32 if dtype in (torch.float32, np.float32, "single") else np.float64
        self.complex = bool(complex_interleaved)
        self.max_level = int(max_level)
        self._h = ctypes.c_void_p(None)
        dims_c = (ctypes.c_int64 * self.ndim)(*self.dims)
        names_c = (ctypes.c_char_p * self.ndim)(*[w.encode() for w in wnames])
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        dt = L.NDWT_F32 if self.np_dtype == np.float32 else L.NDWT_F64
        dil = {"reference": L.NDWT_DILATION_REFERENCE, "atrous": L.NDWT_DILATION_ATROUS}[dilation]
        L.mcheck(L.lib().ndwt_mplan_create(ctypes.byref(self._h), self.ndim, dims_c, names_c, dt,
                                           L.NDWT_COMPLEX_INTERLEAVED if self.complex else L.NDWT_REAL, int(bool(pres_l2_norm)), dil,
                                           self.max_level, devs, len(devices)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                L.lib().ndwt_mplan_destroy(self._h)
                self._h = ctypes.c_void_p(None)
        except Exception:
            pass

    def slabs(self):
        out = []
        for i in range(L.lib().ndwt_mplan_num_slabs(self._h)):
            d, z0, n = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
            L.mcheck(L.lib().ndwt_mplan_slab(self._h, i, ctypes.byref(d), ctypes.byref(z0), ctypes.byref(n)))
            out.append((d.value, z0.value, n.value))
        return out

    def _check(self, a, bands):
        cdt = {np.float32: np.complex64, np.float64: np.complex128}[self.np_dtype] if self.complex else self.np_dtype
        shape = ((bands,) if bands else ()) + tuple(reversed(self.dims))
        if not (isinstance(a, np.ndarray) and a.dtype == cdt and a.flags.c_contiguous and a.shape == shape):
            raise ValueError(f"expected a C-contiguous {np.dtype(cdt).name} array of shape {shape}")

    def dec(self, x, level):
        """x: numpy (nd, ..., n1) -> numpy (bands, nd, ..., n1)"""
        self._check(x, 0)
        y = np.empty((num_bands(self.ndim, level),) + x.shape, dtype=x.dtype)
        L.mcheck(L.lib().ndwt_mdec_host(self._h, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), int(level)))
        return y

    def rec(self, y):
        level = L.lib().ndwt_level_from_bands(self.ndim, int(y.shape[0]))
        if level < 1:
            raise ValueError(f"{y.shape[0]} bands is not a valid {self.ndim}-D coefficient count")
        self._check(y, y.shape[0])
        x = np.empty(y.shape[1:], dtype=y.dtype)
        L.mcheck(L.lib().ndwt_mrec_host(self._h, y.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p), level))
        return x

    def set_exchange(self, scheme):
        """'scatter' (default: one band of partial sums per level; equal to one device to rounding) or 'gather' (bit-identical)"""
        L.mcheck(L.lib().ndwt_mplan_set_exchange(self._h, {"scatter": 0, "gather": 1}[scheme]))
        return self

    def set_overlap(self, on):
        """True (default): copies between slabs on their own streams, overlapped with the launches that do not wait for them"""
        L.mcheck(L.lib().ndwt_mplan_set_overlap(self._h, int(on)))
        return self

    def set_threads(self, on):
        """True (default): one host thread per slab queues that slab's work; False: the calling thread queues everything.  Same results."""
        L.mcheck(L.lib().ndwt_mplan_set_threads(self._h, int(bool(on))))
        return self

    def describe(self) -> str:
        buf = ctypes.create_string_buffer(512)
        L.mcheck(L.lib().ndwt_mplan_describe(self._h, buf, 512))
        return buf.value.decode()

    def _slab_tensors(self, tensors, bands):
        sl = self.slabs()
        tdt = {np.float32: torch.float32, np.float64: torch.float64}[self.np_dtype]
        if self.complex:
            tdt = {torch.float32: torch.complex64, torch.float64: torch.complex128}[tdt]
        if len(tensors) != len(sl):
            raise ValueError(f"{len(sl)} slab tensors expected")
        for t, (dev, _, n) in zip(tensors, sl):
            shape = ((bands,) if bands else ()) + (n,) + tuple(reversed(self.dims[:-1]))
            if not (t.is_cuda and t.device.index == dev and t.dtype == tdt and t.is_contiguous() and tuple(t.shape) == shape):
                raise ValueError(f"slab tensor: contiguous {tdt} of shape {shape} on cuda:{dev} expected")
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def dec_device(self, x_slabs, level):
        """device-resident form: x_slabs[i] = torch tensor (n_i, ..., n1) on slab i's device -> list of (bands, n_i, ..., n1) tensors.
        Ordered against torch by host synchronisation: every slab device is synchronised before the call; returns when the result is."""
        xs = self._slab_tensors(x_slabs, 0)
        nbt = num_bands(self.ndim, level)
        ys = [torch.empty((nbt,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in x_slabs]
        self._sync(x_slabs)
        L.mcheck(L.lib().ndwt_mdec(self._h, xs, self._slab_tensors(ys, nbt), int(level)))
        return ys

    @staticmethod
    def _sync(tensors):
        """The plan's private streams are ordered against torch by HOST synchronisation only: the inputs must be complete, and the output
        tensors just allocated may be blocks the caching allocator recycled from tensors whose kernels are still queued on a torch
        stream -- every slab device is synchronised before the plan's streams touch them."""
        for dev in sorted({t.device.index for t in tensors}):
            torch.cuda.synchronize(dev)

    def rec_device(self, y_slabs):
        nbt = int(y_slabs[0].shape[0])
        level = L.lib().ndwt_level_from_bands(self.ndim, nbt)
        if level < 1:
            raise ValueError(f"{nbt} bands is not a valid {self.ndim}-D coefficient count")
        ys = self._slab_tensors(y_slabs, nbt)
        xs = [torch.empty(tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in y_slabs]
        self._sync(y_slabs)
        L.mcheck(L.lib().ndwt_mrec(self._h, ys, self._slab_tensors(xs, 0), level))
        return xs


def num_bands(ndim, level):
    return int(L.lib().ndwt_num_bands(int(ndim), int(level)))


def _current_stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _NdDwtBase:
    """Common implementation of nd_dwt_{1,2,3,4}D (reference files cited per method)."""

    NDIM = 0

    def __init__(self, wname, sizes, *varargin, **kwargs):
        d = self.NDIM
        sizes = [int(s) for s in np.atleast_1d(sizes)]
        self._check_sizes(sizes)
        self.sizes = sizes
        self.wname = self._check_wname(wname)
        # defaults: nd_dwt_3D.m:101-103 (compute defaults to the HIP engine here)
        self.pres_l2_norm = 0
        self.precision = "double"
        self.compute = "hip"
        self.dilation = "reference"
        self.device = None
        self.devices = None
        self.band_pitch = "packed"
        # name/value pairs as in MATLAB (nd_dwt_3D.m:105-120); keyword arguments are accepted too
        if len(varargin) % 2:
            raise ValueError("Optional inputs must come in pairs")
        opts = [(varargin[i], varargin[i + 1]) for i in range(0, len(varargin), 2)] + list(kwargs.items())
        for ind, (key, val) in enumerate(opts):
            k = str(key).lower()
            if k == "pres_l2_norm":
                self.pres_l2_norm = int(bool(val))
            elif k == "compute":
                self.compute = str(val)
            elif k == "precision":
                self.precision = str(val)
            elif k == "dilation":
                self.dilation = str(val).lower()
            elif k == "device":
                self.device = val
            elif k == "band_pitch":                # 'packed' (reference layout) | 'auto' | elements between bands of dec()'s result
                self.band_pitch = val if isinstance(val, str) else int(val)
            elif k == "devices":                   # shard the outermost axis over these devices (host arrays, one process)
                self.devices = [int(v) for v in np.atleast_1d(val)]
            else:   # unknown keys only warn (nd_dwt_3D.m:118)
                warnings.warn(f"Unknown optional input #{2 * ind + 1} ingoring!")
        if self.compute.lower() == "mex" and self.precision.lower() == "single":   # nd_dwt_3D.m:122-124
            raise ValueError("Single precsision is not currently supported for mex computation")
        c = self.compute.lower()
        if c in ("hip", "gpu"):
            self._offload = False
        elif c in ("hip_off", "gpu_off", "mat", "mex"):
            self._offload = True
        else:
            raise ValueError(f"unknown compute '{self.compute}'")
        if self.precision.lower() not in ("double", "single"):
            raise ValueError("precision must be 'double' or 'single'")
        if self.dilation not in ("reference", "atrous"):
            raise ValueError("dilation must be 'reference' or 'atrous'")
        if isinstance(self.band_pitch, str) and self.band_pitch.lower() not in ("packed", "auto"):
            raise ValueError("band_pitch must be 'packed', 'auto' or a number of elements")
        if self.band_pitch != "packed" and self._offload:
            raise ValueError("'band_pitch' lays out device tensors: use compute='hip'")
        if self.devices is not None and (not self._offload or d < 2):
            raise ValueError("'devices' shards host arrays of 2-D .. 4-D transforms: use compute='hip_off'")
        # get_filters (nd_dwt_3D.m:263-342): per-axis taps instead of N-D FFT-domain kernels
        self.f_dec = [L.wave_filters(w) for w in self.wname[:d]]
        self.f_size = {f"s{a + 1}": len(self.f_dec[a][0]) for a in range(d)}
        for a in range(d):
            if self.f_size[f"s{a + 1}"] > sizes[a]:   # nd_dwt_3D.m:277-286
                raise ValueError(f"{_ORD[a]} Dimension of Data is shorter than the wavelet filter being used")
        self._plans = {}

    # -- per-dimension validation (messages of the reference) --
    def _check_sizes(self, sizes):
        raise NotImplementedError

    def _check_wname(self, wname):
        raise NotImplementedError

    def _level_from_bands(self, n):
        raise NotImplementedError

    # -- plumbing --
    def _torch_dtype(self, is_complex):
        single = self.precision.lower() == "single"
        if is_complex:
            return torch.complex64 if single else torch.complex128
        return torch.float32 if single else torch.float64

    def _dev(self, x=None):
        if self.device is not None:
            return torch.device(self.device)
        if x is not None and isinstance(x, torch.Tensor) and x.is_cuda:
            return x.device
        return torch.device("cuda", torch.cuda.current_device())

    def _plan(self, is_complex, level, dev):
        # ONE plan per (data kind, device): a plan owns scratch (GBs for large volumes) and serves one stream at a time
        # (include/ndwt.h).  A call on another torch stream than the plan's previous one first makes the new stream wait for
        # everything queued on the old one, so the scratch is never shared by two streams in flight and nothing accumulates per stream.
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        key = (is_complex, idx)
        p = self._plans.get(key)
        if p is None or p.max_level < level:
            real_dt = torch.float32 if self.precision.lower() == "single" else torch.float64
            if p is not None:
                torch.cuda.synchronize(idx)                           # the plan being replaced may still be running
            p = Plan(self.sizes, self.wname[: self.NDIM], real_dt, is_complex, self.pres_l2_norm, self.dilation,
                     max_level=max(level, 3), device=idx)
            p._last_stream = None
            self._plans[key] = p
        cur = torch.cuda.current_stream(torch.device("cuda", idx))
        if p._last_stream is not None and p._last_stream != cur:
            cur.wait_event(p._last_stream.record_event())
        p._last_stream = cur
        return p

    def _to_device_kernel_order(self, x, dev, ndim_expected):
        """user array (MATLAB shape) -> contiguous tensor in kernel order (reversed dims) on `dev`."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(np.transpose(x)))     # kernel order on the host
            kernel_order = True
        else:
            kernel_order = False
        if not isinstance(x, torch.Tensor):
            raise TypeError("x must be a torch.Tensor or numpy.ndarray")
        if not self._offload and not x.is_cuda:
            raise ValueError("compute='hip' expects tensors on the GPU (use compute='hip_off' for host arrays)")
        if not kernel_order:
            x = x.permute(*reversed(range(x.dim())))
        dt = self._torch_dtype(x.is_complex())
        return x.to(device=dev, dtype=dt).contiguous()

    def _from_device(self, t_kernel, like_numpy, dev_in):
        out = t_kernel.permute(*reversed(range(t_kernel.dim())))            # MATLAB shape, column-major memory
        if self._offload:
            out = out.cpu()
            if like_numpy:
                return out.numpy()
        return out

    # -- nd_dwt_3D.m:142-199 --
    def dec(self, x, level):
        level = int(level)
        if level < 1:
            raise ValueError("level must be >= 1")
        like_numpy = isinstance(x, np.ndarray)
        x = self._prep_dec_input(x)
        if list(x.shape) != self.sizes:
            raise ValueError(f"input size {list(x.shape)} does not match the object's sizes {self.sizes}")
        if self.devices is not None:
            return self._multi(x, level, None)
        dev = self._dev(x if isinstance(x, torch.Tensor) else None)
        xk = self._to_device_kernel_order(x, dev, self.NDIM)
        is_c = xk.is_complex()
        plan = self._plan(is_c, level, dev)
        nb = num_bands(self.NDIM, level)
        vol = int(np.prod(xk.shape))
        pitch = self._pitch(plan, vol)
        if pitch:
            # a pitched coefficient tensor: same shape and indexing, band b at b * pitch elements (a view of one allocation);
            # .contiguous() gives the packed reference layout
            flat = torch.empty(nb * pitch, dtype=xk.dtype, device=dev)
            cs = [1] * xk.dim()
            for i in range(xk.dim() - 2, -1, -1):
                cs[i] = cs[i + 1] * int(xk.shape[i + 1])
            yk = flat.as_strided((nb,) + tuple(xk.shape), (pitch,) + tuple(cs))
        else:
            yk = torch.empty((nb,) + tuple(xk.shape), dtype=xk.dtype, device=dev)
        with torch.cuda.device(dev):
            plan.dec(xk.data_ptr(), yk.data_ptr(), level, _current_stream(dev), band_pitch=pitch)
        return self._from_device(yk, like_numpy, dev)   # real in -> real out (nd_dwt_3D.m:189-192) by construction

    def _pitch(self, plan, vol):
        if self.band_pitch == "packed":
            return 0
        if isinstance(self.band_pitch, str):
            return plan.band_pitch()
        if self.band_pitch < vol:
            raise ValueError(f"band_pitch {self.band_pitch} is smaller than a band ({vol} elements)")
        return 0 if self.band_pitch == vol else int(self.band_pitch)

    def _coef_kernel_order(self, y, dev):
        """coefficient array (MATLAB shape [dims, bands]) -> (tensor in kernel order, band pitch): a device tensor whose bands are
        each contiguous and evenly spaced (what dec() returns, packed or pitched) is used where it lies"""
        if isinstance(y, torch.Tensor) and y.is_cuda and y.device == dev and y.dtype == self._torch_dtype(y.is_complex()):
            yk = y.permute(*reversed(range(y.dim())))
            if yk[0].is_contiguous() and yk.shape[0] > 1 and yk.stride(0) > yk[0].numel():
                return yk, int(yk.stride(0))
        return self._to_device_kernel_order(y, dev, self.NDIM + 1), 0

    # -- nd_dwt_3D.m:202-256 --
    def rec(self, y):
        like_numpy = isinstance(y, np.ndarray)
        if y.ndim != self.NDIM + 1 or list(y.shape[:-1]) != self.sizes:
            raise ValueError(f"coefficient array must have shape {self.sizes + ['bands']}")
        level = self._level_from_bands(int(y.shape[-1]))
        if num_bands(self.NDIM, level) != int(y.shape[-1]):
            raise ValueError(f"{int(y.shape[-1])} bands is not a valid {self.NDIM}-D coefficient count")
        if self.devices is not None:
            return self._multi(y, level, "rec")
        dev = self._dev(y if isinstance(y, torch.Tensor) else None)
        yk, pitch = self._coef_kernel_order(y, dev)
        plan = self._plan(yk.is_complex(), level, dev)
        xk = torch.empty(tuple(yk.shape[1:]), dtype=yk.dtype, device=dev)
        with torch.cuda.device(dev):
            plan.rec(yk.data_ptr(), xk.data_ptr(), level, _current_stream(dev), band_pitch=pitch)
        return self._from_device(xk, like_numpy, dev)

    # -- consumers for iterative solvers (extension; not in the reference) --
    def shrink(self, y, threshold, mode="soft"):
        """Soft / hard thresholding of every detail band of a coefficient array (band 0, the coarsest approximation,
        is kept); complex data: the magnitude is shrunk.  Returns a new array of the same kind as `y`."""
        if mode not in ("soft", "hard"):
            raise ValueError("mode must be 'soft' or 'hard'")
        like_numpy = isinstance(y, np.ndarray)
        if y.ndim != self.NDIM + 1 or list(y.shape[:-1]) != self.sizes:
            raise ValueError(f"coefficient array must have shape {self.sizes + ['bands']}")
        level = self._level_from_bands(int(y.shape[-1]))
        dev = self._dev(y if isinstance(y, torch.Tensor) else None)
        yk, pitch = self._coef_kernel_order(y, dev)
        if isinstance(y, torch.Tensor) and yk.data_ptr() == y.data_ptr():
            if pitch:                                          # never modify the caller's array: copy, keeping the pitch
                flat = torch.empty(yk.shape[0] * pitch, dtype=yk.dtype, device=dev)
                cp = flat.as_strided(tuple(yk.shape), tuple(yk.stride()))
                cp.copy_(yk)
                yk = cp
            else:
                yk = yk.clone()
        plan = self._plan(yk.is_complex(), level, dev)
        with torch.cuda.device(dev):
            plan.shrink(yk.data_ptr(), level, threshold, mode == "hard", _current_stream(dev), band_pitch=pitch)
        return self._from_device(yk, like_numpy, dev)

    def denoise(self, x, level, threshold, mode="soft"):
        """rec(shrink(dec(x, level), threshold)) in one call: the coefficients stay in a scratch array of the plan."""
        if mode not in ("soft", "hard"):
            raise ValueError("mode must be 'soft' or 'hard'")
        level = int(level)
        if level < 1:
            raise ValueError("level must be >= 1")
        like_numpy = isinstance(x, np.ndarray)
        x = self._prep_dec_input(x)
        if list(x.shape) != self.sizes:
            raise ValueError(f"input size {list(x.shape)} does not match the object's sizes {self.sizes}")
        dev = self._dev(x if isinstance(x, torch.Tensor) else None)
        xk = self._to_device_kernel_order(x, dev, self.NDIM)
        plan = self._plan(xk.is_complex(), level, dev)
        out = torch.empty_like(xk)
        with torch.cuda.device(dev):
            plan.denoise(xk.data_ptr(), out.data_ptr(), level, threshold, mode == "hard", _current_stream(dev))
        return self._from_device(out, like_numpy, dev)

    def _prep_dec_input(self, x):
        return x

    def _multi(self, a, level, direction):
        """host array through the single-process multi-device plan (the 'devices' option)"""
        like_numpy = isinstance(a, np.ndarray)
        an = a if like_numpy else a.cpu().numpy()
        cplx = np.iscomplexobj(an)
        single = self.precision.lower() == "single"
        cdt = (np.complex64 if single else np.complex128) if cplx else (np.float32 if single else np.float64)
        ak = np.ascontiguousarray(np.transpose(an)).astype(cdt, copy=False)
        key = ("multi", cplx)
        mp = self._plans.get(key)
        if mp is None or mp.max_level < level:
            mp = MultiPlan(self.sizes, self.wname[: self.NDIM], torch.float32 if single else torch.float64, self.devices, cplx,
                           self.pres_l2_norm, self.dilation, max_level=max(level, 3))
            self._plans[key] = mp
        out = np.transpose(mp.rec(ak) if direction == "rec" else mp.dec(ak, level))
        return out if like_numpy else torch.from_numpy(np.ascontiguousarray(out))


class nd_dwt_1D(_NdDwtBase):
    """Functions/nd_dwt_1D.m -- 1-D signal of length n; coefficients [n, 1+level]."""
    NDIM = 1

    def _check_sizes(self, sizes):
        if len(sizes) != 1:
            raise ValueError("1D array length must be a scalar")          # nd_dwt_1D.m:88

    def _check_wname(self, wname):
        if not isinstance(wname, str):
            raise ValueError("Wavelet Name Must be a string")             # nd_dwt_1D.m:84
        return [wname, wname]

    def _level_from_bands(self, n):
        return int(np.ceil(n - 1))                                        # nd_dwt_1D.m:213

    def _prep_dec_input(self, x):
        # row vectors are transposed (nd_dwt_1D.m:151-153); accept [n], [n,1] and [1,n]
        if x.ndim == 2 and 1 in x.shape:
            x = x.reshape(-1)
        return x


class nd_dwt_2D(_NdDwtBase):
    """Functions/nd_dwt_2D.m -- coefficients [n1, n2, 4+3(level-1)]."""
    NDIM = 2

    def _check_sizes(self, sizes):
        if len(sizes) != 2:
            raise ValueError("The sizes vector must be length 2")

    def _check_wname(self, wname):
        if isinstance(wname, str):
            return [wname, wname]
        if len(wname) != 2:
            raise ValueError("You must specify two filter names in a cell array of length 2, or a single string for the "
                             "same filter to be used in all dimensions")
        return list(wname)

    def _level_from_bands(self, n):
        return int(1 + (n - 4) / 3)                                       # nd_dwt_2D.m:215


class nd_dwt_3D(_NdDwtBase):
    """Functions/nd_dwt_3D.m -- coefficients [n1, n2, n3, 8+7(level-1)]."""
    NDIM = 3

    def _check_sizes(self, sizes):
        if len(sizes) != 3:
            raise ValueError("The sizes vector must be length 3")          # nd_dwt_3D.m:83

    def _check_wname(self, wname):
        if isinstance(wname, str):
            return [wname, wname, wname]
        if len(wname) != 3:                                               # nd_dwt_3D.m:94-96
            raise ValueError("You must specify three filter names in a cell arrayof length 3, or a single string for "
                             "the same filter to be used in all dimensions")
        return list(wname)

    def _level_from_bands(self, n):
        return int(np.ceil(n / 8))                                        # nd_dwt_3D.m:217 (breaks for level >= 9, as there)


class nd_dwt_4D(_NdDwtBase):
    """Functions/nd_dwt_4D.m ('fft' method semantics) -- coefficients [n1, n2, n3, n4, 16+15(level-1)]."""
    NDIM = 4

    def _check_sizes(self, sizes):
        if len(sizes) != 4:
            raise ValueError("The sizes vector must be length 4")

    def _check_wname(self, wname):
        if isinstance(wname, str):
            return [wname] * 4
        if len(wname) != 4:
            raise ValueError("You must specify four filter names in a cell array of length 4, or a single string for "
                             "the same filter to be used in all dimensions")
        return list(wname)

    def _level_from_bands(self, n):
        return int(1 + (n - 16) / 15)                                     # nd_dwt_4D.m:213
