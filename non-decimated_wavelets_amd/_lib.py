"""ctypes binding of libndwt_hip.so (the C ABI declared in include/ndwt.h).

The library is built in-tree by `make -C csrc` (hipcc, gfx950).  There is no CPU fallback: if the
shared object is missing, or no HIP device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NDWT_LIB_VARIANT=<name> loads libndwt_hip_<name>.so (diagnostic / tuning builds of tools/, `make VARIANT=<name>`)
LIB_PATH = os.path.join(_HERE, "libndwt_hip" + ("_" + os.environ["NDWT_LIB_VARIANT"] if os.environ.get("NDWT_LIB_VARIANT") else "") + ".so")

# status codes / enums of include/ndwt.h
NDWT_OK = 0
NDWT_F32, NDWT_F64 = 0, 1
NDWT_REAL, NDWT_COMPLEX_INTERLEAVED = 0, 1
NDWT_DILATION_REFERENCE, NDWT_DILATION_ATROUS = 0, 1
NDWT_PATH_AUTO, NDWT_PATH_GENERIC = 0, 1

EXPORTS = [
    "ndwt_wave_filters", "ndwt_num_bands", "ndwt_level_from_bands", "ndwt_plan_create", "ndwt_plan_create_slab", "ndwt_plan_destroy",
    "ndwt_plan_set_path", "ndwt_plan_describe", "ndwt_plan_set_tuning", "ndwt_plan_set_variant", "ndwt_plan_set_fused_level1", "ndwt_plan_set_profiling", "ndwt_plan_get_profile", "ndwt_dec", "ndwt_rec", "ndwt_dec_host",
    "ndwt_rec_host", "ndwt_shrink", "ndwt_denoise", "ndwt_denoise_host", "ndwt_dec_split", "ndwt_rec_split", "ndwt_dec_split_host", "ndwt_rec_split_host", "ndwt_slab_halo", "ndwt_analysis_level_slab", "ndwt_synthesis_level_slab",
    "ndwt_analysis_level_slab_split", "ndwt_synthesis_level_slab_ext", "ndwt_analysis_level_slab_part",
    "ndwt_synthesis_level_slab_part", "ndwt_analysis_level_slab_runs", "ndwt_synthesis_level_slab_runs", "ndwt_last_error",
    "ndwt_version", "ndwt_mplan_create", "ndwt_mplan_destroy", "ndwt_mplan_num_slabs", "ndwt_mplan_slab", "ndwt_mdec_host",
    "ndwt_mrec_host", "ndwt_mplan_last_error", "ndwt_mdec", "ndwt_mrec", "ndwt_mplan_set_exchange", "ndwt_mplan_describe", "ndwt_plan_slab_fast", "ndwt_dec_pitched", "ndwt_rec_pitched", "ndwt_shrink_pitched", "ndwt_band_pitch", "ndwt_slab_segments",
    "ndwt_plan_release_staging", "ndwt_mplan_set_overlap", "ndwt_mplan_last_enqueue_us", "ndwt_mplan_set_threads", "ndwt_comm_unique_id", "ndwt_comm_create", "ndwt_comm_destroy", "ndwt_comm_exchange",
    "ndwt_comm_last_error", "ndwt_coef_create", "ndwt_coef_release", "ndwt_coef_info", "ndwt_coef_dec_host", "ndwt_coef_rec_host",
    "ndwt_coef_shrink", "ndwt_coef_get_host", "ndwt_coef_put_host",
]


class NdwtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ndwt error {code}: {msg}")
        self.code = code
        self.message = msg


def build(verbose: bool = False) -> str:
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libndwt_hip.so failed")
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension must be built first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C non-decimated_wavelets_amd/csrc). "
            "This engine has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    c_void_pp = ctypes.POINTER(ctypes.c_void_p)
    L.ndwt_wave_filters.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_int)]
    L.ndwt_num_bands.argtypes = [ctypes.c_int, ctypes.c_int]
    L.ndwt_num_bands.restype = ctypes.c_int64
    L.ndwt_level_from_bands.argtypes = [ctypes.c_int, ctypes.c_int64]
    L.ndwt_plan_create.argtypes = [c_void_pp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_char_p),
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ndwt_plan_create_slab.argtypes = [c_void_pp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.POINTER(ctypes.c_char_p),
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ndwt_plan_destroy.argtypes = [ctypes.c_void_p]
    L.ndwt_plan_set_path.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndwt_plan_set_tuning.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.ndwt_plan_set_variant.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
    L.ndwt_plan_set_fused_level1.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndwt_plan_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndwt_plan_get_profile.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
    L.ndwt_plan_describe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    for f in (L.ndwt_dec, L.ndwt_rec):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    for f in (L.ndwt_dec_host, L.ndwt_rec_host):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.ndwt_shrink.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_dec_pitched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_rec_pitched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_shrink_pitched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_band_pitch.argtypes = [ctypes.c_void_p]
    L.ndwt_band_pitch.restype = ctypes.c_int64
    L.ndwt_denoise.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                               ctypes.c_void_p]
    L.ndwt_denoise_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    for f in (L.ndwt_dec_split, L.ndwt_rec_split):
        f.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
    for f in (L.ndwt_dec_split_host, L.ndwt_rec_split_host):
        f.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
    L.ndwt_slab_halo.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int64)] * 4
    L.ndwt_analysis_level_slab.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_void_pp, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_synthesis_level_slab.argtypes = [ctypes.c_void_p, c_void_pp, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_analysis_level_slab_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_void_pp,
                                                 ctypes.c_int, ctypes.c_void_p]
    L.ndwt_synthesis_level_slab_ext.argtypes = [ctypes.c_void_p, c_void_pp, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_analysis_level_slab_part.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_void_pp,
                                                ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
    L.ndwt_synthesis_level_slab_part.argtypes = [ctypes.c_void_p, c_void_pp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.ndwt_analysis_level_slab_runs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_void_pp, ctypes.c_int, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.ndwt_synthesis_level_slab_runs.argtypes = [ctypes.c_void_p, c_void_pp] + [ctypes.c_int64] * 5 + [ctypes.c_void_p, ctypes.c_int,
                                                                                                    ctypes.c_void_p]
    L.ndwt_mplan_create.argtypes = [c_void_pp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_char_p), ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.ndwt_mplan_destroy.argtypes = [ctypes.c_void_p]
    L.ndwt_mplan_num_slabs.argtypes = [ctypes.c_void_p]
    L.ndwt_mplan_slab.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64),
                                  ctypes.POINTER(ctypes.c_int64)]
    for f in (L.ndwt_mdec_host, L.ndwt_mrec_host):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.ndwt_mplan_last_error.restype = ctypes.c_char_p
    for f in (L.ndwt_mdec, L.ndwt_mrec):
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    L.ndwt_mplan_set_exchange.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndwt_mplan_set_overlap.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndwt_mplan_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ndwt_mplan_last_enqueue_us.argtypes = [ctypes.c_void_p]
    L.ndwt_mplan_last_enqueue_us.restype = ctypes.c_double
    L.ndwt_mplan_describe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.ndwt_plan_slab_fast.argtypes = [ctypes.c_void_p]
    L.ndwt_plan_release_staging.argtypes = [ctypes.c_void_p]
    L.ndwt_comm_unique_id.argtypes = [ctypes.c_void_p]
    L.ndwt_comm_create.argtypes = [c_void_pp, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ndwt_comm_destroy.argtypes = [ctypes.c_void_p]
    L.ndwt_comm_exchange.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), c_void_pp, ctypes.POINTER(ctypes.c_int64),
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    L.ndwt_comm_last_error.restype = ctypes.c_char_p
    L.ndwt_coef_create.argtypes = [ctypes.c_void_p, ctypes.c_int, c_void_pp]
    L.ndwt_coef_release.argtypes = [ctypes.c_void_p]
    L.ndwt_coef_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), c_void_pp]
    L.ndwt_coef_dec_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_void_pp]
    L.ndwt_coef_rec_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.ndwt_coef_shrink.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
    L.ndwt_coef_get_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.ndwt_coef_put_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, c_void_pp]
    L.ndwt_slab_segments.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_void_pp, c_void_pp, ctypes.POINTER(ctypes.c_int64),
                                     ctypes.c_void_p]
    L.ndwt_last_error.restype = ctypes.c_char_p
    L.ndwt_version.restype = ctypes.c_char_p
    _lib = L
    return L


def check(rc: int):
    if rc != NDWT_OK:
        raise NdwtError(rc, lib().ndwt_last_error().decode())


def mcheck(rc: int):
    if rc != NDWT_OK:
        raise NdwtError(rc, lib().ndwt_mplan_last_error().decode())


def wave_filters(wname: str):
    """(LO_D, HI_D) as lists -- Functions/wave_filters.m through the C ABI."""
    lo = (ctypes.c_double * 20)()
    hi = (ctypes.c_double * 20)()
    n = ctypes.c_int(0)
    check(lib().ndwt_wave_filters(wname.encode(), lo, hi, ctypes.byref(n)))
    return list(lo[: n.value]), list(hi[: n.value])
