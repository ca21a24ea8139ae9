"""MI355X-native non-decimated wavelet transform engine (drop-in for the nd_dwt_{1,2,3,4}D hot path).

Import with `importlib.import_module("non-decimated_wavelets_amd")` (the directory name is not a
Python identifier) or through the `ndwt_amd` alias module at the repository root.
"""
from ._lib import LIB_PATH, NdwtError, build, lib, wave_filters  # noqa: F401

__all__ = ["LIB_PATH", "NdwtError", "build", "lib", "wave_filters", "Plan", "num_bands", "nd_dwt_1D", "nd_dwt_2D",
           "nd_dwt_3D", "nd_dwt_4D", "ShardedNdDwt", "MultiPlan", "Coefficients"]


def __getattr__(name):   # torch-dependent parts are imported lazily so the C-ABI checks work without a GPU stack
    if name in ("Plan", "MultiPlan", "Coefficients", "num_bands", "nd_dwt_1D", "nd_dwt_2D", "nd_dwt_3D", "nd_dwt_4D"):
        from . import api
        return getattr(api, name)
    if name == "ShardedNdDwt":
        from . import sharded
        return sharded.ShardedNdDwt
    raise AttributeError(name)
