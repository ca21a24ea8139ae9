"""The host-array forms of the single-process multi-device plan (what the MATLAB gateway calls with 'devices'): ndwt_mdec_host / ndwt_mrec_host of
cfg3 on pageable numpy arrays, G slabs on device 0, with the per-slab host threads (which also queue the slabs' host <-> device copies) and
without.  On one GPU all slabs share one PCIe link; with G devices each thread serves its own.  python tools/mplan_host_path.py [G] [n]"""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
level = 3
x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
for threads in (True, False, True, False):
    mp = api.MultiPlan([n, n, n], ["db4"] * 3, torch.float32, [0] * G, pres_l2_norm=True, max_level=level).set_threads(threads)
    import ctypes
    L = importlib.import_module("non-decimated_wavelets_amd._lib")
    y = mp.dec(x, level)                       # (a fresh numpy array: its pages are touched for the first time by this call)
    r = mp.rec(y)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    t0 = time.perf_counter()
    L.mcheck(L.lib().ndwt_mdec_host(mp._h, vp(x), vp(y), level))     # into the array that exists already
    t1 = time.perf_counter()
    L.mcheck(L.lib().ndwt_mrec_host(mp._h, vp(y), vp(r), level))
    t2 = time.perf_counter()
    gb = y.nbytes / 1e9
    print(f"{G} slabs, one host thread per slab={threads}: dec_host {1e3 * (t1 - t0):7.1f} ms ({gb / (t1 - t0):5.1f} GB/s of coefficients)  "
          f"rec_host {1e3 * (t2 - t1):7.1f} ms ({gb / (t2 - t1):5.1f} GB/s)  max |rec - x| {np.abs(r - x).max():.1e}")
    del mp, y, r
