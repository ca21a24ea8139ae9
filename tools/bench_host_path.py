#!/usr/bin/env python3
"""The host-pointer forms (what the MATLAB gateway calls): ndwt_dec_host / ndwt_rec_host / ndwt_denoise_host on pageable numpy arrays.
Time per call and the implied PCIe rate of the dominant copy.  python tools/bench_host_path.py [n] [reps]"""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
level = 3
plan = api.Plan([n, n, n], ["db4"] * 3, torch.float32, False, True, "reference", max_level=level)
nb = api.num_bands(3, level)
x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
y = np.empty((nb, n, n, n), dtype=np.float32)
r = np.empty_like(x)
d = np.empty_like(x)
y[:] = 0          # touch the pages
r[:] = 0
d[:] = 0
import ctypes  # noqa: E402
lib = importlib.import_module("non-decimated_wavelets_amd._lib")
L = lib.lib()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
for name, fn, nbytes in (("dec_host", lambda: lib.check(L.ndwt_dec_host(plan._h, vp(x), vp(y), level)), y.nbytes),
                         ("rec_host", lambda: lib.check(L.ndwt_rec_host(plan._h, vp(y), vp(r), level)), y.nbytes),
                         ("denoise_host", lambda: plan.denoise_host(x.ctypes.data, d.ctypes.data, level, 0.3), 2 * x.nbytes)):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:13s} {n}^3 fp32 L{level}: {dt * 1e3:9.1f} ms per call, {nbytes / dt / 1e9:6.1f} GB/s over its {nbytes / 1e9:.2f} GB of host traffic", flush=True)
print("round trip", float(np.linalg.norm(r - x) / np.linalg.norm(x)))
# the same solver step with the coefficients behind a device-resident handle (the gateway's dec_keep / shrink / rec_handle): only x crosses
c = api.Coefficients.dec(plan, x, level)
for name, fn, nbytes in (("coef dec", lambda: api.Coefficients.dec(plan, x, level, reuse=c), x.nbytes),
                         ("coef shrink", lambda: c.shrink(0.3), 0),
                         ("coef rec", lambda: c.rec(r), x.nbytes)):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:13s} {n}^3 fp32 L{level}: {dt * 1e3:9.1f} ms per call" + (f", {nbytes / dt / 1e9:6.1f} GB/s over its {nbytes / 1e9:.2f} GB of host traffic" if nbytes else ""),
          flush=True)
c.release()
