#!/usr/bin/env python3
"""Small problems are launch-bound: dec+rec (3 levels, db4 fp32; 1-D: db2 fp64 = cfg1's shape) issued call by call against one replay of
the same calls captured into a HIP graph (torch.cuda.CUDAGraph on the plan's stream).  python tools/bench_graph.py"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
api = importlib.import_module("non-decimated_wavelets_amd.api")
for dims, wname, dt in (([4096], "db2", torch.float64), ([256, 256], "db4", torch.float32), ([1024, 1024], "db4", torch.float32), ([32, 32, 32], "db4", torch.float32),
                        ([64, 64, 64], "db4", torch.float32), ([128, 128, 128], "db4", torch.float32), ([256, 256, 256], "db4", torch.float32),
                        ([32, 32, 16, 16], "db4", torch.float32)):
    d, level = len(dims), 3
    plan = api.Plan(dims, [wname] * d, dt, False, True, "reference", max_level=level)
    shp = tuple(reversed(dims))
    x = torch.randn(*shp, device="cuda", dtype=dt)
    y = torch.empty((api.num_bands(d, level),) + shp, device="cuda", dtype=dt)
    r = torch.empty_like(x)
    st = torch.cuda.Stream()

    def step():
        plan.dec(x.data_ptr(), y.data_ptr(), level, st.cuda_stream)
        plan.rec(y.data_ptr(), r.data_ptr(), level, st.cuda_stream)

    with torch.cuda.stream(st):
        for _ in range(3):
            step()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            step()
        res = {}
        for name, fn in (("eager", step), ("graph", g.replay)):
            for _ in range(20):
                fn()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(200):
                fn()
            e1.record(st)
            st.synchronize()
            res[name] = e0.elapsed_time(e1) / 200 * 1e3
    print(f"{'x'.join(map(str, dims)):>14s} {wname} {str(dt).split('.')[-1]} L{level}: eager {res['eager']:8.1f} us  graph replay {res['graph']:8.1f} us per dec+rec  "
          f"({plan.describe()})", flush=True)
