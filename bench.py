#!/usr/bin/env python3
"""bench.py -- fwd+inv NDWT throughput on MI355X (BASELINE.json metric), one JSON line on rank 0.

Workload at N=1: BASELINE config 3 -- 3-D fp32 512x512x512, db4, 3 levels, reference-parity dilation
(stride-1 taps at every level, what the reference computes), pres_l2_norm on, synthetic N(0,1) input resident
in HBM.  A step = dec(x, 3) followed by rec(y).  N>1: the same 512^3 volume sharded on the outermost axis
(strong scaling), periodic halo exchange per level through torch.distributed (RCCL).

Algorithmic bytes: each level-direction launch moves (1 + 2^d) V sizeof(T) = 36 B/voxel;
fwd+inv over 3 levels = 216 B/voxel.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s is the measured copy ceiling)


def cpu_baseline(level, wname, sample_sizes, workers):
    """The reference's algorithm (FFT-domain fast convolution, op sequence of mex/nddwt.c) restated with scipy.fft on
    the host cores, complex128 like the mex path -- kind 'port'.  Timed on a bounded sample of the workload.
    Arrays are band-planar with each band contiguous, the reference's column-major layout."""
    import numpy as np
    import scipy.fft as sfft
    import ndwt_oracle as orc
    d = len(sample_sizes)
    nb = 1 << d
    rng = np.random.default_rng(0)
    x = rng.standard_normal(sample_sizes)
    m = orc.NdDwtMat(wname, sample_sizes, 1, compute="mex")       # f_dec incl. the 1/N of the mex path; construction is
    f_dec = np.ascontiguousarray(np.moveaxis(m.f_dec, -1, 0))     # untimed, like the class constructor
    axes = tuple(range(1, d + 1))
    nbt = orc.num_bands(d, level)
    t0 = time.perf_counter()
    with sfft.set_workers(workers):
        # dec: nd_dwt_3D.m:157 + nddwt.c:189-239
        approx = sfft.fftn(x)
        out = np.empty([nbt] + list(sample_sizes), dtype=np.complex128)
        for lev in range(level, 0, -1):
            s0 = (nb - 1) * (lev - 1)
            out[s0:s0 + nb] = sfft.ifftn(approx[None] * f_dec, axes=axes, norm="forward")   # pointByPoint + batched inverse
            approx = sfft.fftn(out[s0])
        y = out.real
        # rec: nd_dwt_3D.m:220 + nddwt.c:242-292
        c_f = sfft.fftn(y, axes=axes)
        cur = None
        for ind in range(1, level + 1):
            s0 = (nb - 1) * (ind - 1)
            if cur is not None:
                c_f[s0] = sfft.fftn(cur)
            cur = sfft.ifftn(c_f[s0:s0 + nb] * np.conj(f_dec), axes=axes, norm="forward").sum(axis=0)
        r = cur.real
    dt = time.perf_counter() - t0
    err = float(np.abs(r - x).max())
    assert err < 1e-9, err
    return float(np.prod(sample_sizes)) / dt / 1e6, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--wname", default="db4")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the per-axis kernels (for comparison)")
    ap.add_argument("--zchunk", type=int, default=0)
    ap.add_argument("--target-blocks", type=int, default=0)
    a = ap.parse_args()

    import torch
    import importlib
    pkg = importlib.import_module("non-decimated_wavelets_amd")
    api = importlib.import_module("non-decimated_wavelets_amd.api")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # rehearsal of the N > 1 path on a one-GPU box: NDWT_BENCH_BACKEND=gloo (slabs staged through the host) with
    # NDWT_BENCH_ONE_GPU=1 (every rank on device 0).  The driver's runs use neither: one rank per GPU over RCCL.
    backend = os.environ.get("NDWT_BENCH_BACKEND", "nccl")
    if os.environ.get("NDWT_BENCH_ONE_GPU", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")     # where the scalar reductions of the report live

    n1, n2, n3 = a.size
    V = n1 * n2 * n3
    level = a.level
    nb = api.num_bands(3, level)
    torch.manual_seed(1234 + rank)

    force_sharded = os.environ.get("NDWT_BENCH_FORCE_SHARDED", "0") == "1"   # exercise the N>1 code path on one GPU
    if world == 1 and not force_sharded:
        plan = api.Plan([n1, n2, n3], [a.wname] * 3, torch.float32, False, True, "reference", max_level=max(level, 3), device=local_rank)
        plan.set_path(a.generic)
        plan.set_tuning(a.target_blocks, a.zchunk)
        x = torch.randn(n3, n2, n1, device=dev, dtype=torch.float32)
        y = torch.empty(nb, n3, n2, n1, device=dev, dtype=torch.float32)
        r = torch.empty_like(x)
        stream = torch.cuda.current_stream(dev).cuda_stream

        def step():
            plan.dec(x.data_ptr(), y.data_ptr(), level, stream)
            plan.rec(y.data_ptr(), r.data_ptr(), level, stream)
        kinds = (2, 3) if a.generic else (0, 1)
    else:
        sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
        eng = sh.ShardedNdDwt([a.wname] * 3, [n1, n2, n3], pres_l2_norm=True, precision="single", group=None, device=dev,
                              overlap=os.environ.get("NDWT_BENCH_OVERLAP", "1") == "1")   # 0: exchange and compute in sequence (A/B)
        x = torch.randn(eng.n_local, n2, n1, device=dev, dtype=torch.float32)
        plan = eng.plan
        r_holder = {}

        def step():
            yl = eng.dec(x, level)
            r_holder["r"] = eng.rec(yl)
        kinds = (0, 1)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    plan.set_profiling(True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = {k: plan.get_profile(k) for k in kinds}
    plan.set_profiling(False)

    # round-trip check on the timed data (world == 1)
    rt_err = None
    if world > 1 or force_sharded:
        num = torch.linalg.vector_norm((r_holder["r"] - x).double()) ** 2
        den = torch.linalg.vector_norm(x.double()) ** 2
        if dist:
            num, den = num.to(red_dev), den.to(red_dev)
            dist.all_reduce(num)
            dist.all_reduce(den)
        rt_err = float(torch.sqrt(num / den))
    elif world == 1:
        rt_err = float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double()))

    ms_per_step = dt / a.steps * 1e3
    value = V / (dt / a.steps) / 1e6
    esize = 4
    v_local = V // world
    bytes_per_launch = (1 + 8) * v_local * esize                 # one level, one direction (36 B/voxel fp32)
    # dominant kernel = the kind with the larger total time
    dom = max(kinds, key=lambda k: prof[k][0])
    dom_ms, dom_n = prof[dom]
    # one level of one direction = one launch on a whole volume; the sharded path cuts it into pieces (interior + ends)
    # that are summed here so that the figure stays "time to move one level's algorithmic bytes"
    per_level = (world > 1 or force_sharded)
    avg_ms = dom_ms / max(a.steps * level if per_level else dom_n, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    oth = [k for k in kinds if k != dom][0]
    names = {0: "fused3_synthesis" if False else "fused3_analysis", 1: "fused3_synthesis", 2: "axis_analysis", 3: "axis_synthesis"}
    # HBM-side bytes per launch of that kernel from the committed PMC passes (profiles/r01_traffic.json; measured on
    # this workload, not live) -- null when the run is not the profiled configuration
    traffic = None
    try:
        if world == 1 and [n1, n2, n3] == [512, 512, 512] and a.wname == "db4" and not a.generic:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))[names[dom]]["traffic_bytes"]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes": bytes_per_launch,
                "avg_launch_ms": round(avg_ms, 4), "launches": int(dom_n),
                "other_kernel": {"kernel": names[oth], "avg_launch_ms": round(prof[oth][0] / max(a.steps * level if per_level else prof[oth][1], 1), 4)},
                "whole_step_frac": round((2 * level * bytes_per_launch * world) / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * world), 4)}

    out = {"metric": "Mvoxels/s fwd+inv NDWT (512^3 fp32, 3 lvl db4)", "value": round(value, 1), "unit": "Mvoxels/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"3D fp32 {n1}x{n2}x{n3} {a.wname} {level} levels, dec+rec, pres_l2_norm, reference dilation (stride 1)",
                      "sharding": "none" if world == 1 else f"outer-axis slabs x{world}; per level: analysis halo fetch (1 band) and synthesis "
                                                              f"scatter-add (1 band) via RCCL send/recv, overlapped with the interior planes",
                      "path": "per-axis" if a.generic else "fused3d"},
           "roofline": roofline}
    if rt_err is not None:
        out["roundtrip_rel_l2"] = rt_err
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cores = os.cpu_count() or 1
        workers = min(cores, 16)
        sample = [512, 512, 128] if cores >= 16 else [192, 192, 128]
        v, secs = cpu_baseline(level, a.wname, sample, workers)
        out["cpu_baseline"] = {"value": round(v, 2), "unit": "Mvoxels/s", "cores": workers, "kind": "port",
                               "sample": f"{sample[0]}x{sample[1]}x{sample[2]} fp64/complex128 {a.wname} {level} levels dec+rec, FFT-domain "
                                         f"restatement of mex/nddwt.c with scipy.fft workers={workers}; {secs:.1f} s"}
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
