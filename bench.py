#!/usr/bin/env python3
"""bench.py -- fwd+inv NDWT throughput on MI355X (BASELINE.json metric), one JSON line on rank 0.

Workload at N=1 (default): BASELINE config 3 -- 3-D fp32 512x512x512, db4, 3 levels, reference-parity dilation
(stride-1 taps at every level, what the reference computes), pres_l2_norm on, synthetic N(0,1) input resident in HBM.
A step = dec(x, level) followed by rec(y).  N>1: the same volume sharded on the outermost axis (strong scaling),
periodic halo exchange per level through torch.distributed (RCCL).  `--ndim 4` runs BASELINE config 5's transform
(256x256x256x32, t-sharded for N>1); `--wname db6 --level 4` config 4's.

Launch: `python bench.py --gpus N` starts its own N ranks (one child `torch.distributed.run` process, spawned before
anything touches the GPU) unless it is already running under a launcher (WORLD_SIZE set).

Algorithmic bytes: a level-direction moves (1 + 2^d) V sizeof(T); fwd+inv over L levels =
2 L (1 + 2^d) V sizeof(T) (3-D fp32 3 levels: 216 B/voxel).  The dominant kernel's roofline figure uses that kernel's own
algorithmic bytes per launch: fused 3-D level (1 + 8) V 4 B, one-axis pass (1 + 2) V 4 B.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s is the measured copy ceiling)
TRAFFIC_PROFILE = "r04_traffic.json"   # committed rocprofv3 --pmc figures of the headline command (fallback when the live passes cannot run)
KERNEL_NAMES = {0: "fused_analysis", 1: "fused_synthesis", 2: "axis_analysis", 3: "axis_synthesis"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ndim", type=int, default=3, choices=[2, 3, 4])
    ap.add_argument("--size", type=int, nargs="+", default=None, help="n1 n2 [n3 [n4]]; default 4096^2 / 512^3 / 256^3 x 32")
    ap.add_argument("--wname", default="db4")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="time the CPU baseline on the whole workload instead of the bounded sample "
                    "(512^3 complex128: ~160 GiB of host memory and minutes of CPU time)")
    ap.add_argument("--generic", action="store_true", help="force the per-axis kernels (for comparison)")
    ap.add_argument("--band-pitch", default="packed", choices=["packed", "auto"],
                    help="layout of the coefficient buffer between dec and rec: packed = the reference's; auto = ndwt_band_pitch()")
    ap.add_argument("--packed-only", action="store_true", help="skip the secondary pass with pitched coefficients (profiling runs: "
                    "both passes launch the same kernels, which a kernel-stats average would mix)")
    ap.add_argument("--no-others", action="store_true", help="skip the `sustained` run and the `other_configs` (cfg2 / cfg4 / cfg5) that "
                    "the default N=1 line carries after the headline's timed region")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not start the rocprofv3 --pmc child passes that measure "
                    "roofline.traffic of the headline configuration (the committed profile's figure is reported instead)")
    ap.add_argument("--zchunk", type=int, default=0)
    ap.add_argument("--target-blocks", type=int, default=0)
    a = ap.parse_args()
    if a.size is None:
        a.size = {2: [4096, 4096], 3: [512, 512, 512], 4: [256, 256, 256, 32]}[a.ndim]
    if len(a.size) != a.ndim:
        ap.error(f"--size needs {a.ndim} numbers")
    return a


def self_launch(a):
    """No launcher around us and --gpus N > 1: start the N ranks as ONE child process tree (torch.distributed.run) and relay
    its output.  Nothing in this process has touched the GPU (torch is not even imported yet)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd)
    sys.exit(r.returncode)


def usable_cores():
    """Cores this process may actually run on: the smaller of os.cpu_count(), the scheduler affinity mask and the cgroup's CPU quota (a GPU
    box hands one job a share of its host -- 16 of 256 cores on the pool's one-GPU boxes -- and a pool of os.cpu_count() threads on
    that share runs SLOWER than one of the share's size: 4.4 against 24 Mvoxels/s for the OpenMP filter bank)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_baseline(level, wname, sample_sizes, workers):
    """The reference's algorithm (FFT-domain fast convolution, op sequence of mex/nddwt.c) restated with scipy.fft on
    the host cores, complex128 like the mex path -- kind 'port'.  Timed on a bounded sample of the workload.
    Arrays are band-planar with each band contiguous, the reference's column-major layout."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
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
        y = np.ascontiguousarray(out.real)
        del out, approx
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


def cpu_spatial_baseline(level, wname, sample_sizes, threads):
    """The same transform as a signal-domain filter bank in C / OpenMP on the host cores, fp32 (oracle/ndwt_spatial.c, SURVEY 8d: `our
    spatial CPU backend on all cores`): what a CPU implementation that does not go through the DFT reaches.  Built here with -march=native
    (the portable library of the test suite is the fallback).  Returns (Mvoxels/s of the best of 3 dec+rec, seconds of that pass, threads)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tempfile
    import numpy as np
    import ndwt_spatial as sp
    try:
        lib = sp.load(sp.build(native=True, out_dir=tempfile.mkdtemp(prefix="ndwt_spatial_")))
    except Exception:
        lib = sp.load()
    had = lib.ndwt_c_max_threads()
    lib.ndwt_c_set_threads(threads)
    try:
        x = np.random.default_rng(0).standard_normal(tuple(reversed(sample_sizes)), dtype=np.float32)   # C order: outermost axis first
        wn = [wname] * len(sample_sizes)
        y = np.empty((sp.num_bands(len(sample_sizes), level),) + x.shape, dtype=np.float32)
        r = np.empty_like(x)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            sp.dec_planar(x, wn, level, 1, out=y, lib=lib)
            sp.rec_planar(y, wn, 1, level=level, out=r, lib=lib)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        err = float(np.abs(r - x).max())
        assert err < 2e-5, err
    finally:
        lib.ndwt_c_set_threads(had)
    return float(x.size) / best / 1e6, best, threads


def measure_config(api, torch, dev, d, sizes, wname, level, steps, warmup):
    """One non-headline BASELINE configuration on this GPU, packed (reference) coefficient layout: wall-clock mean and hipEvent
    median per dec+rec step, whole-step roofline fraction and the per-kernel HIP-event averages.  Runs AFTER the headline's
    timed region; its buffers are released before it returns."""
    import statistics
    V = 1
    for n in sizes:
        V *= n
    nbands = api.num_bands(d, level)
    plan = api.Plan(sizes, [wname] * d, torch.float32, False, True, "reference", max_level=max(level, 3), device=dev.index)
    x = torch.randn(tuple(reversed(sizes)), device=dev, dtype=torch.float32)
    y = torch.empty(nbands * V, device=dev, dtype=torch.float32)
    r = torch.empty_like(x)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        plan.dec(x.data_ptr(), y.data_ptr(), level, stream)
        plan.rec(y.data_ptr(), r.data_ptr(), level, stream)

    # warm-up: `warmup` steps and at least 80 ms of them -- a device that has idled (the allocations above) runs its first ~35 ms of work
    # at 1.06 - 1.5x the steady time (tools/ramp_profile.py), which is most of a 50-step measurement of cfg2's 0.4-ms step
    tw = time.perf_counter()
    done = 0
    while done < warmup or time.perf_counter() - tw < 0.08:
        step()
        done += 1
        if done % 8 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1 in evs:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize(dev)
    med = statistics.median(e0.elapsed_time(e1) for e0, e1 in evs)
    plan.set_profiling(True)
    psteps = max(1, min(steps, 5))
    for _ in range(psteps):
        step()
    torch.cuda.synchronize(dev)
    prof = {k: plan.get_profile(k) for k in KERNEL_NAMES}
    plan.set_profiling(False)
    rt = float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double()))
    step_bytes = 2 * level * (1 + (1 << d)) * V * 4
    nb3 = 1 << min(d, 3)                                          # bands of one fused launch: 4 (2-D level), 8 (3-D level / one t-band of a 4-D level)
    per_voxel = {0: 1 + nb3, 1: nb3 + 1, 2: 1 + 2, 3: 2 + 1}
    kern = {}
    for k, (tot, n) in prof.items():
        if n:
            avg = tot / n
            lps = n // psteps
            row = {"avg_launch_ms": round(avg, 4), "launches_per_step": lps}
            if k in (0, 1) and lps and lps < level * (2 if d == 4 else 1):
                # levels cascaded inside one launch (2-D): the launches of this direction together move the algorithmic bytes of `level` levels
                row["levels_per_launch"] = round(level / lps, 2)
                row["frac"] = round(level * per_voxel[k] * V * 4 / (avg * lps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            else:
                row["frac"] = round(per_voxel[k] * V * 4 / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            kern[KERNEL_NAMES[k]] = row
    shape = "x".join(str(n) for n in sizes)
    out = {"workload": f"{d}D fp32 {shape} {wname} {level} levels, dec+rec, packed (reference) coefficients", "path": plan.describe(),
           "steps": steps, "ms_per_step": round(dt * 1e3, 4), "median_ms_per_step_hip_events": round(med, 4),
           "value": round(V / dt / 1e6, 1), "unit": "Mvoxels/s", "algorithmic_bytes_per_step": step_bytes,
           "whole_step_frac": round(step_bytes / dt / 1e9 / HBM_PEAK_GBS, 4), "kernels": kern, "roundtrip_rel_l2": rt}
    del plan, x, y, r
    torch.cuda.empty_cache()
    return out


def one_rank_share(sh, torch, dev, wname, level, whole_ms, n_local=64, steps=100):
    """One rank's share of an 8-GPU run of a 512^3 volume, measured on THIS GPU: the 512 x 512 x n_local slab through the sharded driver
    at world size 1 (every exchange segment is a local copy, so this is the compute side only -- the projection `before communication`),
    with the exchange overlapped (interior planes, then the ends: 2 launches per level and direction) and in one piece per level, plus
    what tune() picks.  `x8_equivalent` = whole-volume ms / share ms: what 8 such ranks would give if the exchange were free."""
    import time
    x = torch.randn(n_local, 512, 512, device=dev, dtype=torch.float32)
    out = {"slab": f"512x512x{n_local} fp32 {wname} {level} levels, world size 1 (local-copy exchange)", "steps": steps,
           "whole_volume_ms": round(whole_ms, 4)}
    for key, mode, two in (("one_piece", False, False), ("overlap", True, False), ("overlap_two_streams", True, True)):
        eng = sh.ShardedNdDwt([wname] * 3, [512, 512, n_local], pres_l2_norm=True, precision="single", device=dev, overlap=mode,
                              two_streams=two)
        for _ in range(5):
            r = eng.rec(eng.dec(x, level))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            r = eng.rec(eng.dec(x, level))
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[key] = {"ms_per_dec_rec": round(ms, 4), "x8_equivalent": round(whole_ms / ms, 2),
                    "roundtrip_rel_l2": float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double()))}
        del eng
    eng = sh.ShardedNdDwt([wname] * 3, [512, 512, n_local], pres_l2_norm=True, precision="single", device=dev)   # overlap="auto"
    out["auto"] = eng.tune(x, level)
    del eng, x
    torch.cuda.empty_cache()
    return out


def mplan_host_time(wname, level, whole_ms, slabs=8):
    """ndwt_mplan_* with 8 slabs of cfg3, all on this device (a child process: tools/mplan_host_time.py): the HOST time ndwt_mdec + ndwt_mrec
    spend queueing a call for 8 devices -- one worker thread of the plan per slab, and from one thread -- next to one device's share of the
    compute.  A call whose queueing takes longer than that share would be bound by the host, not by the GPUs (it was: 4.0 ms from one thread
    at the start of round 4; profiles/r04_mplan_host_time.txt).  None if the child fails."""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mplan_host_time.py"), str(slabs), wname, str(level), "--json"],
                           capture_output=True, text=True, timeout=240)
        res = json.loads(r.stdout.strip().splitlines()[-1])
        res["whole_volume_ms"] = round(whole_ms, 4)
        return res
    except Exception:
        return None


def rccl_self_exchange(wname, level):
    """The same slab step with every exchange a real RCCL batch: a child process (torch.distributed is initialised in it, not here) creates a
    1-rank `nccl` group and runs the sharded driver with its self-segments routed through grouped send / receive (tools/host_overhead_nccl.py),
    once through torch.distributed's point-to-point ops and once as RCCL calls on the transform's own stream (ndwt_comm_*).  Nothing travels
    (sender = receiver = this GPU), so the difference to the local-copy figures is what six (eight) RCCL batches cost on the GPU's timeline and
    on the host: the floor of the exchange cost of an 8-GPU run.  None if the child fails."""
    import re
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_overhead_nccl.py"), wname, str(level)], capture_output=True, text=True,
                           timeout=240, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300)))
        out = {}
        for m in re.finditer(r"exchange=(local copies|RCCL send/recv to self|direct RCCL to self) overlap=(True|False): enqueue ([0-9.]+) ms per step, complete ([0-9.]+) ms", r.stdout):
            # torch_p2p_self: torch.distributed batch_isend_irecv (RCCL on its own stream); direct_rccl_self: ndwt_comm_exchange on the transform's stream
            key = {"l": "local_copies", "R": "torch_p2p_self", "d": "direct_rccl_self"}[m.group(1)[0]] + ("_overlap" if m.group(2) == "True" else "_one_piece")
            out[key] = {"ms_per_dec_rec": float(m.group(4)), "host_enqueue_ms": float(m.group(3))}
        return out or None
    except Exception:
        return None


def live_traffic(extra_args):
    """HBM-side bytes per launch of the fused kernels of THIS command line, measured now: two child processes
    `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 2 --warmup 1 ...` (separate passes, the program directly after
    `--`; MI355X_MICROARCH.md, HBM: bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB on gfx950).  None if rocprofv3 is missing or a pass fails."""
    import collections
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    res = {}
    tmp = tempfile.mkdtemp(prefix="ndwt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", os.path.join(tmp, c), "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--packed-only", "--no-others",
                   "--no-live-traffic"] + extra_args
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            if r.returncode != 0:
                return None
            agg = collections.defaultdict(list)
            for f in glob.glob(os.path.join(tmp, c, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"]
                    if "ndwt" in k and row["Counter_Name"] == c:
                        kind = "fused_synthesis" if "Inv" in k else "fused_analysis" if "Fwd" in k else "axis"
                        agg[kind].append(float(row["Counter_Value"]))
            for kind, v in agg.items():
                res.setdefault(kind, {})[c] = sum(v) / len(v)
        out = {}
        for kind, v in res.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                out[kind] = {"FETCH_SIZE_KiB": round(v["FETCH_SIZE"]), "WRITE_SIZE_KiB": round(v["WRITE_SIZE"]),
                             "traffic_bytes": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)}
        return out or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    a = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0:
        if a.gpus > 1:
            self_launch(a)                                         # does not return
        world = 1
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started {world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    sys.path.insert(0, ROOT)
    import importlib
    import torch
    api = importlib.import_module("non-decimated_wavelets_amd.api")

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

    d, sizes, level = a.ndim, list(a.size), a.level
    V = 1
    for n in sizes:
        V *= n
    nbands = api.num_bands(d, level)
    torch.manual_seed(1234 + rank)
    kshape = tuple(reversed(sizes))                                # kernel order: outermost axis first

    force_sharded = os.environ.get("NDWT_BENCH_FORCE_SHARDED", "0") == "1"   # exercise the N>1 code path on one GPU
    sharded = world > 1 or force_sharded
    if not sharded:
        plan = api.Plan(sizes, [a.wname] * d, torch.float32, False, True, "reference", max_level=max(level, 3), device=local_rank)
        plan.set_path(a.generic)
        plan.set_tuning(a.target_blocks, a.zchunk)
        x = torch.randn(kshape, device=dev, dtype=torch.float32)
        pitch = plan.band_pitch() if a.band_pitch == "auto" else 0
        y = torch.empty(nbands * (pitch if pitch else V), device=dev, dtype=torch.float32)
        r = torch.empty_like(x)
        stream = torch.cuda.current_stream(dev).cuda_stream
        lay = {"pitch": pitch, "y": y}

        def step():
            plan.dec(x.data_ptr(), lay["y"].data_ptr(), level, stream, band_pitch=lay["pitch"])
            plan.rec(lay["y"].data_ptr(), r.data_ptr(), level, stream, band_pitch=lay["pitch"])
    else:
        sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
        ov = os.environ.get("NDWT_BENCH_OVERLAP", "auto")        # A/B: 1 = always overlapped, 0 = exchange and compute in sequence
        eng = sh.ShardedNdDwt([a.wname] * d, sizes, pres_l2_norm=True, precision="single", group=None, device=dev,
                              overlap="auto" if ov == "auto" else ov == "1")
        x = torch.randn((eng.n_local,) + kshape[1:], device=dev, dtype=torch.float32)
        if eng.overlap_mode == "auto":
            eng.tune(x, level)                                    # untimed, before the warm-up: both schedules measured, the faster kept
        plan = eng.plan
        r_holder = {}

        def step():
            yl = eng.dec(x, level)
            r_holder["r"] = eng.rec(yl)

    def fence():
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The interpreter's cyclic garbage collector is kept out of the timed regions: a full collection walks every object torch has created
    # (38 ms measured) and, landing in a 20-step loop of 0.9-ms steps of the sharded driver, doubles the figure (tools/host_overhead_nccl.py).
    # Everything alive now is moved to the permanent generation; what the steps allocate is still collected, cheaply.  This happens BEFORE the
    # warm-up: the collection is 40 ms of host time with an idle device, and a device that has idled runs its next step at 1.5x and the
    # four after it at 1.06x the steady time (tools/ramp_profile.py) -- between the warm-up and the timed region it undid the warm-up
    # (20 timed steps: 5.80 ms per step against 5.58 ms over 500).
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(a.warmup):
        step()
    # ---- timed region: exactly a.steps steps, no per-kernel event recording inside it ----
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- per-step HIP events (median of >= 20 iterations), a pass of its own: torch's current stream is the launch stream ----
    import statistics
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for e0, e1 in evs:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize(dev)
    median_ms = statistics.median(e0.elapsed_time(e1) for e0, e1 in evs)

    # ---- separate pass for the per-kernel figures: HIP events around every launch, on the launch stream ----
    prof_steps = max(1, min(a.steps, 10))
    plan.set_profiling(True)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize(dev)
    prof = {k: plan.get_profile(k) for k in KERNEL_NAMES}
    plan.set_profiling(False)

    # round-trip check on the timed data
    if sharded:
        num = torch.linalg.vector_norm((r_holder["r"] - x).double()) ** 2
        den = torch.linalg.vector_norm(x.double()) ** 2
        if dist:
            num, den = num.to(red_dev), den.to(red_dev)
            dist.all_reduce(num)
            dist.all_reduce(den)
        rt_err = float(torch.sqrt(num / den))
    else:
        rt_err = float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double()))

    esize = 4
    ms_per_step = dt / a.steps * 1e3
    value = V / (dt / a.steps) / 1e6
    v_local = V // world
    step_bytes = 2 * level * (1 + (1 << d)) * V * esize          # whole job, BASELINE.md section 3
    # dominant kernel = the kind with the largest total time; its algorithmic bytes per launch
    dom = max(prof, key=lambda k: prof[k][0])
    nb3 = 1 << min(d, 3)
    per_voxel = {0: 1 + nb3, 1: nb3 + 1, 2: 1 + 2, 3: 2 + 1}   # volumes read + written per launch: fused 2-D / 3-D level, one-axis pass

    def kernel_row(k):
        tot_ms, n = prof[k]
        if n == 0:
            return None
        # one launch covers the rank's whole volume (a 4-D level runs the fused kernel twice, once per t-band, each on all
        # voxels).  The sharded driver cuts a level into pieces (interior planes + the two ends): their times are summed so
        # that the figure stays "time to move the algorithmic bytes of v_local voxels".
        units = prof_steps * level * (2 if (k in (0, 1) and d == 4) else 1) if sharded else n
        avg_ms = tot_ms / units
        bytes_launch = per_voxel[k] * v_local * esize
        ach = bytes_launch / (avg_ms * 1e-3) / 1e9
        return {"kernel": KERNEL_NAMES[k], "avg_launch_ms": round(avg_ms, 4), "launches": int(n), "algorithmic_bytes": bytes_launch,
                "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}

    rows = {k: kernel_row(k) for k in prof}
    drow = rows[dom]
    # HBM-side bytes per launch of the dominant kernel: FETCH_SIZE x 2 + WRITE_SIZE from separate rocprofv3 --pmc passes
    # of this command (MI355X_MICROARCH.md, HBM section), committed as profiles/r02_traffic.json -- null when this run is
    # not the profiled configuration
    traffic, traffic_src, traffic_all = None, None, None
    headline_cfg = (not sharded and d == 3 and sizes == [512, 512, 512] and a.wname == "db4" and level == 3 and not a.generic and
                    a.band_pitch == "packed" and a.zchunk == 0 and a.target_blocks == 0)
    if not sharded and rank == 0 and not a.no_live_traffic and a.band_pitch == "packed":
        # measured now, by two rocprofv3 --pmc child passes of this very command line (short: 2 steps)
        passthrough = ["--ndim", str(d), "--size"] + [str(n) for n in sizes] + ["--wname", a.wname, "--level", str(level)]
        if a.generic:
            passthrough.append("--generic")
        traffic_all = live_traffic(passthrough)
        if traffic_all and drow["kernel"] in traffic_all:
            traffic = traffic_all[drow["kernel"]]["traffic_bytes"]
            traffic_src = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (2 steps), (2*FETCH+WRITE) KiB per launch"
    if traffic is None and headline_cfg:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
            traffic = tj[drow["kernel"]]["traffic_bytes"]
            traffic_src = "profiles/" + TRAFFIC_PROFILE + " (committed profile of this command; not measured in this run)"
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": drow["kernel"], "achieved": drow["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": drow["frac"], "traffic": traffic, "traffic_source": traffic_src, "traffic_all_kernels": traffic_all,
                "algorithmic_bytes": drow["algorithmic_bytes"], "avg_launch_ms": drow["avg_launch_ms"], "launches": drow["launches"],
                "other_kernels": [rows[k] for k in rows if k != dom and rows[k] is not None],
                "whole_step_frac": round(step_bytes / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * world), 4)}

    # the same step with the coefficient buffer pitched (include/ndwt.h: ndwt_dec_pitched): what a caller that owns the buffer
    # gets; reported beside the headline, which keeps the reference's packed layout
    pitched = None
    if not sharded and a.band_pitch == "packed" and not a.packed_only:
        lay["y"] = y = None
        torch.cuda.empty_cache()                                  # (cfg5: 92 GiB; the pitched buffer is a little larger than the cached block)
        lay["pitch"] = plan.band_pitch()
        lay["y"] = torch.empty(nbands * lay["pitch"], device=dev, dtype=torch.float32)
        for _ in range(2):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        dtp = time.perf_counter() - t0
        plan.set_profiling(True)
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize(dev)
        pp = {k: plan.get_profile(k) for k in KERNEL_NAMES}
        plan.set_profiling(False)
        pitched = {"band_pitch_elements": lay["pitch"], "ms_per_step": round(dtp / a.steps * 1e3, 4), "value": round(V / (dtp / a.steps) / 1e6, 1),
                   "whole_step_frac": round(step_bytes / (dtp / a.steps) / 1e9 / HBM_PEAK_GBS, 4),
                   "avg_launch_ms": {KERNEL_NAMES[k]: round(pp[k][0] / pp[k][1], 4) for k in pp if pp[k][1]},
                   "roundtrip_rel_l2": float(torch.linalg.vector_norm((r - x).double()) / torch.linalg.vector_norm(x.double()))}
        try:
            if d == 3 and sizes == [512, 512, 512] and a.wname == "db4" and level == 3 and not a.generic:
                tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
                pitched["traffic"] = {k: tj["pitched"][k]["traffic_bytes"] for k in ("fused_synthesis", "fused_analysis")}
                pitched["traffic_source"] = "profiles/" + TRAFFIC_PROFILE
        except Exception:
            pass

    shape = "x".join(str(n) for n in sizes)
    cube = f"{sizes[0]}^3" if d == 3 and len(set(sizes)) == 1 else shape
    out = {"metric": f"Mvoxels/s fwd+inv NDWT ({cube} fp32, {level} lvl {a.wname})", "value": round(value, 1), "unit": "Mvoxels/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
           "median_ms_per_step_hip_events": round(median_ms, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{d}D fp32 {shape} {a.wname} {level} levels, dec+rec, pres_l2_norm, reference dilation (stride 1)",
                      "sharding": "none" if not sharded else f"outer-axis slabs x{world}; per level: analysis halo fetch (1 band) and synthesis "
                                                             f"scatter-add (1 band) via RCCL send/recv; exchange "
                                                             f"{'overlapped with the interior planes' if eng.overlap else 'then one launch per level'}"
                                                             f" (overlap={eng.overlap_mode}, tune: {eng.tuned})",
                      "path": "per-axis" if a.generic else plan.describe(),
                      "coefficient_layout": ("pitched local slabs owned by the sharded driver" if sharded else
                                             "packed (reference)" if a.band_pitch == "packed" else "pitched bands (ndwt_band_pitch)")},
           "roofline": roofline, "roundtrip_rel_l2": rt_err}
    if pitched is not None:
        out["pitched_coefficients"] = pitched
    # ---- after the headline: a sustained run of the same step and the other single-GPU BASELINE configurations ----
    if not sharded and headline_cfg and not a.no_others:
        lay["pitch"], lay["y"] = 0, None
        y = None
        torch.cuda.empty_cache()
        lay["y"] = torch.empty(nbands * V, device=dev, dtype=torch.float32)
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(500):
            step()
        fence()
        dts = (time.perf_counter() - t0) / 500
        out["sustained"] = {"steps": 500, "ms_per_step": round(dts * 1e3, 4), "value": round(V / dts / 1e6, 1),
                            "whole_step_frac": round(step_bytes / dts / 1e9 / HBM_PEAK_GBS, 4)}
        lay["y"] = None
        del x, r
        torch.cuda.empty_cache()
        others = {}
        for name, (dd, ss, wn, lv, st) in {"cfg2": (2, [4096, 4096], "db4", 3, 50), "cfg4_transform": (3, [512, 512, 512], "db6", 4, 20),
                                           "cfg5": (4, [256, 256, 256, 32], "db4", 3, 5)}.items():
            try:
                others[name] = measure_config(api, torch, dev, dd, ss, wn, lv, st, 3)
            except Exception as e:                                # (a box with less free memory than cfg5's 100 GB: say so, keep the line)
                others[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
        out["other_configs"] = others
        # one rank's share of the 8-GPU shardings of cfg3 / cfg4 on this GPU (no communication: the projection SURVEY 8e's >= 6x rests on)
        try:
            sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
            share = {"cfg3": one_rank_share(sh, torch, dev, "db4", 3, dts * 1e3)}
            if "ms_per_step" in others.get("cfg4_transform", {}):
                share["cfg4"] = one_rank_share(sh, torch, dev, "db6", 4, others["cfg4_transform"]["ms_per_step"])
            rs = rccl_self_exchange("db4", 3)                     # (child process; this one never initialises torch.distributed at N = 1)
            if rs:
                for k, v in rs.items():
                    v["x8_equivalent"] = round(dts * 1e3 / v["ms_per_dec_rec"], 2)
                share["cfg3"]["with_rccl_batches_to_self"] = rs
            try:                                                  # the single-process multi-device plan (the path behind one MATLAB process)
                share["single_process_plan"] = mplan_host_time(a.wname, level, dts * 1e3)
            except Exception as e:
                share["single_process_plan"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            out["one_rank_share"] = share
        except Exception as e:
            out["one_rank_share"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        host = os.cpu_count() or 1
        cores = usable_cores()                                    # affinity mask / cgroup quota: the share of the host this job may use
        if a.cpu_full:
            sample = list(sizes)                                  # the whole workload (512^3: ~160 GiB of host memory, minutes)
        elif d == 2:
            sample = [4096, 4096] if cores >= 16 else [2048, 2048]
        elif d == 3:
            sample = [512, 512, 128] if cores >= 16 else [192, 192, 128]
        else:
            sample = [128, 128, 64, 32] if cores >= 16 else [64, 64, 32, 32]
        # the reference's own thread setting (fftw_plan_with_nthreads(8), mex/nddwt.c:103,146) AND every host core
        w_ref = min(8, cores)
        v, secs = cpu_baseline(level, a.wname, sample, w_ref)
        desc = (f"{'x'.join(map(str, sample))} ({'the whole workload' if sample == list(sizes) else 'bounded sample of ' + shape}) fp64/complex128 "
                f"{a.wname} {level} levels dec+rec, FFT-domain restatement of mex/nddwt.c (scipy.fft/pocketfft in place of FFTW)")
        out["cpu_baseline"] = {"value": round(v, 2), "unit": "Mvoxels/s", "cores": w_ref, "host_cpu_count": host, "usable_cores": cores, "kind": "port",
                               "sample": f"{desc}; workers={w_ref} = the reference's fftw_plan_with_nthreads(8); {secs:.1f} s"}
        if cores > w_ref:
            v2, secs2 = cpu_baseline(level, a.wname, sample, cores)
            out["cpu_baseline"]["all_cores"] = {"value": round(v2, 2), "unit": "Mvoxels/s", "cores": cores,
                                                "sample": f"the same sample, workers={cores} (every core this job may use: affinity mask / cgroup quota; "
                                                          f"the host has {host}); {secs2:.1f} s"}
        try:
            v3, secs3, th = cpu_spatial_baseline(level, a.wname, sample, cores)
            out["cpu_baseline"]["spatial_port"] = {"value": round(v3, 1), "unit": "Mvoxels/s", "cores": th, "kind": "port",
                                                   "sample": f"the same sample in fp32 as a signal-domain filter bank in C / OpenMP (oracle/ndwt_spatial.c: "
                                                             f"periodic correlations axis by axis, not the reference's DFT-domain algorithm), {th} threads, "
                                                             f"best of 3 dec+rec; {secs3:.2f} s"}
        except Exception as e:                                    # (no gcc / no OpenMP on the host: the reference-algorithm figures stand alone)
            out["cpu_baseline"]["spatial_port"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
