"""Outer-axis slab decomposition of the NDWT across the GPUs of one node (one process per GPU).

The reference is single-process (no MPI/NCCL anywhere); this is the multi-GPU extension named
by BASELINE.json: 3-D/4-D volumes are sharded on the outermost axis, every band of every level is sharded the same
way, and each level needs one periodic halo exchange on that axis -- `torch.distributed` point-to-point sends
(backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU tests).

Two exchange schemes:
  * analysis: the (L/2-1)*s planes before and (L/2)*s planes after the slab of the APPROXIMATION band are fetched
    from their owners (1 band per level).
  * synthesis, scheme "scatter" (default where the engine supports it): each rank synthesises its own coefficient
    slab zero-extended, which yields partial sums for the (L/2-1)*s planes before and (L/2)*s planes after its slab;
    those partial planes (1 band) are sent to their owners and added -- 2^d times less traffic than fetching the
    halo of all 2^d bands, and no haloed copy of the coefficients.  Summation order differs from the single-device
    kernel, so results agree to rounding, not bit for bit.
  * synthesis, scheme "gather": fetch the halo planes of all 2^d bands, then run the slab synthesis.  Bit-exact
    with the single-device result; used for the per-axis path.

Tensors are in kernel order: x_local is (n_local, n_{d-1}, ..., n1) contiguous, coefficients are
(bands, n_local, ..., n1) -- i.e. the column-major MATLAB arrays [n1, ..., n_local, bands] of the reference.
The local compute is delegated to an engine object; the product engine is `HipSlabEngine` (HIP kernels through the
C ABI).  The CPU tests inject an oracle-backed engine to exercise the exchange logic without a GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib as L


def partition(n: int, world: int):
    """plane ranges [lo, hi) of every rank (uneven remainders allowed)"""
    return [(r * n // world, (r + 1) * n // world) for r in range(world)]


class HipSlabEngine:
    """Local compute of one slab on one GPU through include/ndwt.h (slab entry points)."""

    def __init__(self, wnames, local_dims, dtype, pres_l2_norm, dilation, device, global_outer=None):
        from .api import Plan
        # a slab plan: the filter-length check of the reference applies to the whole sharded axis (global_outer), the local
        # slab may be thinner than the filter (cfg5: 4 frames per rank, 8 taps)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.plan = Plan(local_dims, wnames, dtype, False, pres_l2_norm, dilation, max_level=1, device=idx,
                         global_outer=global_outer if global_outer is not None else local_dims[-1])
        self.device = torch.device("cuda", idx)
        self._parts = None
        self.dtype = dtype
        self.local_dims = list(local_dims)
        # split-halo analysis + zero-extended synthesis entry points (fused 3-D kernels only)
        lens = [len(L.wave_filters(w)[0]) for w in wnames]
        # copy-free analysis (separate halo buffers), run-of-planes pieces: fused 3-D plans only
        self.supports_split = (dilation == "reference" and len(local_dims) == 3 and self.plan.describe() == "fused3d"
                               and lens[2] == max(lens))
        self.supports_overlap = self.supports_split
        # zero-extended synthesis (scatter-add exchange of 1 band): also 4-D plans sharded on t (3-D part per frame)
        self.supports_scatter = self.supports_split or (dilation == "reference" and len(local_dims) == 4
                                                        and self.plan.describe() == "axis+fused3d")

    def halo(self, stride):
        return self.plan.slab_halo(stride)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def segments(self, add, pairs):
        """pairs of equally shaped contiguous tensors (dst, src): dst = src or dst += src, all pairs in ONE launch (a launch costs ~6 us
        whatever it moves; a level has two such runs: the planes a slab takes from itself, the partial sums of the two neighbours)"""
        scal = 2 if pairs[0][0].is_complex() else 1
        for i in range(0, len(pairs), 8):
            part = pairs[i:i + 8]
            self.plan.slab_segments(add, [d.data_ptr() for d, _ in part], [s.data_ptr() for _, s in part],
                                    [d.numel() * scal for d, _ in part], self._stream())

    def analysis(self, in_with_halo, outs, stride):
        self.plan.analysis_level_slab(in_with_halo.data_ptr(), [o.data_ptr() for o in outs], stride, self._stream())

    def synthesis(self, ins_with_halo, out, stride):
        self.plan.synthesis_level_slab([t.data_ptr() for t in ins_with_halo], out.data_ptr(), stride, self._stream())

    def analysis_split(self, in_local, halo_before, halo_after, outs, stride):
        self.plan.analysis_level_slab_split(in_local.data_ptr(), halo_before.data_ptr(), halo_after.data_ptr(),
                                            [o.data_ptr() for o in outs], stride, self._stream())

    def synthesis_ext(self, ins_local, out_ext, stride):
        self.plan.synthesis_level_slab_ext([t.data_ptr() for t in ins_local], out_ext.data_ptr(), stride, self._stream())

    # runs of planes (ndwt_*_slab_part): the pieces the exchange overlaps with.  Pointers are computed from the base
    # addresses (no tensor views: this runs once per piece per level on the host's critical path).
    def analysis_run(self, cur, hb, ha, outs, z0, z1, stride):
        """output planes [z0, z1) of the slab's analysis level.  The planes around the run come from the slab itself,
        or -- for a run that starts at plane 0 / ends at the last plane -- from the received halo buffers hb / ha."""
        ab, aa, _, _ = self.halo(stride)
        n, pb = cur.shape[0], cur.stride(0) * cur.element_size()
        base = cur.data_ptr()
        assert (z0 == 0 or z0 >= ab) and (z1 == n or z1 + aa <= n)
        before = (hb.data_ptr() if ab else None) if z0 == 0 else base + (z0 - ab) * pb
        after = ha.data_ptr() if z1 == n else base + z1 * pb
        self.plan.analysis_level_slab_part(base + z0 * pb, before, after, [o.data_ptr() + z0 * pb for o in outs], z1 - z0,
                                           stride, self._stream())

    def analysis_ends(self, slab_buf, outs, stride):
        """the first and the last max(ab, aa) output planes in ONE launch.  slab_buf: [halo_before | slab | halo_after]
        contiguous, (ab + n + aa) planes, the halo planes received in place."""
        ab, aa, _, _ = self.halo(stride)
        n, m = slab_buf.shape[0] - ab - aa, max(ab, aa)
        self.plan.analysis_level_slab_runs(slab_buf.data_ptr(), [o.data_ptr() for o in outs], m, 2, n - m, stride, self._stream())

    def synthesis_part(self, ins_local, e0, out_run, stride):
        """planes [e0, e0 + len(out_run)) of the zero-extended synthesis of the local coefficient slab"""
        self.plan.synthesis_level_slab_part([t.data_ptr() for t in ins_local], ins_local[0].shape[0], e0, out_run.shape[0],
                                            out_run.data_ptr(), stride, self._stream())

    def synthesis_send_parts(self, ins_local, stride):
        """(part_before, part_after): the partial sums owed to the slabs ahead (sa planes) and behind (sb planes), both
        from ONE launch (two runs of max(sa, sb) planes at the two ends of the zero-extended result)."""
        _, _, sb, sa = self.halo(stride)
        n, m = ins_local[0].shape[0], max(sa, sb)
        shape = (2, m) + tuple(ins_local[0].shape[1:])
        if self._parts is None or tuple(self._parts.shape) != shape or self._parts.dtype != ins_local[0].dtype:
            self._parts = ins_local[0].new_empty(shape)
        buf = self._parts
        self.plan.synthesis_level_slab_runs([t.data_ptr() for t in ins_local], n, 0, n + sa + sb - m, 2, m, buf.data_ptr(), stride,
                                            self._stream())
        return buf[0, :sa], buf[1, m - sb:]


class DirectRccl:
    """The exchange as RCCL point-to-point calls on the CURRENT stream (include/ndwt.h: ndwt_comm_*), without the two cross-stream
    dependencies torch.distributed's NCCL work pays per batch (its own stream): 20 us instead of 69 us per exchange on one MI355X.  Created
    collectively over a torch.distributed group of any backend: rank 0's id travels through broadcast_object_list."""

    def __init__(self, group, device):
        import ctypes
        self._ct = ctypes
        self._h = ctypes.c_void_p(None)
        self._cache = {}
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = [None]
        err = None
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            if L.lib().ndwt_comm_unique_id(buf) != L.NDWT_OK:
                err = L.lib().ndwt_comm_last_error().decode()
            ident = [bytes(buf.raw)]
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None and group is not dist.group.WORLD else 0, group=group)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if err is None and L.lib().ndwt_comm_create(ctypes.byref(self._h), ident[0], world, rank, idx) != L.NDWT_OK:
            err = L.lib().ndwt_comm_last_error().decode()
            self._h = ctypes.c_void_p(None)
        # usable only if EVERY rank has it
        flags = [None] * world
        dist.all_gather_object(flags, err, group=group)
        bad = [f"rank {r}: {e}" for r, e in enumerate(flags) if e is not None]
        if bad:
            self.close()
            raise RuntimeError("direct RCCL transport unavailable (" + "; ".join(bad) + ")")
        self.device = device

    def exchange(self, ops):
        """ops: [(is_send, contiguous tensor, peer rank)] in the group's segment order -> one ncclGroup on the current stream"""
        if not ops:
            return
        ct, n = self._ct, len(ops)
        # the driver's exchange buffers are allocated once, so the same few argument lists come back every step: built once, looked up after
        key = tuple((o[0], o[1].data_ptr(), o[1].numel() * o[1].element_size(), o[2]) for o in ops)
        args = self._cache.get(key)
        if args is None:
            if len(self._cache) > 256:
                self._cache.clear()
            args = self._cache[key] = ((ct.c_int * n)(*[1 if k[0] else 0 for k in key]), (ct.c_void_p * n)(*[k[1] for k in key]),
                                       (ct.c_int64 * n)(*[k[2] for k in key]), (ct.c_int * n)(*[k[3] for k in key]))
        rc = L.lib().ndwt_comm_exchange(self._h, n, args[0], args[1], args[2], args[3],
                                        ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != L.NDWT_OK:
            raise RuntimeError("RCCL exchange failed: " + L.lib().ndwt_comm_last_error().decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib().ndwt_comm_destroy(self._h)
            self._h = self._ct.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SKEW_BYTES = 256


def _skewed_empty(shape, like):
    """uninitialised tensor whose first element sits 256 bytes past its allocation's (power-of-two) alignment: the approximation
    planes then differ from the packed detail bands of a level in their address bits below 1 KiB"""
    n = 1
    for v in shape:
        n *= int(v)
    skew = _SKEW_BYTES // like.element_size()
    flat = torch.empty(n + skew, dtype=like.dtype, device=like.device)
    return flat[skew:].view(tuple(shape))


def _pitched_bands(nbands, band_shape, like):
    """(nbands, *band_shape) tensor whose bands are each contiguous and `prod(band_shape)` + 256 bytes apart (a view of one
    allocation): the 2^d band streams of a level no longer share their address bits below 1 KiB.  .contiguous() packs it."""
    vol = 1
    for v in band_shape:
        vol *= int(v)
    pitch = vol + _SKEW_BYTES // like.element_size()
    flat = torch.empty(nbands * pitch, dtype=like.dtype, device=like.device)
    strides = [1] * len(band_shape)
    for i in range(len(band_shape) - 2, -1, -1):
        strides[i] = strides[i + 1] * int(band_shape[i + 1])
    return flat.as_strided((nbands,) + tuple(band_shape), (pitch,) + tuple(strides))


def _bands_in_place(y):
    """True when every band of y is contiguous (packed or pitched): the engines take band pointers"""
    return y.dim() >= 2 and y[0].is_contiguous() and (y.shape[0] == 1 or y.stride(0) >= y[0].numel())


class ShardedNdDwt:
    def __init__(self, wname, sizes, pres_l2_norm=False, precision="double", dilation="reference", group=None, device=None,
                 engine=None, synthesis_scheme="auto", overlap="auto", band_pitch="auto", two_streams=False, transport="torch", _self_p2p=False):
        """band_pitch: layout of the coefficient slab dec() returns on the GPU -- 'auto' (default): the bands of one allocation, each
        contiguous, prod(local shape) + 256 bytes apart (a strided view: index it like any tensor; .contiguous() packs it; rec()
        takes either) -- the layout the synthesis kernels read 10 % faster; 'packed': a contiguous tensor, for
        callers that hand the result to collectives, .view(-1) or raw pointers."""
        if band_pitch not in ("auto", "packed"):
            raise ValueError("band_pitch must be 'auto' or 'packed'")
        self.band_pitch = band_pitch
        # test hook: segments a rank owes ITSELF (the periodic wrap inside its own slab) go through the same grouped send / receive as
        # the ones between ranks instead of a local copy -- a 1-rank `nccl` group then runs the whole RCCL branch (in-place halo
        # receives, scatter_recv buffers, work.wait() ordering) on one GPU (tests/test_gpu_parity.py)
        self._self_p2p = bool(_self_p2p)
        self.sizes = [int(s) for s in sizes]
        self.d = len(self.sizes)
        self.wname = [wname] * self.d if isinstance(wname, str) else list(wname)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_outer = self.sizes[-1]
        self.parts = partition(self.n_outer, self.world)
        self.z0, self.z1 = self.parts[self.rank]
        self.n_local = self.z1 - self.z0
        if min(hi - lo for lo, hi in self.parts) < 1:
            raise ValueError("every rank needs at least one plane of the outer axis")
        self.dtype = torch.float32 if precision == "single" else torch.float64
        self.dilation = dilation
        self.device = device if device is not None else torch.device("cpu")
        self.inner_shape = tuple(reversed(self.sizes[:-1]))          # kernel order of the unsharded axes
        local_dims = self.sizes[:-1] + [self.n_local]
        self.engine = engine if engine is not None else HipSlabEngine(self.wname, local_dims, self.dtype, pres_l2_norm,
                                                                     dilation, self.device, global_outer=self.n_outer)
        self.plan = getattr(self.engine, "plan", None)
        if synthesis_scheme == "auto":
            synthesis_scheme = "scatter" if getattr(self.engine, "supports_scatter", False) else "gather"
        self.scheme = synthesis_scheme
        self.nb = 1 << self.d
        self._exchange_cache = {}
        self._bufs = {}                         # scratch tensors reused across calls (halo margins, partial sums, receive buffers)
        # gloo moves host memory only: GPU slabs exchanged over gloo (debugging / rehearsing ranks that share one GPU) are
        # staged through host copies.  RCCL ("nccl") sends the device buffers as they are.
        self._host_stage = bool(self.device.type == "cuda" and dist.is_initialized() and dist.get_backend(group) == "gloo")
        # overlap of the exchange with the planes that do not depend on it needs the run-of-planes entry points
        self.can_overlap = bool(self.scheme == "scatter" and hasattr(self.engine, "analysis_run")
                                and hasattr(self.engine, "synthesis_part") and getattr(self.engine, "supports_overlap", True))
        # overlap: True / False, or "auto" = overlapped until tune() has measured both schedules on this machine.  The pieces cost
        # (a slab level cut into interior + ends pays the (L-1)-plane march prologue twice more: cfg3's 64-plane slab 0.86 -> 0.97 ms per
        # dec+rec on one MI355X), so they pay only where an exchange takes longer than that -- which depends on the fabric, not on us.
        if overlap not in (True, False, "auto"):
            raise ValueError("overlap must be True, False or 'auto'")
        self.overlap_mode = overlap
        self.overlap = bool(overlap) and self.can_overlap
        # the pieces of an overlapped level are independent of each other (the ends wait for the exchange, the interior does not): on the
        # GPU the edge pieces and the exchange they feed / wait for go to a high-priority side stream, the big piece stays on the
        # caller's stream, and the two meet at the end of the level -- the small launch then hides in the big one's ramp and tail
        # instead of adding its own (one MI355X, cfg3's slab: see DESIGN.md section 5)
        self.two_streams = bool(two_streams) and self.device.type == "cuda"
        self._side = None
        # transport of the exchange between ranks: "torch" = torch.distributed point-to-point ops (RCCL on the GPUs, its own stream);
        # "rccl" = RCCL calls on the current stream (DirectRccl; GPU slabs in an initialised process group, not over gloo staging);
        # tune() measures both
        if transport not in ("torch", "rccl"):
            raise ValueError("transport must be 'torch' or 'rccl'")
        self.transport = "torch"
        self._comm = None
        if transport == "rccl":
            self._open_direct()                           # raises if any rank cannot
            self.transport = "rccl"
        self.tuned = None                       # tune(): {"schedule": chosen, "ms_<schedule>": t, ...}

    # ---------------------------------------------------------------------------------- plumbing
    def _owner(self, g):
        for r, (lo, hi) in enumerate(self.parts):
            if lo <= g < hi:
                return r
        raise AssertionError(g)

    def _buf(self, name, shape, like):
        """scratch tensor `name` of this shape (dtype / device of `like`), allocated once and reused by later calls.  Only
        intermediates live here: what dec() / rec() return is always a fresh tensor."""
        shape = tuple(int(v) for v in shape)
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != shape or t.dtype != like.dtype or t.device != like.device:
            t = _skewed_empty(shape, like)
            self._bufs[name] = t
        return t

    def _plan_exchange(self, before, after):
        """Who needs which planes.  Returns contiguous segments (dst_rank, side, src_rank, k0, l0, count) in a
        rank-independent order: `count` planes starting at local plane l0 of src go to position k0 of dst's
        before- (side 0) or after- (side 1) halo.  Cached per (before, after)."""
        key = (before, after)
        segs = self._exchange_cache.get(key)
        if segs is not None:
            return segs
        segs = []
        for q, (lo, hi) in enumerate(self.parts):
            for side, start, n in ((0, lo - before, before), (1, hi, after)):
                k = 0
                while k < n:
                    g = (start + k) % self.n_outer
                    p = self._owner(g)
                    run = min(n - k, self.parts[p][1] - g)          # contiguous inside the owner's slab
                    segs.append((q, side, p, k, g - self.parts[p][0], run))
                    k += run
        self._exchange_cache[key] = segs
        return segs

    def _fetch_halo(self, t, ax, before, after):
        """t: local tensor whose dim `ax` (0 or 1) is the sharded axis.  Returns (halo_before, halo_after).
        Segments are contiguous plane ranges: with ax == 0 they are sent / received in place (no staging copies)."""
        hb, ha, pending = self._start_fetch_halo(t, ax, before, after)
        self._finish_exchange(pending)
        return hb, ha

    def _finish_exchange(self, pending):
        works, post, _keep = pending
        for w in works:
            w.wait()
        for fn in post:
            fn()

    def _start_fetch_halo(self, t, ax, before, after, hb=None, ha=None):
        """Posts the exchange and returns (halo_before, halo_after, pending); the halos are valid after
        _finish_exchange(pending).  Work that does not read them may be issued in between.  hb / ha: optional
        destination buffers (the margins of a [halo | slab | halo] buffer)."""
        shp = list(t.shape)
        if hb is None:
            hb = t.new_empty(shp[:ax] + [before] + shp[ax + 1:])
        if ha is None:
            ha = t.new_empty(shp[:ax] + [after] + shp[ax + 1:])
        ops, keep, post, local = [], [], [], []
        direct = self.transport == "rccl"
        for q, side, p, k0, l0, n in self._plan_exchange(before, after):
            if p != self.rank and q != self.rank:
                continue
            dst = (hb if side == 0 else ha).narrow(ax, k0, n) if q == self.rank else None
            src = t.narrow(ax, l0, n) if p == self.rank else None
            if p == self.rank and q == self.rank and not self._self_p2p:   # own planes (periodic wrap inside the slab)
                local.append((dst, src))
                continue
            if p == self.rank:
                buf = src if src.is_contiguous() else src.contiguous()
                if self._host_stage:
                    buf = buf.cpu()
                keep.append(buf)
                ops.append((True, buf, q) if direct else dist.P2POp(dist.isend, buf, self._global_rank(q), self.group))
            if q == self.rank:
                if dst.is_contiguous() and not self._host_stage:
                    ops.append((False, dst, p) if direct else dist.P2POp(dist.irecv, dst, self._global_rank(p), self.group))
                else:
                    buf = torch.empty(dst.shape, dtype=dst.dtype, device="cpu" if self._host_stage else dst.device)
                    ops.append((False, buf, p) if direct else dist.P2POp(dist.irecv, buf, self._global_rank(p), self.group))
                    post.append(lambda dst=dst, buf=buf: dst.copy_(buf))
        self._apply(False, local)
        if direct:
            self._comm.exchange(ops)                      # in stream order on the current stream: nothing to wait for afterwards
            return hb, ha, ([], post, keep)
        works = dist.batch_isend_irecv(ops) if ops else []
        return hb, ha, (works, post, keep)

    def _scatter_add(self, ext, before, after):
        """ext: (before + n_local + after, ...) partial sums; planes outside the slab go to their owners and are added.
        Returns the completed local slab (a view of ext)."""
        own = ext.narrow(0, before, self.n_local)
        pending = self._start_scatter(ext.narrow(0, 0, before), ext.narrow(0, before + self.n_local, after), before, after)
        self._finish_scatter(pending, own)
        return own

    def _start_scatter(self, part_before, part_after, before, after):
        """Posts the sends of the partial planes this rank owes (part_before: for the `before` planes ahead of its
        slab, part_after: for the `after` planes behind it) and the receives of what it is owed."""
        ops, adds_local, adds_recv, keep = [], [], [], []
        direct = self.transport == "rccl"
        parts = (part_before, part_after)
        # rank q PRODUCES partial planes for the global planes around its slab; the owner p ADDS them
        for q, side, p, k0, l0, n in self._plan_exchange(before, after):
            if q == self.rank:
                part = parts[side].narrow(0, k0, n)                 # contiguous view
                if p == self.rank and not self._self_p2p:
                    adds_local.append((l0, n, part))
                    continue
                if self._host_stage:
                    part = part.cpu()
                    keep.append(part)
                ops.append((True, part, p) if direct else dist.P2POp(dist.isend, part, self._global_rank(p), self.group))
            if p == self.rank:
                ref = parts[side]
                if self._host_stage:
                    buf = torch.empty([n] + list(ref.shape[1:]), dtype=ref.dtype, device="cpu")
                else:
                    buf = self._buf(("scatter_recv", q, side, k0), [n] + list(ref.shape[1:]), ref)
                ops.append((False, buf, q) if direct else dist.P2POp(dist.irecv, buf, self._global_rank(q), self.group))
                adds_recv.append((l0, n, buf))
        if direct:
            self._comm.exchange(ops)
            return [], adds_local + adds_recv, (parts, keep)
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, adds_local + adds_recv, (parts, keep)

    def _finish_scatter(self, pending, own):
        works, adds, _keep = pending
        for w in works:
            w.wait()
        self._apply(True, [(own.narrow(0, l0, n), buf if buf.device == own.device else buf.to(own.device)) for l0, n, buf in adds])

    def _apply(self, add, pairs):
        """dst = src / dst += src for every pair; engines with a multi-run kernel take all contiguous pairs in one launch"""
        if not pairs:
            return
        seg = getattr(self.engine, "segments", None)
        if seg is not None and pairs[0][0].is_cuda and all(d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype for d, s in pairs):
            # one launch per batch of runs whose destinations are disjoint: a slab thinner than the halo gets several addends for the
            # same planes (multi-hop exchange), and those must be added one after the other
            batch, spans = [], []
            for d, s_ in pairs:
                lo = d.data_ptr()
                hi = lo + d.numel() * d.element_size()
                if any(lo < h and l < hi for l, h in spans):
                    seg(add, batch)
                    batch, spans = [], []
                batch.append((d, s_))
                spans.append((lo, hi))
            seg(add, batch)
            return
        for d, s in pairs:
            if add:
                d.add_(s)
            else:
                d.copy_(s)

    def _one_stream(self):
        """The driver's scratch buffers (halo margins, partial sums, receive buffers) are reused across calls and every hazard on
        them is ordered THROUGH THE CURRENT STREAM (kernels in stream order; `work.wait()` of every send and receive makes the
        current stream wait for the communication stream; posting an exchange makes the communication stream wait for the current
        one).  A call on another stream than the previous one therefore first waits for everything queued on the old stream."""
        if self.device.type != "cuda":
            return
        cur = torch.cuda.current_stream(self.device)
        last = getattr(self, "_last_stream", None)
        if last is not None and last != cur:
            cur.wait_event(last.record_event())
        self._last_stream = cur

    def _open_direct(self):
        """the direct RCCL transport, created once (collective); raises RuntimeError on every rank if one of them cannot"""
        if self._comm is None:
            if self.device.type != "cuda" or not dist.is_initialized() or self._host_stage:
                raise RuntimeError("the direct RCCL transport needs GPU slabs in an initialised (non-gloo-staged) process group")
            self._comm = DirectRccl(self.group, self.device)
        return self._comm

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(self.device, priority=-1)
        return self._side

    def _global_rank(self, r):
        return r if self.group is None or self.group is dist.group.WORLD else dist.get_global_rank(self.group, r)

    def _stride(self, lev):
        return 1 if self.dilation == "reference" else 1 << (lev - 1)

    def tune(self, x_local, level, steps=10):
        """Measures dec + rec of this slab under every schedule -- one piece per level (exchange, then one launch), the exchange
        overlapped with the interior planes on one stream, the same with the edge pieces and the exchange on a high-priority side
        stream; on GPUs each with torch.distributed's point-to-point ops and with RCCL calls on the transform's own stream (DirectRccl) --
        and keeps the fastest; the same on every rank (the times are MAX-reduced over the group, so all ranks take the same
        decision from the same numbers).  Collective: every rank of the group calls it with its own slab and the same level.
        Returns the record it stores in self.tuned."""
        import time
        if not self.can_overlap:
            self.tuned = {"schedule": "one_piece", "reason": "engine has no run-of-planes entry points"}
            return self.tuned
        distributed = dist.is_initialized() and self.world > 1
        cuda = self.device.type == "cuda"

        def fence():
            if cuda:
                torch.cuda.synchronize(self.device)
            if distributed:
                dist.barrier(self.group)

        modes = [("one_piece", False, False, "torch"), ("overlap", True, False, "torch")] + ([("overlap_two_streams", True, True, "torch")] if cuda else [])
        note = None
        if cuda and dist.is_initialized() and not self._host_stage and (self.world > 1 or self._self_p2p):
            try:                                          # the exchange as RCCL calls on the transform's own stream (collective: all ranks or none)
                self._open_direct()
                modes += [("rccl_one_piece", False, False, "rccl"), ("rccl_overlap_two_streams", True, True, "rccl")]
            except RuntimeError as exc:
                note = str(exc)[:200]
        ms = []
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()                                      # (a full collection of the interpreter's GC is 40 ms with torch loaded: not inside a timing)
        try:
            for _, ov, ts, tr in modes:
                self.overlap, self.two_streams, self.transport = ov, ts, tr
                self.rec(self.dec(x_local, level))        # buffers, plans, communicators of this schedule
                fence()
                t0 = time.perf_counter()
                for _ in range(steps):
                    self.rec(self.dec(x_local, level))
                fence()
                ms.append((time.perf_counter() - t0) / steps * 1e3)
        finally:
            if gc_was_on:
                gc.enable()
        if distributed:
            red = torch.tensor(ms, dtype=torch.float64, device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(red, op=dist.ReduceOp.MAX, group=self.group)
            ms = [float(v) for v in red]
        best = min(range(len(modes)), key=lambda i: ms[i])
        _, self.overlap, self.two_streams, self.transport = modes[best]
        self.tuned = {"schedule": modes[best][0], "steps": steps, **{f"ms_{m[0]}": round(t, 4) for m, t in zip(modes, ms)}}
        if note:
            self.tuned["direct_rccl"] = note
        return self.tuned

    # --------------------------------------------------------------------------------- transform
    def dec(self, x_local, level):
        """x_local: (n_local, ..., n1) -> (bands, n_local, ..., n1); band order of the reference (nddwt.c:210).

        The approximation band of every level lives in a scratch buffer [halo_before | slab | halo_after] whose margins
        receive the neighbours' planes in place, so no haloed copy is assembled (the first level copies x once; engines
        with the split-halo entry point read x and the two received halo buffers from where they are)."""
        nb, nbt = self.nb, self.nb + (self.nb - 1) * (level - 1)
        self._one_stream()
        x_local = x_local.to(self.dtype).contiguous()
        # the local coefficient slab is this driver's own array: bands pitched (a strided view; .contiguous() packs it)
        y = (_pitched_bands(nbt, tuple(x_local.shape), x_local) if x_local.is_cuda and self.band_pitch == "auto"
             else x_local.new_empty((nbt,) + tuple(x_local.shape)))
        n, inner = self.n_local, tuple(x_local.shape[1:])
        split = bool(hasattr(self.engine, "analysis_split") and getattr(self.engine, "supports_split",
                                                                     getattr(self.engine, "supports_scatter", False)))
        cur, cur_buf = x_local, None          # cur_buf: the [halo | cur | halo] buffer cur lives in (None: bare tensor)
        for lev in range(1, level + 1):
            s = self._stride(lev)
            ab, aa, _, _ = self.engine.halo(s)
            overlap = self.overlap and n > ab + aa
            if cur_buf is None and not split:
                # engines that need the haloed slab contiguous: the level's input moves into a margin buffer once
                cur_buf = self._buf(("ana_in", lev & 1), (ab + n + aa,) + inner, x_local)
                cur_buf[ab:ab + n].copy_(cur)
                cur = cur_buf[ab:ab + n]
            # where this level's approximation goes: the final band 0, or the margin buffer of the next level
            a_buf = None
            if lev == level:
                a_out = y[0]
            elif split and not (overlap and hasattr(self.engine, "analysis_ends")):
                a_out = self._buf(("ana_plain", lev & 1), (n,) + inner, x_local)
            else:
                ab2, aa2 = self.engine.halo(self._stride(lev + 1))[:2]
                a_buf = self._buf(("ana_in", (lev + 1) & 1), (ab2 + n + aa2,) + inner, x_local)
                a_out = a_buf[ab2:ab2 + n]
            outs = [a_out] + [y[1 + (nb - 1) * (level - lev) + (b - 1)] for b in range(1, nb)]
            if cur_buf is not None:
                hb_dst, ha_dst = cur_buf[:ab], cur_buf[ab + n:]
            else:
                hb_dst = self._buf(("halo_b", lev & 1), (ab,) + inner, x_local)
                ha_dst = self._buf(("halo_a", lev & 1), (aa,) + inner, x_local)
            if overlap:
                # interior planes [ab, n-aa) read the slab only: run them while the halo planes travel
                def ends(pending, hb, ha):
                    self._finish_exchange(pending)
                    if cur_buf is not None and hasattr(self.engine, "analysis_ends"):
                        self.engine.analysis_ends(cur_buf, outs, s)           # both ends, one launch
                    else:
                        if ab:
                            self.engine.analysis_run(cur, hb, ha, outs, 0, ab, s)
                        self.engine.analysis_run(cur, hb, ha, outs, n - aa, n, s)
                if self.two_streams:
                    main, side = torch.cuda.current_stream(self.device), self._side_stream()
                    side.wait_stream(main)                                # the level's input is complete
                    with torch.cuda.stream(side):                         # exchange + ends: the side stream (the exchange waits for it)
                        hb, ha, pending = self._start_fetch_halo(cur, 0, ab, aa, hb_dst, ha_dst)
                    self.engine.analysis_run(cur, hb, ha, outs, ab, n - aa, s)
                    with torch.cuda.stream(side):
                        ends(pending, hb, ha)
                    main.wait_stream(side)
                else:
                    hb, ha, pending = self._start_fetch_halo(cur, 0, ab, aa, hb_dst, ha_dst)
                    self.engine.analysis_run(cur, hb, ha, outs, ab, n - aa, s)
                    ends(pending, hb, ha)
            else:
                hb, ha, pending = self._start_fetch_halo(cur, 0, ab, aa, hb_dst, ha_dst)
                self._finish_exchange(pending)
                if cur_buf is not None:
                    self.engine.analysis(cur_buf, outs, s)
                else:
                    self.engine.analysis_split(cur, hb, ha, outs, s)
            cur, cur_buf = a_out, a_buf
        return y

    def rec(self, y):
        """(bands, n_local, ..., n1) -> (n_local, ..., n1)"""
        nb = self.nb
        self._one_stream()
        level = 1 + (y.shape[0] - nb) // (nb - 1)
        y = y.to(self.dtype)
        if not _bands_in_place(y):
            y = y.contiguous()
        prev = y[0]
        for ind in range(1, level + 1):
            lev = level - ind + 1
            s = self._stride(lev)
            _, _, sb, sa = self.engine.halo(s)
            ins = [prev] + [y[1 + (nb - 1) * (level - lev) + (b - 1)] for b in range(1, nb)]
            if self.scheme == "scatter" and self.overlap:
                # the partial sums owed to the neighbours first (sa planes ahead of the slab, sb behind it: they depend
                # on the first / last coefficient planes only), then the slab's own planes while those travel
                n, inner = self.n_local, tuple(prev.shape[1:])

                def send_parts():
                    if hasattr(self.engine, "synthesis_send_parts"):
                        part_b, part_a = self.engine.synthesis_send_parts(ins, s)
                    else:
                        part_b, part_a = prev.new_empty((sa,) + inner), prev.new_empty((sb,) + inner)
                        if sa:
                            self.engine.synthesis_part(ins, 0, part_b, s)
                        self.engine.synthesis_part(ins, sa + n, part_a, s)
                    return self._start_scatter(part_b, part_a, sa, sb)
                own = prev.new_empty((n,) + inner) if lev == 1 else self._buf(("syn_own", lev & 1), (n,) + inner, prev)
                if self.two_streams:
                    main, side = torch.cuda.current_stream(self.device), self._side_stream()
                    side.wait_stream(main)                                # the level's coefficients are complete
                    with torch.cuda.stream(side):                         # partial sums + their exchange: the side stream
                        pending = send_parts()
                    self.engine.synthesis_part(ins, sa, own, s)           # the slab's own planes meanwhile
                    main.wait_stream(side)                                # (the partial sums this rank owes itself)
                else:
                    pending = send_parts()
                    self.engine.synthesis_part(ins, sa, own, s)
                self._finish_scatter(pending, own)
                prev = own
            elif self.scheme == "scatter":
                # zero-extended synthesis: partial sums for sa planes before and sb planes after the slab
                ext_shape = (sa + self.n_local + sb,) + tuple(prev.shape[1:])
                ext = prev.new_empty(ext_shape) if lev == 1 else self._buf(("syn_ext", lev & 1), ext_shape, prev)
                self.engine.synthesis_ext(ins, ext, s)
                prev = self._scatter_add(ext, sa, sb)
            else:
                # gather scheme (engines without the zero-extended synthesis: per-axis kernels, dilated levels): the 2^d bands
                # are assembled with their halo planes in one scratch array, the halos received in place
                n = self.n_local
                full = self._buf(("syn_full",), (nb, sb + n + sa) + tuple(prev.shape[1:]), prev)
                for b in range(nb):
                    full[b, sb:sb + n].copy_(ins[b])
                hb, ha, pending = self._start_fetch_halo(full[:, sb:sb + n], 1, sb, sa, full[:, :sb], full[:, sb + n:])
                self._finish_exchange(pending)
                out = torch.empty_like(prev) if lev == 1 else self._buf(("syn_out", lev & 1), prev.shape, prev)
                self.engine.synthesis([full[b] for b in range(nb)], out, s)
                prev = out
        return prev.contiguous()
