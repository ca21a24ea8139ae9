"""Multi-rank slab decomposition on CPU (gloo): the halo-exchange / scatter-add logic of
non-decimated_wavelets_amd/sharded.py with the oracle standing in for the per-GPU engine.

The product engine is HipSlabEngine (HIP kernels); injecting an oracle-backed engine here is what lets the
N>1 path be exercised without GPUs.  Gate: the sharded result equals the single-process oracle
(periodic wrap across rank G-1 -> 0, uneven shards).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleSlabEngine:
    """Slab compute with explicit halos, from the periodic oracle: transform the haloed slab as if periodic and crop
    (the filter support never reaches the wrap inside the cropped region)."""

    supports_scatter = True

    def __init__(self, wname, l2):
        import ndwt_oracle as orc
        self.orc = orc
        self.filt = [orc.wave_filters(w) for w in wname]
        self.l2 = l2
        self.L = len(self.filt[-1][0])

    def halo(self, stride):
        L = self.L
        return ((L // 2 - 1) * stride, (L // 2) * stride, (L // 2) * stride, (L // 2 - 1) * stride)

    def _to_mat(self, t):            # kernel order (outer first) -> MATLAB shape
        return np.transpose(t.numpy())

    def analysis(self, in_with_halo, outs, stride):
        ab, aa, _, _ = self.halo(stride)
        y = self.orc.spatial_level_dec(self._to_mat(in_with_halo), self.filt, self.l2, stride)
        n = in_with_halo.shape[0] - ab - aa
        for b, o in enumerate(outs):
            o.copy_(torch.from_numpy(np.ascontiguousarray(np.transpose(y[..., b])))[ab:ab + n])

    def analysis_split(self, in_local, hb, ha, outs, stride):
        self.analysis(torch.cat([hb, in_local, ha], 0), outs, stride)

    def synthesis(self, ins_with_halo, out, stride):
        _, _, sb, sa = self.halo(stride)
        c = np.stack([self._to_mat(t) for t in ins_with_halo], axis=-1)
        r = self.orc.spatial_level_rec(c, self.filt, self.l2, stride)
        out.copy_(torch.from_numpy(np.ascontiguousarray(np.transpose(r)))[sb:sb + out.shape[0]])

    def synthesis_ext(self, ins_local, out_ext, stride):
        # zero-extended synthesis: partial sums for sa planes before and sb planes after the slab
        _, _, sb, sa = self.halo(stride)
        pad = (self.L - 1) * stride
        padded = [torch.nn.functional.pad(t, [0, 0] * (t.dim() - 1) + [pad, pad]) for t in ins_local]
        c = np.stack([self._to_mat(t) for t in padded], axis=-1)
        r = torch.from_numpy(np.ascontiguousarray(np.transpose(self.orc.spatial_level_rec(c, self.filt, self.l2, stride))))
        n = ins_local[0].shape[0]
        out_ext.copy_(r[pad - sa: pad + n + sb])


    def analysis_run(self, cur, hb, ha, outs, z0, z1, stride):
        ab, aa, _, _ = self.halo(stride)
        n = cur.shape[0]
        assert (z0 == 0 or z0 >= ab) and (z1 == n or z1 + aa <= n)
        before = hb if z0 == 0 else cur[z0 - ab:z0]
        after = ha if z1 == n else cur[z1:z1 + aa]
        self.analysis(torch.cat([before, cur[z0:z1], after], 0), [o[z0:z1] for o in outs], stride)

    def analysis_ends(self, slab_buf, outs, stride):
        ab, aa, _, _ = self.halo(stride)
        n, m = slab_buf.shape[0] - ab - aa, max(ab, aa)
        for z0 in (0, n - m):
            self.analysis(slab_buf[z0:z0 + ab + m + aa], [o[z0:z0 + m] for o in outs], stride)

    def synthesis_send_parts(self, ins_local, stride):
        _, _, sb, sa = self.halo(stride)
        n = ins_local[0].shape[0]
        ext = ins_local[0].new_empty((sa + n + sb,) + tuple(ins_local[0].shape[1:]))
        self.synthesis_ext(ins_local, ext, stride)
        return ext[:sa].clone(), ext[sa + n:].clone()

    def synthesis_part(self, ins_local, e0, out_run, stride):
        _, _, sb, sa = self.halo(stride)
        ext = ins_local[0].new_empty((sa + ins_local[0].shape[0] + sb,) + tuple(ins_local[0].shape[1:]))
        self.synthesis_ext(ins_local, ext, stride)
        out_run.copy_(ext[e0:e0 + out_run.shape[0]])


def _worker(rank, world, port, sizes, wname, level, l2, dilation, scheme, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import ndwt_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = importlib.import_module("non-decimated_wavelets_amd.sharded")
        rng = np.random.default_rng(7)
        x = rng.standard_normal(sizes)                         # MATLAB shape, same on every rank
        xk = torch.from_numpy(np.ascontiguousarray(np.transpose(x)))
        want = np.ascontiguousarray(np.transpose(orc.spatial_dec(x, wname, level, l2, dilation)))   # (bands, n_d, ..., n1)
        c = rng.standard_normal(list(sizes) + [orc.num_bands(len(sizes), level)])
        ck = torch.from_numpy(np.ascontiguousarray(np.transpose(c)))
        want2 = np.ascontiguousarray(np.transpose(orc.spatial_rec(c, wname, l2, dilation)))
        e_dec = e_rec = e_rec2 = 0.0
        # the scatter scheme runs twice: exchange overlapped with the interior planes (default), and one piece
        for overlap in ((True, False) if scheme == "scatter" else (False,)):
            eng = sh.ShardedNdDwt(wname, sizes, pres_l2_norm=l2, precision="double", dilation=dilation,
                                  engine=OracleSlabEngine(wname, l2), synthesis_scheme=scheme, overlap=overlap)
            assert eng.overlap == overlap
            y_loc = eng.dec(xk[eng.z0:eng.z1].contiguous(), level)
            e_dec = max(e_dec, float(np.abs(y_loc.numpy() - want[:, eng.z0:eng.z1]).max()))
            r_loc = eng.rec(y_loc)
            e_rec = max(e_rec, float(np.abs(r_loc.numpy() - xk[eng.z0:eng.z1].numpy()).max()))
            # rec of arbitrary coefficients (not in the range of dec)
            r2 = eng.rec(ck[:, eng.z0:eng.z1].contiguous())
            e_rec2 = max(e_rec2, float(np.abs(r2.numpy() - want2[eng.z0:eng.z1]).max()))
        if scheme == "scatter":
            # overlap="auto" (the default): tune() times both schedules, MAX-reduces the times and every rank keeps the same one
            eng = sh.ShardedNdDwt(wname, sizes, pres_l2_norm=l2, precision="double", dilation=dilation,
                                  engine=OracleSlabEngine(wname, l2), synthesis_scheme=scheme)
            assert eng.overlap_mode == "auto" and eng.overlap and eng.tuned is None
            rec = eng.tune(xk[eng.z0:eng.z1].contiguous(), level, steps=1)
            assert (rec["schedule"] == "overlap") == eng.overlap and rec["ms_overlap"] > 0 and rec["ms_one_piece"] > 0
            votes = [None] * world
            dist.all_gather_object(votes, (rec["schedule"], rec["ms_overlap"], rec["ms_one_piece"]))
            assert len(set(votes)) == 1, votes                 # one decision from one set of numbers
            y_loc = eng.dec(xk[eng.z0:eng.z1].contiguous(), level)
            e_dec = max(e_dec, float(np.abs(y_loc.numpy() - want[:, eng.z0:eng.z1]).max()))
        q.put((rank, e_dec, e_rec, e_rec2))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [
    # world, sizes,          wavelets,                 level, l2, dilation,    scheme
    (2, [12, 10, 16], ["db2", "db1", "db3"], 2, 1, "reference", "gather"),
    (2, [12, 10, 16], ["db2", "db1", "db3"], 2, 0, "reference", "scatter"),
    (3, [10, 9, 17], ["db1", "db2", "db4"], 2, 1, "reference", "scatter"),     # uneven shards, halo wider than a neighbour
    (2, [8, 6, 7, 12], ["db1", "db2", "db1", "db2"], 2, 1, "reference", "scatter"),   # 4-D, t-sharded
    (2, [16, 12], ["db3", "db2"], 3, 0, "atrous", "gather"),                   # 2-D, dilated taps: multi-plane halos
    (4, [8, 8, 12], ["db2", "db2", "db3"], 1, 1, "reference", "scatter"),      # 3 planes per rank < halo of db3: multi-hop
]


@pytest.mark.parametrize("world,sizes,wname,level,l2,dilation,scheme", CASES)
def test_sharded_equals_single_process(world, sizes, wname, level, l2, dilation, scheme):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sizes, wname, level, l2, dilation, scheme, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, e_dec, e_rec, e_rec2 in res:
        assert e_dec < 1e-12 and e_rec < 1e-12 and e_rec2 < 1e-12, (rank, e_dec, e_rec, e_rec2)
