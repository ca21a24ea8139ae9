"""Every BASELINE.json configuration at FULL size against the CPU oracle (-m gpu).

The oracle cannot transform 512^3 or 256^3x32 voxels in seconds, so each configuration is pinned at full size through two
inputs whose full-size transform follows exactly from small oracle runs (stride-1 periodic filtering is linear and
shift-invariant):

  * tile-periodic: x = a small block repeated along every axis.  dec(tile(x)) == tile(dec(x)) and
    rec(tile(c)) == tile(rec(c)), so `orc.spatial_dec / spatial_rec` on the block gives the expected value of EVERY
    voxel of every band.  Blocks are larger than one workgroup tile along x and y (128 x 64 against 64 x 16 / 64 x 32
    tiles), so neighbouring tiles hold different data, and the periodic seam of every z-chunk / t-frame is checked.
  * separable: x = sum_r a_r (x) b_r (x) c_r with random vectors of the full axis lengths.  Every band of every level is
    then the same sum of outer products of 1-D transforms (`orc.spatial_dec` on the vectors), computed in fp64 on the
    device.  No two voxels, tiles, z-chunks or batch items see the same data, 64-bit offsets included.  The inverse is
    checked the same way with separable coefficients in every band.

Tolerance: fp32, 2e-6 relative to max|c| for coefficients; reconstructions of random
coefficients 4e-6 relative to max|r| (the sum of up to 46 bands).  Reference: Test/nddwt3D_test.m:25-27 (what the
reference's own scripts print), nd_dwt_4D.m:137-195.
"""
import json
import os
import zlib

import numpy as np
import pytest
import torch

import ndwt_amd as ndwt
import ndwt_oracle as orc

pytestmark = pytest.mark.gpu

TOL_DEC, TOL_REC = 2e-6, 4e-6
CLS = {2: ndwt.nd_dwt_2D, 3: ndwt.nd_dwt_3D, 4: ndwt.nd_dwt_4D}

# name: (sizes [n1..nd], wavelet, levels, tile-periodic block [b1..bd])
CONFIGS = {
    "cfg2_2d_4096sq_db4_L3": ([4096, 4096], "db4", 3, [256, 128]),
    "cfg3_3d_512cube_db4_L3": ([512, 512, 512], "db4", 3, [128, 64, 64]),
    "cfg4_3d_512cube_db6_L4": ([512, 512, 512], "db6", 4, [128, 64, 64]),
    "cfg5_4d_256cube_x32_db4_L3": ([256, 256, 256, 32], "db4", 3, [64, 64, 32, 16]),
}
_REPORT = {}


def _kernel(a):
    """MATLAB-shaped numpy array [n1..nd(,bands)] -> (bands,) nd..n1 contiguous"""
    return np.ascontiguousarray(np.transpose(a))


def _tiled_view(t, block_shape):
    """contiguous (nd..n1) tensor -> view (nd/bd, bd, ..., n1/b1, b1)"""
    shape = []
    for n, b in zip(t.shape, block_shape):
        assert n % b == 0
        shape += [n // b, b]
    return t.view(shape)


def _bcast(block):
    return block.view([s for b in block.shape for s in (1, b)])


def _max_err_vs_tiled(vol, block):
    return float((_tiled_view(vol, block.shape) - _bcast(block)).abs().max())


def _outer(vecs):
    """vecs: 1-D fp64 tensors in kernel order (outermost axis first) -> their outer product"""
    out = vecs[0]
    for v in vecs[1:]:
        out = out.unsqueeze(-1) * v
    return out


def _band_axis_vectors_dec(vec, wname, level, l2):
    """1-D transforms of one axis vector: {(lev, bit): vector} -- bit 0: lo^lev(v), bit 1: hi(lo^(lev-1)(v))"""
    out = {}
    for lev in range(1, level + 1):
        y = orc.spatial_dec(vec, wname, lev, l2)                 # [n, 1 + lev]: band 0 = A_lev, band 1 = D_lev
        out[(lev, 0)] = y[:, 0]
        out[(lev, 1)] = y[:, 1]
    return out


def _syn1d(v, bit, wname, l2):
    """one 1-D synthesis step of `v` fed into the low (bit 0) or high (bit 1) channel, the other channel zero"""
    c = np.zeros((len(v), 2))
    c[:, bit] = v
    return orc.spatial_rec(c, wname, l2)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_config_against_the_oracle(name):
    sizes, wname, level, block = CONFIGS[name]
    d = len(sizes)
    nb, nbt = 1 << d, orc.num_bands(d, level)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    rep = {}

    # ------------------------------------------------------------------ tile-periodic, pres_l2_norm on
    l2 = 1
    xb = rng.standard_normal(block)
    want_y = _kernel(orc.spatial_dec(xb, wname, level, l2))                       # (bands, bd..b1)
    cb = rng.standard_normal(list(block) + [nbt])
    want_r = _kernel(orc.spatial_rec(cb, wname, l2))
    w = CLS[d](wname, sizes, "pres_l2_norm", l2, "precision", "single")
    reps = [n // b for n, b in zip(reversed(sizes), reversed(block))]
    xk = torch.from_numpy(_kernel(xb)).to(dev, torch.float32).repeat(*reps)
    assert list(xk.shape) == list(reversed(sizes))
    y = w.dec(xk.permute(*reversed(range(d))), level)                              # MATLAB shape [n1..nd, bands]
    assert tuple(y.shape) == tuple(sizes) + (nbt,)
    yk = y.permute(*reversed(range(d + 1)))                                        # (bands, nd..n1), contiguous
    assert yk.is_contiguous()
    scale = float(np.abs(want_y).max())
    e_dec = 0.0
    for b in range(nbt):
        e_dec = max(e_dec, _max_err_vs_tiled(yk[b], torch.from_numpy(want_y[b]).to(dev, torch.float32)))
    rep["tile_dec_rel_err"] = e_dec / scale
    assert e_dec <= TOL_DEC * scale, (name, e_dec / scale)
    # inverse of tile-periodic coefficients, written over the same buffer
    ck = _kernel(cb)
    for b in range(nbt):
        _tiled_view(yk[b], list(reversed(block))).copy_(_bcast(torch.from_numpy(ck[b]).to(dev, torch.float32)))
    r = w.rec(y)
    rk = r.permute(*reversed(range(d)))
    e_rec = _max_err_vs_tiled(rk, torch.from_numpy(want_r).to(dev, torch.float32))
    rep["tile_rec_rel_err"] = e_rec / float(np.abs(want_r).max())
    assert e_rec <= TOL_REC * float(np.abs(want_r).max()), (name, rep)
    del w, r, rk, xk, y, yk
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ separable, pres_l2_norm off
    l2 = 0
    R = 2
    w = CLS[d](wname, sizes, "pres_l2_norm", l2, "precision", "single")
    vecs = [[rng.standard_normal(sizes[a]) for a in range(d)] for _ in range(R)]     # [r][axis a]
    t64 = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev, torch.float64)
    x64 = sum(_outer([t64(vecs[r][a]) for a in reversed(range(d))]) for r in range(R))
    xk = x64.to(torch.float32)
    del x64
    tabs = [[_band_axis_vectors_dec(vecs[r][a], wname, level, l2) for a in range(d)] for r in range(R)]
    y = w.dec(xk.permute(*reversed(range(d))), level)
    yk2 = y.permute(*reversed(range(d + 1)))
    e_dec, scale = 0.0, 0.0
    for lev in range(1, level + 1):
        for bits in range(nb):
            if bits == 0 and lev != level:
                continue                                                           # only the coarsest approximation is kept
            slot = 0 if bits == 0 else 1 + (nb - 1) * (level - lev) + (bits - 1)
            want = sum(_outer([t64(tabs[r][a][(lev, (bits >> a) & 1)]) for a in reversed(range(d))]) for r in range(R))
            scale = max(scale, float(want.abs().max()))
            e_dec = max(e_dec, float((yk2[slot].double() - want).abs().max()))
            del want
    rep["separable_dec_rel_err"] = e_dec / scale
    assert e_dec <= TOL_DEC * scale, (name, rep)
    del xk
    # inverse: separable random coefficients in every band; expected = sum over bands of outer products of the 1-D
    # synthesis chains (band of level lev: its own filter at that level, then the low channel of levels lev-1 .. 1)
    want = None
    for lev in range(1, level + 1):
        for bits in range(nb):
            if bits == 0 and lev != level:
                continue
            slot = 0 if bits == 0 else 1 + (nb - 1) * (level - lev) + (bits - 1)
            cv = [rng.standard_normal(sizes[a]) for a in range(d)]
            yk2[slot].copy_(_outer([t64(cv[a]) for a in reversed(range(d))]))
            fv = []
            for a in range(d):
                v = _syn1d(cv[a], (bits >> a) & 1, wname, l2)
                for _ in range(lev - 1):
                    v = _syn1d(v, 0, wname, l2)
                fv.append(v)
            term = _outer([t64(fv[a]) for a in reversed(range(d))])
            want = term if want is None else want.add_(term)
            del term
    r = w.rec(y)
    rk = r.permute(*reversed(range(d)))
    e_rec = float((rk.double() - want).abs().max())
    rep["separable_rec_rel_err"] = e_rec / float(want.abs().max())
    assert e_rec <= TOL_REC * float(want.abs().max()), (name, rep)
    rep["coefficient_bytes"] = int(yk2.numel() * 4)
    _REPORT[name] = rep
    del w, y, yk2, r, rk, want
    torch.cuda.empty_cache()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fullsize_parity.json"), "w") as f:
            json.dump(_REPORT, f, indent=1)
    except OSError:
        pass
