"""Usage example of the nd_dwt_{1,2,3,4}D classes on an MI355X (needs a GPU; there is no CPU path).

The same walk-through as the reference's example_nd_dwt_{1,2,3,4}D.m: a random complex signal, a multilevel
decomposition, the reconstruction, then the energy in both domains (equal with pres_l2_norm) and the
reconstruction error.  python examples/example_nd_dwt.py [1|2|3|4]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ndwt_amd as ndwt  # noqa: E402

CASES = {
    1: (ndwt.nd_dwt_1D, [1024], "db3", 3),
    2: (ndwt.nd_dwt_2D, [256, 256], ["db1", "db4"], 2),
    3: (ndwt.nd_dwt_3D, [64, 64, 20], ["db1", "db3", "db9"], 2),
    4: (ndwt.nd_dwt_4D, [32, 32, 16, 16], ["db1", "db3", "db2", "db4"], 2),
}


def run(d):
    cls, sizes, wnames, level = CASES[d]
    torch.manual_seed(0)
    # column-major memory like a MATLAB array: build it transposed, then view it with the MATLAB shape
    xk = torch.randn(*reversed(sizes), dtype=torch.complex128, device="cuda")
    x = xk.permute(*reversed(range(d)))
    nddwt = cls(wnames, sizes, "pres_l2_norm", True)
    x_trans = nddwt.dec(x, level)                 # [sizes..., bands]
    x_recon = nddwt.rec(x_trans)
    print(f"{d}-D {sizes} {wnames} level {level}: coefficients {list(x_trans.shape)}")
    print(f"  Energy in signal domain = {torch.linalg.vector_norm(x).item():.6f}"
          f"   Energy in wavelet domain = {torch.linalg.vector_norm(x_trans).item():.6f}")
    print(f"  Absolute max reconstruction error = {(x_recon - x).abs().max().item():.3e}")


def extras():
    """not in the reference: a solver-style loop on device tensors -- pitched coefficient tensors and thresholding on the GPU"""
    sizes = [128, 128, 64]
    x = torch.randn(*reversed(sizes), device="cuda").permute(2, 1, 0)
    w = ndwt.nd_dwt_3D("db4", sizes, "pres_l2_norm", True, "precision", "single", "band_pitch", "auto")
    y = w.dec(x, 3)                               # same shape / values as the packed array; bands 256 bytes further apart
    y = w.shrink(y, 0.5, "soft")                  # detail bands only; returns a new (pitched) tensor
    x1 = w.rec(y)
    x2 = w.denoise(x, 3, 0.5, "soft")             # the same in one call, coefficients never leave the plan's scratch
    print(f"3-D {sizes} db4 level 3, pitched coefficients: band stride {y.stride()[-1]} elements for {x.numel()} per band; "
          f"|rec(shrink(dec(x))) - denoise(x)| = {(x1 - x2).abs().max().item():.1e}")


if __name__ == "__main__":
    for d in ([int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]):
        run(d)
    if len(sys.argv) == 1:
        extras()
