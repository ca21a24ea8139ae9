"""The C-ABI shared library: loads, exports every symbol include/ndwt.h declares, host-side logic (no GPU).

No compute call is made here -- on a machine without a GPU the library must fail loudly, never fall back.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import ndwt_amd as ndwt
import ndwt_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "ndwt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ndwt_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ndwt.lib()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ndwt.h but not exported"


@pytest.mark.parametrize("K", range(1, 11))
def test_wave_filters_through_the_abi_match_the_oracle(K):
    lo, hi = ndwt.wave_filters(f"db{K}")
    olo, ohi = orc.wave_filters(f"DB{K}")
    np.testing.assert_allclose(lo, olo, rtol=0, atol=1e-16)
    np.testing.assert_allclose(hi, ohi, rtol=0, atol=1e-16)


def test_unknown_wavelet_is_an_error():
    with pytest.raises(ndwt.NdwtError, match="Unknown Wavelet Name"):
        ndwt.wave_filters("haar")


def test_band_bookkeeping():
    lib = ndwt.lib()
    for d in (1, 2, 3, 4):
        for lev in range(1, 8):
            nb = lib.ndwt_num_bands(d, lev)
            assert nb == orc.num_bands(d, lev)
            assert lib.ndwt_level_from_bands(d, nb) == lev
        assert lib.ndwt_level_from_bands(d, (1 << d) + 1) == (-1 if d > 1 else 2)
    assert lib.ndwt_num_bands(5, 1) == -1


def _plan_create(dims, names, dtype=0, cplx=0, l2=0, dil=0, maxlev=1, dev=0):
    lib = ndwt.lib()
    h = ctypes.c_void_p(None)
    d = len(dims)
    rc = lib.ndwt_plan_create(ctypes.byref(h), d, (ctypes.c_int64 * d)(*dims), (ctypes.c_char_p * d)(*[n.encode() for n in names]),
                              dtype, cplx, l2, dil, maxlev, dev)
    return rc, lib.ndwt_last_error().decode(), h


def test_plan_validation_messages_follow_the_reference():
    rc, msg, _ = _plan_create([16, 6], ["db1", "db4"])
    assert rc == 3 and msg == "Second Dimension of Data is shorter than the wavelet filter being used"   # nd_dwt_3D.m:277-286
    rc, msg, _ = _plan_create([16, 16], ["db1", "coif2"])
    assert rc == 2 and msg == "Unknown Wavelet Name"                                                      # wave_filters.m:159
    rc, msg, _ = _plan_create([16] * 5, ["db1"] * 5)
    assert rc == 1
    rc, msg, _ = _plan_create([16, 16], ["db1", "db1"], dtype=7)
    assert rc == 1


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc, msg, h = _plan_create([16, 16, 16], ["db2"] * 3)
    assert rc == 4 and "no CPU path" in msg and not h.value


def test_host_classes_validate_like_the_reference():
    # constructor checks run before any GPU work
    with pytest.raises(ValueError, match="The sizes vector must be length 3"):
        ndwt.nd_dwt_3D("db1", [8, 8])
    with pytest.raises(ValueError, match="Optional inputs must come in pairs"):
        ndwt.nd_dwt_2D("db1", [8, 8], "pres_l2_norm")
    with pytest.raises(ValueError, match="Single precsision is not currently supported for mex computation"):
        ndwt.nd_dwt_1D("db1", 64, "compute", "mex", "precision", "single")
    with pytest.raises(ValueError, match="Third Dimension of Data is shorter"):
        ndwt.nd_dwt_3D(["db1", "db1", "db5"], [16, 16, 8])
    with pytest.raises(ValueError, match="Wavelet Name Must be a string"):
        ndwt.nd_dwt_1D(["db1"], 64)
    with pytest.warns(UserWarning, match="Unknown optional input #1"):
        w = ndwt.nd_dwt_1D("db2", 4096, "perserve_l2_norm", 1)      # the typo of example_nd_dwt_1D.m:14 only warns
    assert w.pres_l2_norm == 0 and w.f_size == {"s1": 4} and w.sizes == [4096]
    w = ndwt.nd_dwt_4D("db2", [8, 8, 8, 8], pres_l2_norm=1, precision="single")
    assert w.wname == ["db2"] * 4 and w.pres_l2_norm == 1 and w._level_from_bands(46) == 3


def test_mex_shim_syntax_against_declaration_stubs():
    """matlab/nd_dwt_hip_mex.c cannot be built here (no MATLAB): the compiler at least checks its syntax and the types it passes
    to the C ABI, against declarations-only stand-ins of mex.h / matrix.h (tests/stubs/), for both complex-storage APIs.
    Syntax only: nothing is linked or run, and no behaviour is pinned by this."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["-DMX_HAS_INTERLEAVED_COMPLEX=1"]):
        r = subprocess.run([cc, "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-std=c99", "-I", os.path.join(root, "tests", "stubs"),
                            "-I", os.path.join(root, "include")] + extra + [os.path.join(root, "matlab", "nd_dwt_hip_mex.c")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_headline_kernels_do_not_spill():
    """A register spill inside a plane loop waits for every load in flight (vmcnt(0)): the kernels of the BASELINE configurations must
    keep their state in registers.  Reads the resource notes of the built objects (csrc/build/*.o); skipped before the first build."""
    import glob
    import re
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    objs = sorted(glob.glob(os.path.join(ROOT, "non-decimated_wavelets_amd", "csrc", "build", "*.o")))
    if not objs or not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        pytest.skip("no built objects / no ROCm llvm tools")
    spills = {}
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            b = os.path.join(tmp, os.path.basename(o))
            if subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={b}.fat", o], capture_output=True).returncode:
                continue
            if subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={b}.fat", f"--output={b}.elf"], capture_output=True).returncode:
                continue
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", f"{b}.elf"], capture_output=True, text=True).stdout
            name = None
            for line in notes.splitlines():
                m = re.match(r"\s+\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.match(r"\s+\.vgpr_spill_count:\s+(\d+)", line)
                if m and name:
                    spills[name] = int(m.group(1))
    assert len(spills) > 100
    # mangled fragments: kernel struct, float, tap length, tile / threads, VEC4 (b1 / b0), ...
    must_not_spill = [
        "Inv3YIfLi8ELi64ELi32ELi1024ELb1ELi4ELi2",      # cfg3 / cfg5 synthesis: two register sets
        "Inv3YIfLi8ELi64ELi32ELi1024ELb0ELi4ELi2",      # the same on rows that are not whole groups of 4 scalars
        "Inv3YIfLi12ELi64ELi32ELi1024ELb1ELi4ELi2",     # cfg4 (db6) synthesis: two register sets, 6 z sums in LDS
        "Fwd3IfLi8ELi64ELi32ELi1024ELi4ELb1",           # cfg3 / cfg5 analysis, tall tile
        "Fwd3IfLi12ELi64ELi32ELi1024ELi2ELb1",          # cfg4 analysis (plain form and the pinned-tap form, ..ELb0ELb0ELb1)
        "Fwd3IfLi10ELi64ELi32ELi1024ELi2ELb1", "Fwd3IfLi14ELi64ELi32ELi1024ELi2ELb1",
        "Fwd3IfLi16ELi64ELi32ELi1024ELi2ELb1ELi4ELi1ELb0ELb0ELb0ELi2",   # 16 taps: two window slots in LDS
        "Fwd3IfLi20ELi64ELi16ELi512ELi4ELb1ELi2ELi1ELb0ELb0ELb0ELi4",    # 20 taps: four
        "Fwd2SIfLi8ELb1", "Inv2SIfLi8ELb1", "Inv2PIfLi8ELi4E",   # cfg2 (synthesis: Inv2P, 4 rows in flight)
        "Den3IfLi2E", "Den3IfLi4E", "Den3IfLi6E", "Den3IfLi8E",
        "Inv3YIfLi18E", "Inv3YIfLi20E",
    ]
    for frag in must_not_spill:
        hit = {k: v for k, v in spills.items() if frag in k and "ELb0ELb1EEE" not in k}      # (not the folded-t A/B variant)
        assert hit, frag
        bad = {k: v for k, v in hit.items() if v and not (frag.startswith("Inv3YIfLi1") and "ELb0ELi4E" in k)}
        assert not bad, bad
