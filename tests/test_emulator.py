"""Drive the kernel's per-thread phases (regularizepsf_amd/csrc/rpsf_core.hpp, and rpsf_core2.hpp for the
second-generation 128/256-pixel plans) on the CPU, thread by thread, and check the result against the reference goldens.  This pins the index algebra shared with the HIP
kernel (digit layouts, LDS addressing, slot table, packed-K format) without needing a GPU."""

import ctypes
import pathlib
import shutil
import subprocess

import numpy as np
import pytest

from tests.helpers import APPLY_CASES, load_apply_case, rel_errors

ROOT = pathlib.Path(__file__).resolve().parent.parent
EMU_SRC = ROOT / "tests" / "emu" / "emu.cpp"
EMU_LIB = ROOT / "tests" / "emu" / "libemu.so"
CORE = ROOT / "regularizepsf_amd" / "csrc" / "rpsf_core.hpp"
MODES = {"constant": 0, "symmetric": 1, "reflect": 2, "edge": 3, "wrap": 4}


@pytest.fixture(scope="module")
def emu():
    if not EMU_LIB.exists() or EMU_LIB.stat().st_mtime < max(EMU_SRC.stat().st_mtime, CORE.stat().st_mtime):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
        if not pathlib.Path(clang).exists():
            clang = shutil.which("clang++")
        if clang is None:
            pytest.skip("no clang++ to build the emulator")
        subprocess.run([clang, "-std=c++20", "-O1", "-shared", "-fPIC", "-o", str(EMU_LIB), str(EMU_SRC)], check=True)
    return ctypes.CDLL(str(EMU_LIB))


@pytest.mark.parametrize("case", [c[0] for c in APPLY_CASES if c[7] in MODES])
def test_emulated_kernel_matches_reference_golden(emu, case):
    fx, coords, k = load_apply_case(case)
    image = np.ascontiguousarray(fx["image"], np.float32)
    h, w = image.shape
    c = np.ascontiguousarray(np.array(coords, np.int32))
    kk = np.ascontiguousarray(k, np.complex64)
    out = np.zeros((h, w), np.float32)
    vp = ctypes.c_void_p
    rc = emu.emu_apply(k.shape[1], len(coords), c.ctypes.data_as(vp), h, w, MODES[str(fx["pad_mode"])],
                       ctypes.c_float(0.0), image.ctypes.data_as(vp), kk.ctypes.data_as(vp), out.ctypes.data_as(vp))
    assert rc == 0
    rel_max, rel_l2 = rel_errors(out, fx["expected"])
    assert rel_max <= 1e-5 and rel_l2 <= 1e-5, (rel_max, rel_l2)


EMU2_SRC = ROOT / "tests" / "emu" / "emu2.cpp"
EMU2_LIB = ROOT / "tests" / "emu" / "libemu2.so"
CORE2 = ROOT / "regularizepsf_amd" / "csrc" / "rpsf_core2.hpp"


@pytest.fixture(scope="module")
def emu2():
    if not EMU2_LIB.exists() or EMU2_LIB.stat().st_mtime < max(EMU2_SRC.stat().st_mtime, CORE.stat().st_mtime, CORE2.stat().st_mtime):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
        if not pathlib.Path(clang).exists():
            clang = shutil.which("clang++")
        if clang is None:
            pytest.skip("no clang++ to build the emulator")
        subprocess.run([clang, "-std=c++20", "-O1", "-shared", "-fPIC", "-o", str(EMU2_LIB), str(EMU2_SRC)], check=True)
    return ctypes.CDLL(str(EMU2_LIB))


@pytest.mark.parametrize("direct", [0, 1, 2])
@pytest.mark.parametrize("case", [c[0] for c in APPLY_CASES if c[7] in MODES and c[3] >= 128])
def test_emulated_second_generation_kernel_matches_reference_golden(emu2, case, direct):
    """direct = 1: quadrant by quadrant, the first patch over a lattice tile stores and the later ones accumulate
    (what the tile flags arrange on the GPU); direct = 0: plain adds into a cleared image; direct = 2: four colour planes
    (16-byte stores, rim patches included where the image width allows) summed at the end."""
    fx, coords, k = load_apply_case(case)
    image = np.ascontiguousarray(fx["image"], np.float32)
    h, w = image.shape
    c = np.ascontiguousarray(np.array(coords, np.int32))
    kk = np.ascontiguousarray(k, np.complex64)
    out = np.zeros((h, w), np.float32)
    vp = ctypes.c_void_p
    rc = emu2.emu2_apply(k.shape[1], len(coords), c.ctypes.data_as(vp), h, w, MODES[str(fx["pad_mode"])],
                         ctypes.c_float(0.0), image.ctypes.data_as(vp), kk.ctypes.data_as(vp), out.ctypes.data_as(vp), direct)
    assert rc == 0
    rel_max, rel_l2 = rel_errors(out, fx["expected"])
    assert rel_max <= 1e-5 and rel_l2 <= 1e-5, (rel_max, rel_l2)


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("shape", [(256, 384), (200, 250)])
def test_emulated_rim_patches_every_pad_mode(emu2, mode, shape):
    """Rim patches of the second-generation plans take 16-byte loads where the np.pad map keeps a unit of four pixels
    together (ascending, or mirrored by 'symmetric') and pixel-by-pixel loads elsewhere; 16-byte plane stores where the
    image width is a multiple of 4.  Every pad mode the kernel evaluates itself, an aligned and an unaligned width,
    against the oracle."""
    from oracle import regpsf_oracle as orc

    n = 128
    h, w = shape
    rng = np.random.default_rng(5)
    image = (100 + 5 * rng.standard_normal((h, w))).astype(np.float32)
    coords = [tuple(int(v) for v in c) for c in orc.calculate_covering((h, w), n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    expected = orc.apply_transfer(image, coords, k, pad_mode=mode)
    c = np.ascontiguousarray(np.array(coords, np.int32))
    vp = ctypes.c_void_p
    for direct in (0, 2):
        out = np.zeros((h, w), np.float32)
        rc = emu2.emu2_apply(n, len(coords), c.ctypes.data_as(vp), h, w, MODES[mode], ctypes.c_float(0.0), image.ctypes.data_as(vp),
                             k.ctypes.data_as(vp), out.ctypes.data_as(vp), direct)
        assert rc == 0
        rel_max, rel_l2 = rel_errors(out, expected)
        assert rel_max <= 1e-5 and rel_l2 <= 1e-5, (mode, direct, rel_max, rel_l2)
