"""The C-ABI shared library loads and exports every symbol include/rpsf.h declares; with no GPU the
product path fails loudly instead of falling back to anything."""

import ctypes
import pathlib
import re

import numpy as np
import pytest

import regularizepsf_amd as rp
from regularizepsf_amd import _native

ROOT = pathlib.Path(__file__).resolve().parent.parent


def declared_functions():
    text = (ROOT / "include" / "rpsf.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rpsf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    handle = _native.lib()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(handle, name), f"{name} is declared in include/rpsf.h but not exported"
    assert set(names) == set(_native._PROTOTYPES), "ctypes prototypes and header disagree"


def test_geometry_struct_matches_header_layout():
    assert ctypes.sizeof(_native.Geometry) == 12 * 4
    g = _native.Geometry.whole(10, 20, 1)
    assert (g.height, g.width, g.image_rows, g.ld_image, g.out_rows, g.ld_out) == (10, 20, 10, 20, 10, 20)


def test_bad_arguments_are_reported_not_crashed():
    handle = _native.lib()
    assert handle.rpsf_plan_create(None, 0, 64, 1, None) == _native.E_BADARG
    assert b"null" in handle.rpsf_last_error()
    out = ctypes.c_void_p()
    coords = np.zeros((1, 2), np.int32)
    for size in (1, 5000):  # 16..256 have compiled plans, 2..4096 the hipFFT fallback, nothing else exists
        rc = handle.rpsf_plan_create(ctypes.byref(out), 0, size, 1, coords.ctypes.data_as(ctypes.c_void_p))
        assert rc == _native.E_UNSUPPORTED and str(size).encode() in handle.rpsf_last_error()
    # the communicator's rank count (ncclCommCount behind it): null arguments are an error code, not a crash
    n = ctypes.c_int(-1)
    assert handle.rpsf_comm_ranks(None, ctypes.byref(n)) == _native.E_BADARG and n.value == -1


def _gpu_present():
    try:
        return _native.device_count() > 0
    except _native.NativeError:
        return False


@pytest.mark.skipif(_gpu_present(), reason="only meaningful on a machine without a GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    k = np.ones((1, 32, 32), np.complex64)
    with pytest.raises(_native.NativeError):
        rp.ArrayPSFTransform(rp.IndexedCube([(0, 0)], k)).apply(np.zeros((64, 64), np.float32))
    src = rp.ArrayPSF(rp.IndexedCube([(0, 0)], np.ones((1, 16, 16))))
    with pytest.raises(_native.NativeError):
        rp.ArrayPSFTransform.construct(src, src, 3.0, 0.1)


def test_product_never_imports_the_oracle():
    for path in (ROOT / "regularizepsf_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".hpp"}:
            assert "oracle" not in path.read_text().replace("# oracle", ""), f"{path} mentions the oracle"


def test_saturation_fill_matches_the_reference_loop():
    """rpsf_saturation_fill (host code in the library) vs the per-pixel np.nanmean loop of transform.py:135-138,
    including windows that hang over the array edge (Python negative-slice rule) and all-NaN windows."""
    rng = np.random.default_rng(21)
    for width in (7, 5, 1, 0, 2):
        padded = rng.standard_normal((40, 37)) * 10
        mask = rng.random((40, 37)) > 0.8
        mask[:3, :] |= rng.random((3, 37)) > 0.3      # dense near the top edge
        mask[:, -2:] = True                            # and on the right edge
        mask[20:30, 5:15] = True                       # a block big enough to produce empty (all-NaN) windows
        expect = padded.copy()
        expect[mask] = np.nan
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for i, j in zip(*np.where(mask)):
                    expect[i, j] = np.nanmean(expect[i - width // 2 : i + width // 2, j - width // 2 : j + width // 2])
        got = padded.copy()
        got[mask] = np.nan
        _native.saturation_fill(got, mask, width)
        assert np.array_equal(np.isnan(got), np.isnan(expect)), width
        ok = ~np.isnan(expect)
        assert np.allclose(got[ok], expect[ok], rtol=1e-13, atol=0), width


def test_development_switches_still_compile(tmp_path):
    """The timing ablations and phase stamps that stay in the kernel sources (RPSF2_ABL_*, RPSF_STAMPS / RPSF_WAVE_STAMPS; `if constexpr
    (dev::...)` branches in rpsf_kernels2.hpp) are never set by the product build: compile the persistent kernels with all of them, so
    that an edit of the hot path cannot silently rot them.  (No GPU needed: hipcc cross-compiles.)"""
    import shutil
    import subprocess

    from regularizepsf_amd import build as hip_build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    sets = {
        "k2_256p.hip": ["-DRPSF_STAMPS", "-DRPSF_WAVE_STAMPS", "-DRPSF2_ABL_NOGATHER", "-DRPSF2_ABL_NOK", "-DRPSF2_ABL_NOSTORE",
                        "-DRPSF2_ABL_NOVALU", "-DRPSF2_ABL_NOLDS", "-DRPSF2_ABL_NOBAR"],
        "k2_128p.hip": ["-DRPSF_STAMPS", "-DRPSF2_ABL_NOBAR_MASK=20"],
        "k2_128pcs.hip": ["-DRPSF2_SKEL_PRESUM"],  # the round-6 timing skeleton of a lattice-row pre-sum
        "k3_32.hip": ["-DRPSF3_ABL_ONE_K", "-DRPSF3_ABL_ONE_SLAB", "-DRPSF3_ABL_NO_FLUSH", "-DRPSF3_ABL_NO_COLUMN_FFTS", "-DRPSF3_ABL_NO_WAIT", "-DRPSF3_ABL_FORCE_ERR", "-DRPSF3_DIRECT_GATHER", "-DRPSF3_STAMPS"],  # the sweep kernel's ablations (5.8)
        "rpsf.hip": ["-DRPSF_DEV_ENV"],  # the environment knobs of the development sweeps
    }
    for source, defines in sets.items():
        out = tmp_path / (source + ".o")
        done = subprocess.run([hipcc, *hip_build.FLAGS, *defines, "-c", str(hip_build.CSRC / source), "-o", str(out)],
                              capture_output=True, text=True)
        assert done.returncode == 0, done.stderr[-2000:]
        assert out.stat().st_size > 0


def test_the_shipped_library_reads_two_environment_variables():
    """A stray variable in a production environment must not change which kernel runs: the product sources call getenv for the host
    pool's two variables only (documented in include/rpsf.h); the development knobs go through dev_env(), which compiles to nothing
    without -DRPSF_DEV_ENV, and everything a test or a caller may pin is rpsf_plan_set_option."""
    import re

    from regularizepsf_amd import build as hip_build

    reads = {}
    for path in sorted(p for p in hip_build.CSRC.glob("*") if p.is_file()):
        text = path.read_text()
        for name in re.findall(r'std::getenv\("([A-Z_0-9]+)"\)', text):
            reads.setdefault(name, path.name)
    assert set(reads) == {"RPSF_HOST_THREADS", "RPSF_HOST_AFFINITY"}, reads
    header = (ROOT / "include" / "rpsf.h").read_text()
    assert all(name in header for name in reads)
    # the one other getenv of the sources is dev_env's own, under the define the product build never sets
    host = (hip_build.CSRC / "rpsf.hip").read_text()
    assert host.count("std::getenv(") == 1 and "#if defined(RPSF_DEV_ENV)\n  return std::getenv(name);" in host
    assert "RPSF_DEV_ENV" not in " ".join(hip_build.FLAGS)
    binary = (ROOT / "regularizepsf_amd" / "librpsf_hip.so").read_bytes()
    for knob in (b"RPSF_STRIPS", b"RPSF_SUM_FIRST", b"RPSF_V1\0", b"RPSF_NO_PERSIST", b"RPSF_K_CACHED", b"RPSF_STREAM_GROUP"):
        assert knob not in binary, knob


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_host_worker_pool_runs_every_part_exactly_once(threads):
    """The persistent host pool behind the host-array entry points (csrc/rpsf_hostpipe.hpp) needs no GPU: jobs of 1 ... 37 parts from four
    calling threads at once (the pool serves one job at a time; callers take turns), thousands of times - every part runs exactly once and
    run() returns only when all have.  The pool's width is fixed when a process first uses it, so each width runs in a process of its own."""
    import subprocess
    import sys

    code = ("import ctypes, sys; sys.path.insert(0, %r); from regularizepsf_amd import _native; lib = _native.lib(); "
            "bad = ctypes.c_int(-1); n = ctypes.c_int(0); "
            "assert lib.rpsf_host_pool_selftest(4, 1500, 37, ctypes.byref(bad)) == 0, lib.rpsf_last_error(); "
            "assert lib.rpsf_host_threads(ctypes.byref(n)) == 0; print(bad.value, n.value)" % str(ROOT))
    env = dict(__import__("os").environ, RPSF_HOST_THREADS=threads)
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert done.returncode == 0, done.stderr[-1500:]
    bad, width = (int(v) for v in done.stdout.split())
    assert bad == 0 and width == max(1, int(threads))
