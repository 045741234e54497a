"""Build librpsf_hip.so in-tree with hipcc for gfx950:  python -m regularizepsf_amd.build"""

from __future__ import annotations

import pathlib
import shutil
import subprocess
import sys

PKG = pathlib.Path(__file__).resolve().parent
SOURCES = [PKG / "csrc" / "rpsf.hip"]
HEADERS = [PKG / "csrc" / "rpsf_core.hpp", PKG / "csrc" / "rpsf_kernels.hpp", PKG.parent / "include" / "rpsf.h"]
TARGET = PKG / "librpsf_hip.so"


def is_stale() -> bool:
    if not TARGET.exists():
        return True
    built = TARGET.stat().st_mtime
    return any(p.stat().st_mtime > built for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    """Compile every HIP kernel and the C ABI into one shared library (no GPU needed: cross-compiles)."""
    if not force and not is_stale():
        return TARGET
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-std=c++20", "-O3", "-fno-slp-vectorize", "-munsafe-fp-atomics", "-shared", "-fPIC",
           "-o", str(TARGET), *[str(s) for s in SOURCES], "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return TARGET


if __name__ == "__main__":
    build(force="--force" in sys.argv)
