"""Build librpsf_hip.so in-tree with hipcc for gfx950:  python -m regularizepsf_amd.build [--force] [-DRPSF_STAMPS ...]

The library is several translation units (csrc/rpsf.hip: host side + plan-independent kernels; csrc/k1_*.hip,
k2_*.hip: the patch kernels of one group of plans each) compiled in parallel and linked into one shared object.
"""

from __future__ import annotations

import concurrent.futures
import os
import pathlib
import shutil
import subprocess
import sys

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
SOURCES = [CSRC / n for n in ("rpsf.hip", "k1_256.hip", "k1_128.hip", "k1_small.hip", "k2_256.hip", "k2_128.hip")]
HEADERS = [CSRC / n for n in ("rpsf_core.hpp", "rpsf_core2.hpp", "rpsf_kernels.hpp", "rpsf_kernels2.hpp", "rpsf_device.hpp")] + [
    PKG.parent / "include" / "rpsf.h"]
TARGET = PKG / "librpsf_hip.so"
OBJDIR = PKG / "build"
FLAGS = ["--offload-arch=gfx950", "-std=c++20", "-O3", "-fno-slp-vectorize", "-munsafe-fp-atomics", "-fPIC"]


def is_stale() -> bool:
    if not TARGET.exists():
        return True
    built = TARGET.stat().st_mtime
    return any(p.stat().st_mtime > built for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True, defines: tuple[str, ...] = (), target: pathlib.Path | None = None) -> pathlib.Path:
    """Compile every HIP kernel and the C ABI into one shared library (no GPU needed: cross-compiles).

    ``defines`` (e.g. ("-DRPSF_STAMPS",)) and ``target`` select a development variant built next to the product library.
    """
    out = pathlib.Path(target) if target else TARGET
    if not force and not defines and target is None and not is_stale():
        return TARGET
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = OBJDIR / (out.stem if (defines or target) else "product")
    objdir.mkdir(parents=True, exist_ok=True)
    newest_header = max(p.stat().st_mtime for p in HEADERS)

    def compile_one(src: pathlib.Path) -> pathlib.Path:
        obj = objdir / (src.stem + ".o")
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, newest_header):
            return obj
        cmd = [hipcc, *FLAGS, *defines, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    workers = max(1, min(len(SOURCES), (os.cpu_count() or 2)))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *[str(o) for o in objs], "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    defs = tuple(a for a in sys.argv[1:] if a.startswith("-D"))
    tgt = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--target=")), None)
    build(force="--force" in sys.argv, defines=defs, target=pathlib.Path(tgt) if tgt else None)
