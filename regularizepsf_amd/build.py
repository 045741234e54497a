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
SOURCES = [CSRC / n for n in ("rpsf.hip", "k1_256.hip", "k1_128.hip", "k1_small.hip", "k2_256.hip", "k2_256p.hip", "k2_128.hip", "k2_128p.hip", "k2_128pc.hip", "k2_128pcs.hip", "k3_16.hip", "k3_32.hip", "k3_64.hip")]
HEADERS = [CSRC / n for n in ("rpsf_core.hpp", "rpsf_core2.hpp", "rpsf_core3.hpp", "rpsf_plan3.hpp", "rpsf_kernels.hpp", "rpsf_kernels2.hpp", "rpsf_kernels3.hpp", "rpsf_device.hpp", "rpsf_hostpipe.hpp")] + [
    PKG.parent / "include" / "rpsf.h"]
TARGET = PKG / "librpsf_hip.so"
OBJDIR = PKG / "build"
FLAGS = ["--offload-arch=gfx950", "-std=c++20", "-O3", "-fno-slp-vectorize", "-munsafe-fp-atomics", "-fPIC"]


def is_stale() -> bool:
    if not TARGET.exists():
        return True
    built = TARGET.stat().st_mtime
    return any(p.stat().st_mtime > built for p in SOURCES + HEADERS)


def llvm_bin_dir() -> pathlib.Path:
    """Where llvm-objcopy and clang-offload-bundler of the ROCm install in use live (next to the hipcc that compiles)."""
    tools = ("llvm-objcopy", "clang-offload-bundler")
    roots = [os.environ.get(v) for v in ("ROCM_PATH", "HIP_PATH")]
    hipcc = shutil.which("hipcc")
    if hipcc:
        roots.append(str(pathlib.Path(hipcc).resolve().parent.parent))
    roots.append("/opt/rocm")
    tried = []
    for root in roots:
        if not root:
            continue
        for sub in ("lib/llvm/bin", "llvm/bin", "bin"):
            d = pathlib.Path(root) / sub
            tried.append(str(d))
            if all((d / t).exists() for t in tools):
                return d
    raise RuntimeError(f"llvm-objcopy / clang-offload-bundler not found (needed to check the persistent kernels' descriptors); "
                       f"looked in {', '.join(tried)} - set ROCM_PATH")


def check_reentry_contract(obj: pathlib.Path, kernel: str = "patch_kernel2_256p") -> None:
    """The persistent patch kernel jumps back to its own first instruction (RPSF_REENTER, csrc/rpsf_kernels2.hpp) and
    rebuilds the state a fresh workgroup starts with: s[0:1] = kernarg segment pointer, s2 = workgroup id x, v0 = workitem
    id x.  That is only right while the kernel descriptor asks the hardware for exactly these; read it back from the
    built code object and fail the build otherwise."""
    import struct
    import tempfile

    llvm = llvm_bin_dir()
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = pathlib.Path(tmp) / "fat.bin", pathlib.Path(tmp) / "k.co"
        for cmd in ([str(llvm / "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", str(obj), str(pathlib.Path(tmp) / "x.o")],
                    [str(llvm / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                     "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"]):
            done = subprocess.run(cmd, capture_output=True, text=True)
            if done.returncode != 0:
                raise RuntimeError(f"{kernel}: cannot read the code object back ({' '.join(cmd)}): {done.stderr.strip()}")
        elf = co.read_bytes()
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    kd, kd_addr, entry_addr = None, None, None
    for name_off, typ, _, addr, off, size, link, _, _, entsize in secs:
        if typ not in (2, 11):  # SHT_SYMTAB, SHT_DYNSYM
            continue
        str_off = secs[link][4]
        for j in range(size // entsize):
            st_name, _, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + j * entsize)
            end = elf.index(b"\0", str_off + st_name)
            sym = elf[str_off + st_name:end].decode()
            if sym == kernel + ".kd":
                sec = secs[st_shndx]
                kd = elf[sec[4] + st_value - sec[3]:sec[4] + st_value - sec[3] + 64]
                kd_addr = st_value
            elif sym == kernel:
                entry_addr = st_value
    if kd is None or len(kd) != 64 or entry_addr is None:
        raise RuntimeError(f"{kernel} / {kernel}.kd not found in {obj}")
    private_size, = struct.unpack_from("<I", kd, 4)
    entry_off, = struct.unpack_from("<q", kd, 16)
    rsrc2, props, preload = struct.unpack_from("<IHH", kd, 52)
    got = {"user_sgprs": (rsrc2 >> 1) & 31, "wg_id_x": (rsrc2 >> 7) & 1, "wg_id_y": (rsrc2 >> 8) & 1, "wg_id_z": (rsrc2 >> 9) & 1,
           "wg_info": (rsrc2 >> 10) & 1, "workitem_id": (rsrc2 >> 11) & 3, "code_properties": props & 0x7F,
           # scratch: gfx950 sets flat scratch up by itself (architected), so the hardware enable bit must simply follow "the
           # kernel has scratch"; a compiler that wanted a scratch offset or buffer in SGPRs would show in user_sgprs /
           # code_properties above, and one that set the bit without scratch (or the reverse) shows here
           "private_segment_enable": rsrc2 & 1,
           "kernarg_preload": preload,
           # RPSF_REENTER jumps to the kernel's symbol: that must be the address the descriptor starts a fresh workgroup at
           "entry_is_symbol": int(kd_addr + entry_off == entry_addr)}
    want = {"user_sgprs": 2, "wg_id_x": 1, "wg_id_y": 0, "wg_id_z": 0, "wg_info": 0, "workitem_id": 0, "code_properties": 0x08,
            "private_segment_enable": int(private_size > 0), "kernarg_preload": 0, "entry_is_symbol": 1}
    if got != want:
        raise RuntimeError(f"{kernel}: the kernel descriptor no longer matches what RPSF_REENTER restores: {got} != {want}")


def build(force: bool = False, verbose: bool = True, defines: tuple[str, ...] = (), target: pathlib.Path | None = None,
          only: tuple[str, ...] = ()) -> pathlib.Path:
    """Compile every HIP kernel and the C ABI into one shared library (no GPU needed: cross-compiles).

    ``defines`` (e.g. ("-DRPSF_STAMPS",)) and ``target`` select a development variant built next to the product library;
    ``only`` (source stems, e.g. ("k2_256p",)) compiles just those translation units with the variant's defines and takes the
    product's objects for the rest (a variant that touches one kernel builds in seconds).
    """
    out = pathlib.Path(target) if target else TARGET
    if not force and not defines and target is None and not is_stale():
        return TARGET
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = OBJDIR / (out.stem if (defines or target) else "product")
    objdir.mkdir(parents=True, exist_ok=True)
    newest_header = max(p.stat().st_mtime for p in HEADERS)

    def compile_one(src: pathlib.Path) -> pathlib.Path:
        if only and src.stem not in only:
            obj = OBJDIR / "product" / (src.stem + ".o")
            if not obj.exists():
                raise RuntimeError(f"{obj} missing: build the product library first")
            return obj
        obj = objdir / (src.stem + ".o")
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, newest_header):
            return obj
        # (development: e.g. -mllvm options - honoured for VARIANT builds only, never for the product library, whose objects carry no
        # record of a stray environment variable)
        extra = os.environ.get("RPSF_EXTRA_HIPCC_FLAGS", "").split() if (defines or target) else []
        cmd = [hipcc, *FLAGS, *extra, *defines, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    workers = max(1, min(len(SOURCES), (os.cpu_count() or 2)))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    check_reentry_contract(next(o for o in objs if o.stem == "k2_256p"), "patch_kernel2_256p")
    check_reentry_contract(next(o for o in objs if o.stem == "k2_128p"), "patch_kernel2_128p")
    check_reentry_contract(next(o for o in objs if o.stem == "k2_128pc"), "patch_kernel2_128pc")
    check_reentry_contract(next(o for o in objs if o.stem == "k2_128pcs"), "patch_kernel2_128pcs")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *[str(o) for o in objs], "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    defs = tuple(a for a in sys.argv[1:] if a.startswith("-D"))
    tgt = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--target=")), None)
    only = next((tuple(a.split("=", 1)[1].split(",")) for a in sys.argv[1:] if a.startswith("--only=")), ())
    build(force="--force" in sys.argv, defines=defs, target=pathlib.Path(tgt) if tgt else None, only=only)
