"""Development aid: is the sweep kernel held back by something two neighbouring CUs share (instruction cache, scalar cache)?  4096^2 / N on all 256 CUs with
256 regions, then on 128 CUs picked by several bit patterns of the CU mask with 128 regions (one region per CU either way, twice the jobs each): if CUs were
independent the half-mask runs take exactly twice as long.   python scripts/sweep_cu_pairs.py [N]"""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = 4096
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
k = np.empty((len(coords), n, n), np.complex64)
k.real = rng.standard_normal(k.shape, dtype=np.float32)
k.imag = rng.standard_normal(k.shape, dtype=np.float32)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img)
d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)


def pattern(name):
    bits = np.zeros(256, bool)
    if name == "all":
        bits[:] = True
    elif name == "even bits":
        bits[0::2] = True
    elif name == "pairs 0,1 of 4":
        bits[(np.arange(256) % 4) < 2] = True
    elif name == "low 16 of every 32":
        bits[(np.arange(256) % 32) < 16] = True
    elif name == "first 128":
        bits[:128] = True
    elif name == "every 4th pair of words":  # words 0, 2, 4, 6 (whole 32-bit words)
        bits[(np.arange(256) // 32) % 2 == 0] = True
    words = np.zeros(8, np.uint32)
    for i in np.nonzero(bits)[0]:
        words[i // 32] |= np.uint32(1 << (i % 32))
    return words, int(bits.sum())


base = None
for name in ("all", "even bits", "pairs 0,1 of 4", "low 16 of every 32", "first 128", "every 4th pair of words"):
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    mask, cus = pattern(name)
    plan.set_cu_mask(mask)
    plan.set_sweep_regions(cus)
    info = plan.sweep_info()
    plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
    tot, _ = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 30)
    ms = float(np.median(tot))
    base = base or ms
    print(f"N={n} mask '{name}': {cus} CUs, {info['regions']} regions, {info['jobs']} jobs: {ms:.4f} ms = {ms / base:.2f} x the full chip", flush=True)
