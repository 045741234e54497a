"""Development aid: which output pixels does the sweep kernel write, and are the written ones right?"""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402

rng = np.random.default_rng(0)
n, size = 32, 512
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
k = np.empty((len(coords), n, n), np.complex64)
k.real = rng.standard_normal(k.shape, dtype=np.float32)
k.imag = rng.standard_normal(k.shape, dtype=np.float32)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
geom = _native.Geometry.whole(size, size, 1)
d_img = _native.DeviceBuffer(img.nbytes).upload(img)
ref = None
for mode in ("planes", "sweep"):
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    plan.set_overlap_mode(mode)
    d_out = _native.DeviceBuffer(img.nbytes).upload(np.full((size, size), 12345.0, np.float32))
    plan.apply_device(d_img.ptr, d_out.ptr, geom)
    plan.synchronize()
    out = d_out.download((size, size)).copy()
    if mode == "planes":
        ref = out
        continue
    untouched = out == 12345.0
    good = np.abs(out - ref) <= 1e-4 * np.abs(ref).max()
    print("untouched", int(untouched.sum()), "right", int(good.sum()), "wrong", int((~good & ~untouched).sum()), "of", out.size)
    h = n // 2
    for name, m in (("untouched", untouched), ("wrong", ~good & ~untouched)):
        blk = m.reshape(size // h, h, size // h, h).mean(axis=(1, 3))
        print(name, "fraction per half-patch block (0-9):")
        for r in range(size // h):
            print("   ", "".join(str(min(9, int(x * 10))) for x in blk[r]))
    # inside one block: which rows / columns
    m = (~good & ~untouched)[64:96, 64:192]
    print("wrong pixels inside rows 64..95, cols 64..191:")
    for r in range(32):
        print("   ", "".join("#" if x else "." for x in m[r]))
