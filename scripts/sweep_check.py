"""Sweep kernel against the colour-plane path on random frames (development aid): where do they differ?"""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402

rng = np.random.default_rng(0)
for n, size in ((32, 512), (16, 256), (64, 512)):
    coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
    k = np.empty((len(coords), n, n), np.complex64)
    k.real = rng.standard_normal(k.shape, dtype=np.float32)
    k.imag = rng.standard_normal(k.shape, dtype=np.float32)
    img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
    res = {}
    for mode in ("sweep", "planes"):
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        plan.set_overlap_mode(mode)
        res[mode] = plan.apply(img, 1).astype(np.float64)
    d = np.abs(res["sweep"] - res["planes"]) / np.abs(res["planes"]).max()
    bad = d > 1e-5
    print(n, size, "max rel", d.max(), "bad pixels", int(bad.sum()), "nan", int(np.isnan(res["sweep"]).sum()))
    if bad.any() or np.isnan(d).any():
        bad |= np.isnan(d)
        rows = np.where(bad.any(axis=1))[0]
        cols = np.where(bad.any(axis=0))[0]
        print("   bad rows", rows[:8], "...", rows[-4:], "count", len(rows), " bad cols", cols[:8], "...", cols[-4:], "count", len(cols))
        h = n // 2
        blk = bad.reshape(size // h, h, size // h, h).any(axis=(1, 3))
        print("   bad half-patch blocks:", int(blk.sum()), "of", blk.size)
        for r in range(min(blk.shape[0], 34)):
            print("   ", "".join("#" if x else "." for x in blk[r][:80]))
