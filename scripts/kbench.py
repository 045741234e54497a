"""Development micro-benchmark: time K1 on a device-resident frame with a random transfer kernel.

    RPSF_LIB=/path/to/variant.so python scripts/kbench.py [--n 256] [--size 4096] [--iters 50] [--tag name]
"""
import argparse
import json
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--tag", default="")
ap.add_argument("--overlap", default="auto")
ap.add_argument("--stagger", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((a.size, a.size), a.n)]
plan = _native.Plan(a.n, coords)
k = np.empty((len(coords), a.n, a.n), np.complex64)
k.real = rng.standard_normal(k.shape, dtype=np.float32)
k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
plan.set_overlap_mode(a.overlap)
plan.set_stagger(a.stagger)
img = (100 + 5 * rng.standard_normal((a.size, a.size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img)
d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(a.size, a.size, 1)
plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, a.iters)
alg = plan.transfer_bytes + 2 * img.nbytes
print(json.dumps({"tag": a.tag, "stagger": a.stagger, "n": a.n, "size": a.size, "patches": len(coords), "kernel_ms_med": round(float(np.median(ker)), 4),
                  "kernel_ms_min": round(float(ker.min()), 4), "total_ms_med": round(float(np.median(tot)), 4),
                  "GBs": round(float(alg / np.median(ker) / 1e6), 1), "frac": round(float(alg / np.median(ker) / 1e6 / 8000), 4)}))
