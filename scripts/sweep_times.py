"""Timing of the sweep kernel at the sizes VERDICT round 5 names (development aid; RPSF_LIB selects a variant build).
    python scripts/sweep_times.py [--cases 32:4096,64:4096,...] [--iters 30]"""
import argparse
import json
import os
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="32:512,64:512,16:512,32:2048,32:4096,64:4096,16:4096")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--mode", default="auto")
ap.add_argument("--regions", default="0", help="comma-separated target region counts (0 = the library's own)")
a = ap.parse_args()
rng = np.random.default_rng(0)
tag = pathlib.Path(os.environ.get("RPSF_LIB", "product")).stem
for case in a.cases.split(","):
    n, size = (int(x) for x in case.split(":"))
    coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
    k = np.empty((len(coords), n, n), np.complex64)
    k.real = rng.standard_normal(k.shape, dtype=np.float32)
    k.imag = rng.standard_normal(k.shape, dtype=np.float32)
    img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
    d_img = _native.DeviceBuffer(img.nbytes).upload(img)
    d_out = _native.DeviceBuffer(img.nbytes)
    geom = _native.Geometry.whole(size, size, 1)
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    plan.set_overlap_mode(a.mode)
    for target in [int(t) for t in a.regions.split(",")]:
        if target:
            plan.set_sweep_regions(target)
        info = plan.sweep_info()
        plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
        tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, a.iters)
        alg = plan.transfer_bytes + 2 * img.nbytes
        print(json.dumps({"lib": tag, "n": n, "size": size, "mode": a.mode, "target": target, "regions": info["regions"], "jobs": info["jobs"],
                          "recompute": round(info["patch_slots"] / len(coords), 3), "ks": info["slabs_per_phase"],
                          "ms_med": round(float(np.median(tot)), 4), "ms_min": round(float(tot.min()), 4),
                          "frac": round(float(alg / np.median(tot) / 1e6 / 8000), 4)}), flush=True)
