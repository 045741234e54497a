"""Development aid: device-resident batch of frames sharing one transfer array (any plan), per-frame kernel time.
    python scripts/batch_kbench.py --n 256 --size 4096 --frames 4"""
import argparse, json, pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256); ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--frames", type=int, default=4); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
n, size, f = a.n, a.size, a.frames
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
plan = _native.Plan(n, coords)
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
imgs = rng.standard_normal((f, size, size), dtype=np.float32)
d_img = _native.DeviceBuffer(imgs.nbytes).upload(imgs); d_out = _native.DeviceBuffer(imgs.nbytes)
geom = _native.Geometry.whole(size, size, 1)
plan.apply_batch_device_timed(d_img.ptr, d_out.ptr, f, size * size, size * size, geom, 3)
tot, ker = plan.apply_batch_device_timed(d_img.ptr, d_out.ptr, f, size * size, size * size, geom, a.iters)
print(json.dumps({"n": n, "size": size, "frames": f, "patches": len(coords) * f,
                  "kernel_ms_per_frame": round(float(np.median(ker)) / f, 4), "total_ms_per_frame": round(float(np.median(tot)) / f, 4),
                  "Gpx_s": round(size * size * f / float(np.median(tot)) / 1e6, 1)}))
