import sys, time, pathlib
sys.path.insert(0, "/root/repo")
import numpy as np
from regularizepsf_amd import _native, calculate_covering
n, size = 64, 512
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
plan = _native.Plan(n, coords); plan.set_transfer(k)
img = rng.standard_normal((size, size)).astype(np.float32)
out = np.zeros((size, size), np.float64)
pad = _native.PAD_MODES["symmetric"]
for _ in range(5): plan.apply_host(img, pad, out=out)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); plan.apply_host(img, pad, out=out); ts.append(time.perf_counter() - t0)
print("apply_host 512^2/64 f32->f64: median %.1f us, min %.1f us" % (1e6 * np.median(ts), 1e6 * min(ts)))
out32 = np.zeros((size, size), np.float32)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); plan.apply_host(img, pad, out=out32); ts.append(time.perf_counter() - t0)
print("apply_host 512^2/64 f32->f32: median %.1f us, min %.1f us" % (1e6 * np.median(ts), 1e6 * min(ts)))
d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
print("device loop ms per apply:", plan.apply_device_loop_ms(d_img.ptr, d_out.ptr, geom, 200))
