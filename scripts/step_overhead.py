"""Development aid: wall time of K back-to-back applies between two synchronisations, for several K (fixed + per-step part)."""
import pathlib, sys, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
n, size = 256, 4096
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
plan = _native.Plan(n, coords)
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
img = rng.standard_normal((size, size), dtype=np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
for _ in range(5):
    plan.apply_device(d_img.ptr, d_out.ptr, geom)
plan.synchronize()
res = {}
for steps in (1, 5, 20, 50, 200, 1000):
    best = 1e9
    for _ in range(3):
        plan.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.apply_device(d_img.ptr, d_out.ptr, geom)
        plan.synchronize()
        best = min(best, time.perf_counter() - t0)
    res[steps] = best
    print(f"steps {steps:5d}: {best*1e3:9.3f} ms total, {best*1e6/steps:8.1f} us/step")
a = (res[50] * 1000 - res[1000] * 50) / 950
print(f"fixed part ~ {a*1e6:.0f} us, per step ~ {(res[1000]-a)/1000*1e6:.1f} us")
