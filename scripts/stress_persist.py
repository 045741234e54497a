"""One-off robustness run for the persistent 256-pixel launch: thousands of device-resident applies of one plan, the output
compared bit for bit with the first apply every time (a plane store published too early or a lost tile count would show up
as a changed pixel).    python scripts/stress_persist.py [applies] [size]"""
import pathlib, sys, zlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering

applies = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = 256
rng = np.random.default_rng(3)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
plan = _native.Plan(n, coords)
k = np.empty((len(coords), n, n), np.complex64)
k.real = rng.standard_normal(k.shape, dtype=np.float32)
k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img)
d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
plan.apply_device(d_img.ptr, d_out.ptr, geom)
plan.synchronize()
first = d_out.download(img.shape)
want = zlib.crc32(first.tobytes())
bad = 0
for i in range(applies):
    plan.apply_device(d_img.ptr, d_out.ptr, geom)
    if i % 7 == 0 or i == applies - 1:  # (back-to-back launches in between: the next apply starts while nothing has been read)
        plan.synchronize()
        got = zlib.crc32(d_out.download(img.shape).tobytes())
        bad += got != want
print(f"stress: {applies} applies of {size}x{size}/{n}: {bad} mismatching outputs")
sys.exit(1 if bad else 0)
