"""Development aid: throughput of back-to-back applies when two plans (two streams, own planes and outputs, the same K) take
turns, against one plan - how much of the launch's head and tail a frame pipeline hides.   python scripts/pipeline_two_plans.py"""
import pathlib, sys, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
n, size, steps = 256, 4096, 60
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
plans = [_native.Plan(n, coords) for _ in range(2)]
for p in plans:
    p.set_transfer(k)
d_img = _native.DeviceBuffer(img.nbytes).upload(img)
outs = [_native.DeviceBuffer(img.nbytes) for _ in range(2)]
geom = _native.Geometry.whole(size, size, 1)
def run(which):
    for p in plans: p.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        j = which(i)
        plans[j].apply_device(d_img.ptr, outs[j].ptr, geom)
    for p in plans: p.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    run(lambda i: 0); a = run(lambda i: 0)
    run(lambda i: i & 1); b = run(lambda i: i & 1)
    print(f"one plan {a:.4f} ms per apply, two plans alternating {b:.4f} ms per apply ({100*(b/a-1):+.1f} %)")
ref = outs[0].download(img.shape); oth = outs[1].download(img.shape)
print("outputs identical:", bool(np.array_equal(ref, oth)))
