"""Development aid: does the headline depend on the frame being the same every step (it stays in the Infinity Cache)?  Applies of
one plan to R different device-resident frames / outputs in rotation, against re-applying one frame.
    python scripts/distinct_frames.py"""
import pathlib, sys, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
n, size, steps = 256, 4096, 64
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan = _native.Plan(n, coords); plan.set_transfer(k)
geom = _native.Geometry.whole(size, size, 1)
R = 8
imgs = [_native.DeviceBuffer(size * size * 4).upload((100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)) for _ in range(R)]
outs = [_native.DeviceBuffer(size * size * 4) for _ in range(R)]
def run(r, ro=None):
    ro = r if ro is None else ro
    plan.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        plan.apply_device(imgs[i % r].ptr, outs[i % ro].ptr, geom)
    plan.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    run(1); a = run(1); run(R); b = run(R); run(2); c = run(2)
    print(f"same frame {a:.4f} ms per apply, 2 frames in rotation {c:.4f}, {R} frames in rotation {b:.4f} ({100*(b/a-1):+.1f} %)")
    run(R, 1); d = run(R, 1); run(1, R); e = run(1, R)
    print(f"   {R} images into one output {d:.4f} ({100*(d/a-1):+.1f} %), one image into {R} outputs {e:.4f} ({100*(e/a-1):+.1f} %)")
