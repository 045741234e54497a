"""Development aid: end-to-end (host array in, host float64 array out) timing of ArrayPSFTransform.apply."""
import sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import regularizepsf_amd as rp
from regularizepsf_amd import _native

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((size, size), n)]
rng = np.random.default_rng(0)
k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
for dt in (np.float32, np.float64):
    img = (rng.standard_normal((size, size)) * 5 + 100).astype(dt)
    t.apply(img)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = t.apply(img); ts.append(time.perf_counter() - t0)
    plan = t._device_plan()
    i32 = img.astype(np.float32)
    t0 = time.perf_counter(); a = img.astype(np.float32, copy=False); t1 = time.perf_counter()
    o32 = plan.apply(i32, _native.PAD_MODES["symmetric"]); t2 = time.perf_counter()
    o64 = o32.astype(np.float64); t3 = time.perf_counter()
    print(f"{size}^2 N={n} input {np.dtype(dt).name}: apply() median {1e3*np.median(ts):.1f} ms  "
          f"[to f32 {1e3*(t1-t0):.1f} | rpsf_apply (H2D+kernels+D2H) {1e3*(t2-t1):.1f} | to f64 {1e3*(t3-t2):.1f}]")
