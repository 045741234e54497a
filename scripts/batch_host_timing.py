"""Development aid: PCIe-inclusive timing of the host-pointer entry points (frame loop vs rpsf_apply_batch)."""
import sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import regpsf_oracle as orc
from regularizepsf_amd import _native

h = w = 2048; n = 128; frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((h, w), n)]
rng = np.random.default_rng(0)
k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
images = (rng.standard_normal((frames, h, w)) * 5 + 100).astype(np.float32)
plan = _native.Plan(n, coords); plan.set_transfer(k)
pad = _native.PAD_MODES["symmetric"]
plan.apply(images[0], pad); plan.apply_batch(images[:2], pad)
for rep in range(2):
    t0 = time.perf_counter(); a = [plan.apply(im, pad) for im in images]; t1 = time.perf_counter()
    b = plan.apply_batch(images, pad); t2 = time.perf_counter()
    print(f"{frames} frames {h}x{w}: loop over rpsf_apply {1e3*(t1-t0):.1f} ms ({frames*h*w/(t1-t0)/1e6:.0f} Mpx/s), "
          f"rpsf_apply_batch {1e3*(t2-t1):.1f} ms ({frames*h*w/(t2-t1)/1e6:.0f} Mpx/s), identical: {np.array_equal(np.stack(a), b)}")

import regularizepsf_amd as rp
t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
t.apply_batch(images[:2]); t.apply_batch(images[:2], dtype=np.float32)
for dt in (np.float64, np.float32):
    t0 = time.perf_counter(); o = t.apply_batch(images, dtype=dt); t1 = time.perf_counter()
    print(f"ArrayPSFTransform.apply_batch -> {np.dtype(dt).name}: {1e3*(t1-t0):.1f} ms ({frames*h*w/(t1-t0)/1e6:.0f} Mpx/s)")
