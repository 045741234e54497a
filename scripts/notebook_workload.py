"""The reference's own example workload end to end (docs/source/example.ipynb cells 2 / 25: psf_size = 64, a list of frames, every one
corrected with `transform.apply(image, saturation_threshold=2_000)`), host arrays in and out, with and without saturated pixels.

    python scripts/notebook_workload.py [--size 512] [--frames 200] [--patch 64]
"""
import argparse
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import regularizepsf_amd as rp  # noqa: E402
from oracle import regpsf_oracle as orc  # noqa: E402  (synthetic inputs and the check)

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--patch", type=int, default=64)
a = ap.parse_args()
h = w = a.size
n = a.patch
coords, k = orc.synthetic_transfer(h, w, n, alpha=1.0, epsilon=0.1)
t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
frames = [orc.starfield(h, w, 100 + i) for i in range(a.frames)]  # amplitudes up to 1e5: a few pixels per frame exceed 2000
n_sat = sum(int((f > 2000).sum()) for f in frames)
t.apply(frames[0])
t.apply(frames[0], saturation_threshold=2000)
t_warm = time.perf_counter()
while time.perf_counter() - t_warm < 0.3:  # (as bench.py's prewarm: the first 100 ms of work after start-up run below steady clocks - 200 small frames are 30 ms)
    t.apply(frames[0])
for label, kwargs in (("default (no saturation branch)", {}), ("saturation_threshold=2000", {"saturation_threshold": 2000})):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        outs = [t.apply(f, **kwargs) for f in frames]
        best = min(best, time.perf_counter() - t0)
    print(f"{a.frames} frames of {h}x{w}, N={n}, {label}: {1e3 * best / a.frames:.3f} ms per frame ({a.frames * h * w / best / 1e6:.0f} Mpx/s)", flush=True)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    stack = t.apply_batch(frames)
    best = min(best, time.perf_counter() - t0)
print(f"apply_batch (streamed, no saturation): {1e3 * best / a.frames:.3f} ms per frame ({a.frames * h * w / best / 1e6:.0f} Mpx/s)")
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    stack_sat = t.apply_batch(frames, saturation_threshold=2000)
    best = min(best, time.perf_counter() - t0)
print(f"apply_batch (saturation_threshold=2000: host steps of frame i + 1 beside the GPU's frame i): {1e3 * best / a.frames:.3f} ms per frame "
      f"({a.frames * h * w / best / 1e6:.0f} Mpx/s), identical to the loop: {np.array_equal(stack_sat, np.stack(outs), equal_nan=True)}")
ref = orc.apply_transfer(frames[0], coords, k, saturation_threshold=2000)
out = t.apply(frames[0], saturation_threshold=2000)
print(f"saturated pixels in the set: {n_sat}; frame 0 against the oracle: max|d|/max|ref| = {np.abs(out - ref).max() / np.abs(ref).max():.2e}")
