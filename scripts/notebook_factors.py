"""Development aid: what separates the notebook workload's per-frame time (200 different frames, every result kept) from a hot loop on one frame."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import regularizepsf_amd as rp  # noqa: E402
from oracle import regpsf_oracle as orc  # noqa: E402

h = w = 512
n = 64
coords, k = orc.synthetic_transfer(h, w, n, alpha=1.0, epsilon=0.1)
t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
frames = [orc.starfield(h, w, 100 + i) for i in range(200)]
t.apply(frames[0])


def run(label, images, keep):
    best = 1e9
    for _ in range(3):
        kept = []
        t0 = time.perf_counter()
        for f in images:
            o = t.apply(f)
            if keep:
                kept.append(o)
        best = min(best, time.perf_counter() - t0)
    print(f"{label}: {1e3 * best / len(images):.3f} ms per frame", flush=True)


run("one frame 200 times, results dropped", [frames[0]] * 200, False)
run("one frame 200 times, results kept", [frames[0]] * 200, True)
run("200 frames, results dropped", frames, False)
run("200 frames, results kept (the notebook)", frames, True)
out = np.empty((h, w))
plan = t._device_plan()
from regularizepsf_amd import _native  # noqa: E402
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for f in frames:
        plan.apply_host(f, _native.PAD_MODES["symmetric"], out=out)
    best = min(best, time.perf_counter() - t0)
print(f"200 frames through Plan.apply_host into one reused result array: {1e3 * best / 200:.3f} ms per frame")
