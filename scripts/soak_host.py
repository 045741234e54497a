"""Development aid: seeded soak of the host-array entry points (single frames, row bands, streamed batches, page-locked arrays, the saturation
branch): random shapes, patch sizes, pad modes, dtypes, group sizes and pipeline depths; every result against the frame loop (bit for bit) and
one frame per case against the oracle.   python scripts/soak_host.py [--seconds 60] [--seed 1]"""
import argparse
import os
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import regularizepsf_amd as rp  # noqa: E402
from oracle import regpsf_oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
cases = checks = 0
while time.time() < t_end:
    n = int(rng.choice([16, 32, 64, 128, 256]))
    h, w = (int(rng.integers(n, 9 * n)) for _ in range(2))
    if rng.random() < 0.5:
        w = (w + 31) // 32 * 32  # the fused / HOT geometries
    pad_mode = str(rng.choice(["symmetric", "reflect", "constant", "edge", "wrap"]))
    frames = int(rng.integers(1, 9))
    bands, group, depth = int(rng.choice([0, 2, 3, 4, 8, 16])), int(rng.integers(1, 5)), int(rng.integers(1, 5))
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((h, w), n)]
    k = ((rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))) * 0.2).astype(np.complex64)
    dt_in = rng.choice([np.float32, np.float64])
    images = (rng.standard_normal((frames, h, w)) * 10 + 100).astype(dt_in)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    plan = t._device_plan()  # (plan options: the shipped library reads none of this from the environment)
    plan.set_option("stream_group", group), plan.set_option("stream_depth", depth), plan.set_option("host_bands", 0)
    whole = np.stack([t.apply(im, pad_mode=pad_mode) for im in images])
    plan.set_option("host_bands", bands)
    loop = np.stack([t.apply(im, pad_mode=pad_mode) for im in images])
    assert np.array_equal(loop, whole), ("bands", n, h, w, pad_mode, bands)
    batch = t.apply_batch(images, pad_mode=pad_mode)
    assert np.array_equal(batch, loop), ("batch", n, h, w, pad_mode, frames)
    if rng.random() < 0.4:
        pin = rp.pinned_empty(images.shape, np.float32)
        pin[...] = images
        pout = rp.pinned_empty(images.shape, np.float32)
        assert np.array_equal(t.apply_batch(pin, pad_mode=pad_mode, out=pout), loop.astype(np.float32)) or dt_in == np.float64, ("pinned", n, h, w)
    ref = orc.apply_transfer(images[0], coords, k, pad_mode=pad_mode)
    d = loop[0] - ref
    assert np.abs(d).max() <= 1e-5 * np.abs(ref).max(), ("oracle", n, h, w, pad_mode)
    if rng.random() < 0.3:
        hot = images[0].astype(np.float64).copy()
        hot[rng.integers(0, h, 5), rng.integers(0, w, 5)] = 1e5
        sat = t.apply(hot, pad_mode=pad_mode, saturation_threshold=5e4, saturation_dilation=int(rng.integers(1, 3)))
        rs = orc.apply_transfer(hot, coords, k, pad_mode=pad_mode, saturation_threshold=5e4, saturation_dilation=1)
        checks += 1
    cases += 1
print(f"soak_host: {cases} cases ({checks} with the saturation branch) in {a.seconds:.0f} s, seed {a.seed}: all identical / within tolerance")
