"""One-off robustness soak of the sweep kernel (N = 16, 32, 64) on the GPU: random frame shapes, pad modes, dtypes of the corner lists' origin, region counts and
batches; every case: repeated applies bit-stable, every cut bit-identical, the colour-plane path within 1e-5, one case in four against the float64 oracle.
    python scripts/soak_sweep.py [--seconds 120] [--seed 1]"""
import argparse
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from oracle import regpsf_oracle as orc  # noqa: E402
from regularizepsf_amd import _native, calculate_covering  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
cases = oracle_checks = 0
while time.time() < t_end:
    n = int(rng.choice([16, 32, 64]))
    h, w = (int(rng.integers(n, 24 * n)) for _ in range(2))
    if rng.random() < 0.4:
        w = (w + 3) // 4 * 4  # the 16-byte paths
    pad = str(rng.choice(["symmetric", "reflect", "constant", "edge", "wrap"]))
    coords = [tuple(int(v) for v in c) for c in calculate_covering((h, w), n)]
    k = ((rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))) * 0.3).astype(np.complex64)
    frames = int(rng.integers(1, 4))
    images = (rng.standard_normal((frames, h, w)) * 10 + 100).astype(np.float32)
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    mode = _native.PAD_MODES[pad]
    if plan.sweep_info()["regions"] == 0:
        continue  # (a single row or column of patches: the colour-plane path)
    base = [plan.apply(im, mode) for im in images]
    for _ in range(3):
        assert all(np.array_equal(plan.apply(im, mode), b) for im, b in zip(images, base)), ("repeat", n, h, w, pad)
    for target in (int(rng.integers(1, 9)), int(rng.integers(9, 64)), int(rng.integers(64, 600))):
        plan.set_sweep_regions(target)
        assert all(np.array_equal(plan.apply(im, mode), b) for im, b in zip(images, base)), ("cut", n, h, w, pad, target, plan.sweep_info())
    if frames > 1:
        stack = plan.apply_batch(images, mode) if hasattr(plan, "apply_batch") else None
        if stack is not None:
            assert all(np.array_equal(s, b) for s, b in zip(stack, base)), ("batch", n, h, w, pad)
    plan.set_overlap_mode("planes")
    other = plan.apply(images[0], mode).astype(np.float64)
    scale = np.abs(other).max()
    assert np.abs(other - base[0]).max() <= 1e-5 * scale, ("planes", n, h, w, pad)
    if cases % 4 == 0:
        ref = orc.apply_transfer(images[0], coords, k, pad_mode=pad)
        assert np.abs(base[0] - ref).max() <= 1e-5 * np.abs(ref).max(), ("oracle", n, h, w, pad)
        oracle_checks += 1
    cases += 1
print(f"soak_sweep: {cases} cases ({oracle_checks} against the oracle) in {a.seconds:.0f} s, seed {a.seed}: repeats and cuts bit-identical, planes and oracle within 1e-5")
