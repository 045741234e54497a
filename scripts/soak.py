"""One-off robustness soak on the GPU: the seeded sweep of tests/test_gpu_parity.py over many more seeds, plus the same
sweep with 256-px patches (two-pass special slot) on small odd-shaped images.  python scripts/soak.py [first] [last]"""
import pathlib, sys
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import regularizepsf_amd as rp
from oracle import regpsf_oracle as orc
from tests.test_gpu_parity import test_randomized_shapes_pads_and_corner_lists as sweep, check

first, last = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
bad = []
for seed in range(first, last):
    try:
        sweep(seed)
    except AssertionError as e:  # noqa: PERF203
        bad.append(("sweep", seed, str(e)[:200]))
for seed in range(first, first + (last - first) // 4):
    rng = np.random.default_rng(5000 + seed)
    n = 256
    h, w = (int(v) for v in rng.integers(n // 2 + 1, 3 * n + 17, size=2))
    pad_mode = str(rng.choice(["symmetric", "reflect", "edge", "wrap", "constant"]))
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((h, w), n)]
    if seed % 3 == 1:
        keep = rng.random(len(coords)) < 0.7
        keep[0] = True
        coords = [c for c, k_ in zip(coords, keep) if k_]
    elif seed % 3 == 2:
        dr, dc = (int(v) for v in rng.integers(-n // 4, n // 4 + 1, size=2))
        coords = [(r + dr + int(rng.integers(0, 3)), c + dc) for r, c in coords]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = (rng.standard_normal((h, w)) * 20 + 50).astype(np.float32)
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply(image, pad_mode=pad_mode)
    try:
        check(out, orc.apply_transfer(image, coords, k, pad_mode=pad_mode))
    except AssertionError as e:
        bad.append(("n256", seed, str(e)[:200]))
print(f"soak: seeds {first}..{last - 1}: {len(bad)} failures", bad[:5])
sys.exit(1 if bad else 0)
