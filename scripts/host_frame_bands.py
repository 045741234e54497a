"""Development aid: python scripts/host_frame_bands.py [size patch] - one host-array apply end to end for band counts 0 (no bands), 4 ... 8,
pageable float32 -> float64 / float32 and page-locked float32 -> float32, best and median of 12, each result checked against the first."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from oracle import regpsf_oracle as orc  # noqa: E402
from regularizepsf_amd import _native  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((size, size), n)]
rng = np.random.default_rng(5)
k = ((rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))) * 0.1).astype(np.complex64)
plan = _native.Plan(n, coords)
plan.set_transfer(k)
pad = _native.PAD_MODES["symmetric"]
_native.bind_to_device_node(0)
img = (rng.standard_normal((size, size)) * 5 + 100).astype(np.float32)
ref = None
print("pcie probe:", {k: round(v, 3) for k, v in _native.pcie_probe(img.nbytes).items()}, flush=True)
for bands in (0, 4, 6, 8, 10, 12, 16, -1):
    plan.set_option("host_bands", bands)
    row = []
    for label, out_dt, pinned in (("f32->f64", np.float64, False), ("f32->f32", np.float32, False), ("pinned f32->f32", np.float32, True)):
        if pinned:
            a, b = _native.pinned_empty((size, size), np.float32), _native.pinned_empty((size, size), np.float32)
            a[...] = img
        else:
            a, b = img, np.zeros((size, size), out_dt)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            plan.apply_host(a, pad, out=b)
            ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = np.array(b, np.float64)
        same = bool(np.array_equal(np.asarray(b, np.float64), ref))
        row.append(f"{label}: best {1e3*min(ts):.3f} median {1e3*sorted(ts)[6]:.3f} same={same}")
    print(f"bands {bands:2d} | " + " | ".join(row), flush=True)
