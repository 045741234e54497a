"""Development aid: RPSF_HOST_TRACE=1 python scripts/host_frame_trace.py [size patch] - phase times of single host-array applies."""
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
img32 = (rng.standard_normal((size, size)) * 5 + 100).astype(np.float32)
for in_dt, out_dt in ((np.float32, np.float32), (np.float32, np.float64), (np.float64, np.float64)):
    img = img32.astype(in_dt)
    out = np.zeros((size, size), out_dt)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        plan.apply_host(img, pad, out=out)
        best = min(best, time.perf_counter() - t0)
    print(f"{size}^2 {np.dtype(in_dt).name}->{np.dtype(out_dt).name}: best {1e3*best:.3f} ms", flush=True)
