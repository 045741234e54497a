"""Development aid, under `rocprofv3 --kernel-trace --memory-copy-trace`: a few host-array applies of one frame (argv: size patch bands pinned_in(0/1) pinned_out(0/1) f64out(0/1)),
so that the copies and launches of the last one can be laid on one time axis (scripts/host_frame_timeline_read.py)."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from oracle import regpsf_oracle as orc  # noqa: E402
from regularizepsf_amd import _native  # noqa: E402

size, n, bands, pin_in, pin_out, f64 = (int(v) for v in (sys.argv[1:] + ["4096", "256", "8", "0", "0", "1"][len(sys.argv) - 1:]))
coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((size, size), n)]
rng = np.random.default_rng(5)
k = ((rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))) * 0.1).astype(np.complex64)
plan = _native.Plan(n, coords)
plan.set_transfer(k)
plan.set_option("host_bands", bands)
_native.bind_to_device_node(0)
img = (rng.standard_normal((size, size)) * 5 + 100).astype(np.float32)
a, b = img, np.zeros((size, size), np.float64 if f64 else np.float32)
if pin_in:
    a = _native.pinned_empty((size, size), np.float32)
    a[...] = img
if pin_out:
    b = _native.pinned_empty((size, size), np.float32)
for i in range(6):
    time.sleep(0.02)  # a gap on the time axis between applies
    t0 = time.perf_counter()
    plan.apply_host(a, _native.PAD_MODES["symmetric"], out=b)
    print(f"apply {i}: {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
