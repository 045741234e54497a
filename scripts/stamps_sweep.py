"""Phase timestamps of the sweep kernel (development build with -DRPSF3_STAMPS):
    RPSF_LIB=regularizepsf_amd/variants/librpsf_stamps3.so python scripts/stamps_sweep.py --n 32 --size 4096
Prints, per phase, the mean time over the stamped jobs (the first 8 jobs of every wave of every region) in microseconds."""
import argparse
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402
from regularizepsf_amd._native import lib, check, _ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--size", type=int, default=4096)
a = ap.parse_args()
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((a.size, a.size), a.n)]
plan = _native.Plan(a.n, coords)
k = np.empty((len(coords), a.n, a.n), np.complex64)
k.real = rng.standard_normal(k.shape, dtype=np.float32)
k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
img = (100 + 5 * rng.standard_normal((a.size, a.size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img)
d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(a.size, a.size, 1)
for _ in range(3):
    plan.apply_device(d_img.ptr, d_out.ptr, geom)
plan.synchronize()
buf = np.zeros(4096 * 8 * 8 * 16, np.uint64)
check(lib().rpsf_plan_debug_stamps(plan._handle, _ptr(buf), buf.size))
st = buf.reshape(-1, 8, 8, 16).astype(np.int64)
st = st[st[:, 0, 0, 0] != 0]
names = ["draw", "descriptor", "gather issue", "T0 (+ pixel latency)", "window + row FFT + unpack", "T1", "column FFT", "x K (+ K latency)", "inverse column FFT",
         "T2", "repack + inverse row FFT", "dependency wait", "accumulate", "flush", "flag"]
print(f"N = {a.n}, {a.size}^2: {st.shape[0]} regions stamped; times in us (10 ns ticks), mean over regions and waves")
start = st[:, :, :, 0].min(axis=(1, 2), keepdims=True)
for slot in range(8):
    row = st[:, :, slot, :]
    ok = row[:, :, 14] != 0
    if not ok.any():
        continue
    d = np.diff(row, axis=2)[ok] / 100.0
    begin = (row[:, :, 0] - start[:, :, 0])[ok] / 100.0
    print(f"job slot {slot}: starts at {begin.mean():7.2f}, takes {d.sum(axis=1).mean():7.2f}: " + ", ".join(f"{names[i]} {d[:, i].mean():.2f}" for i in range(14)))
end = st[:, :, :, 14].max(axis=(1, 2))
print("last stamped job ends at (mean over regions)", ((end - start[:, 0, 0]) / 100.0).mean())
