"""Phase timestamps of the sweep kernel (development build with -DRPSF3_STAMPS):
    python -m regularizepsf_amd.build --target=devlibs/librpsf_stamps3.so --only=k3_16,k3_32,k3_64,rpsf -DRPSF3_STAMPS; RPSF_LIB=$PWD/devlibs/librpsf_stamps3.so python scripts/stamps_sweep.py --n 32 --size 4096
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
         "T2", "repack + inverse row FFT + window", "dependency wait", "accumulate", "flush", "flag"]
print(f"N = {a.n}, {a.size}^2: {st.shape[0]} regions stamped; times in us (10 ns ticks), mean over regions, waves and job slots 1..7")
ok = (st[:, :, :, :15] != 0).all(axis=3)
ok[:, :, 0] = False
d = np.diff(st[:, :, :, :15], axis=3)[ok] / 100.0
if d.shape[0] == 0:
    raise SystemExit("no wave ran a second job (a frame this small is one job per wave deep): nothing to average")
print(f"jobs: {d.shape[0]}, mean job time {d.sum(axis=1).mean():.2f} us")
for i in range(14):
    print(f"  {names[i]:42s} {d[:, i].mean():6.2f}   (p90 {np.percentile(d[:, i], 90):6.2f})")
first = st[:, :, 0, 0].astype(np.float64)
first[first == 0] = np.nan
last = st[:, :, :, 14].max(axis=(1, 2)).astype(np.float64)
print("stamped span per region (first draw -> last stamped flag), mean us:", float(np.nanmean(last - np.nanmin(first, axis=1)) / 100.0))
