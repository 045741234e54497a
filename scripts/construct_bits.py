"""Development aid: K2 (rpsf_build_transfer) against the reference's own transfer kernels of tests/golden/construct.npz, bit for bit - how many values
differ, and by how many float32 steps at most - and the time of one construct of 1089 x 256^2 bins on device-resident spectra."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native  # noqa: E402

fx = np.load(pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden" / "construct.npz")
coords = [(i, 2 * i) for i in range(6)]
from oracle import regpsf_oracle as orc  # noqa: E402

s_fft = fx["rand_fft_float32"] if "rand_fft_float32" in fx else None
for alpha in (0.5, 1.0, 2.0, 3.0):
    for eps in (0.1, 0.01):
        ref = fx[f"rand_float32_a{alpha}_e{eps}"]
        s = orc.psf_fft(fx["rand_values_s"].astype(np.float32)).astype(np.complex64)
        t = orc.psf_fft(fx["rand_values_t"].astype(np.float32)).astype(np.complex64)
        k = _native.build_transfer(s, t, alpha, eps)
        a, b = k.view(np.float32).ravel(), ref.view(np.float32).ravel()
        ok = np.isfinite(a) & np.isfinite(b)
        ulps = np.abs(a[ok].view(np.int32).astype(np.int64) - b[ok].view(np.int32).astype(np.int64))
        print(f"alpha {alpha} eps {eps}: {int((ulps != 0).sum())} of {ulps.size} float32 values differ from the reference's, at most {int(ulps.max())} steps, "
              f"max relative difference {float(np.abs(k - ref)[np.isfinite(ref)].max() / np.abs(ref[np.isfinite(ref)]).max()):.2e}", flush=True)
rng = np.random.default_rng(0)
n, count = 256, 1089
s = (rng.standard_normal((count, n, n)) + 1j * rng.standard_normal((count, n, n))).astype(np.complex64)
t = (rng.standard_normal((count, n, n)) + 1j * rng.standard_normal((count, n, n))).astype(np.complex64)
for _ in range(3):
    t0 = time.perf_counter()
    _native.build_transfer(s, t, 3.0, 0.1)
    print(f"rpsf_build_transfer of {count} x {n}^2 bins from host arrays: {1e3 * (time.perf_counter() - t0):.1f} ms")
