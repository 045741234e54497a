"""Development aid: per-phase timing of the second-generation patch kernel from an RPSF_STAMPS build.
    RPSF_LIB=devlibs/stamps.so python scripts/stamps2.py [--n 256] [--size 4096] [--overlap planes|direct]"""
import argparse, pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--overlap", default="planes")
a = ap.parse_args()
n, size = a.n, a.size
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
plan = _native.Plan(n, coords)
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
plan.set_overlap_mode(a.overlap)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 1)
st = plan.debug_stamps().astype(np.int64)
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9] + ([10, 11, 12] if a.overlap == "direct" else []) + [13]
names = ["setup+gather", "S1 h0,h1 + X1 h0 + X1w h1", "S2 h0 + X1r h1", "X2 (+S2 h1, K issue)", "freq_a (+orbits)", "freq_b (pointwise)",
         "X2' (+S2' h0)", "X1' h0 (+S2' h1) + X1w' h1", "S1' h0,h1 + X1r' h1"] + (["direct_begin (flag wait)", "RMW + stores issued", "drain + publish", "-"] if a.overlap == "direct" else ["store"])
sel = st[:, idx]
d = np.diff(sel, axis=1) * 0.01  # us
print(f"overlap {a.overlap}: kernel {ker[0]*1e3:.1f} us, total {tot[0]*1e3:.1f} us; patches {len(st)}; per-patch total mean {(st[:,13]-st[:,0]).mean()*0.01:.1f} us")
for i, nm in enumerate(names[: d.shape[1]]):
    print(f"  {nm:34s} mean {d[:, i].mean():6.2f}  p10 {np.percentile(d[:, i],10):6.2f}  p90 {np.percentile(d[:, i],90):6.2f} us")
t0 = st[:, 0].min()
start = (st[:, 0] - t0) * 0.01; end = (st[:, 13] - t0) * 0.01
print("start times (us) percentiles:", np.percentile(start, [0, 25, 50, 75, 100]).round(1))
print("end   times (us) percentiles:", np.percentile(end, [0, 25, 50, 75, 100]).round(1))
order = np.argsort(start)
for r in range((len(st) + 255) // 256):
    s2 = order[r*256:(r+1)*256]
    if len(s2): print(f"  round {r}: n={len(s2)} start {start[s2].mean():6.1f} dur {(end[s2]-start[s2]).mean():6.1f} us")
