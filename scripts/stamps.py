"""Development aid: per-phase timing of the patch kernel from an RPSF_STAMPS build.
    RPSF_LIB=devlibs/stamps.so python scripts/stamps.py"""
import pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
n, size = (int(sys.argv[2]) if len(sys.argv) > 2 else 256), 4096
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
plan = _native.Plan(n, coords)
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
plan.set_stagger(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 1)
st = plan.debug_stamps().astype(np.int64)
names = ["setup->load", "stage1", "X1", "stage2", "X2(+K issue)", "last fwd", "pointwise", "last inv", "X2'", "stage2'", "X1'", "stage1'", "store"]
d = np.diff(st[:, :14], axis=1) * 0.01  # us
print(f"kernel {ker[0]*1e3:.1f} us; patches {len(st)}; per-patch total mean {(st[:,13]-st[:,0]).mean()*0.01:.1f} us")
for i, nm in enumerate(names):
    print(f"  {nm:14s} mean {d[:, i].mean():6.2f}  p10 {np.percentile(d[:, i],10):6.2f}  p90 {np.percentile(d[:, i],90):6.2f} us")
t0 = st[:, 0].min()
start = (st[:, 0] - t0) * 0.01; end = (st[:, 13] - t0) * 0.01
print("start times (us) percentiles:", np.percentile(start, [0, 25, 50, 75, 100]).round(1))
print("end   times (us) percentiles:", np.percentile(end, [0, 25, 50, 75, 100]).round(1))
for r in range(5):
    sel = np.argsort(start)[r*256:(r+1)*256]
    if len(sel): print(f"  round {r}: n={len(sel)} start {start[sel].mean():6.1f} dur {(end[sel]-start[sel]).mean():6.1f} us")
