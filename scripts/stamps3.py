"""Development aid: per-phase timing of the 256-pixel patch kernel (RPSF_STAMPS build), persistent form included: the gap
between a workgroup's consecutive patches is measured on the timeline of start stamps.
    RPSF_LIB=devlibs/stamps_p.so [RPSF_PERSIST=1] python scripts/stamps3.py [--size 4096]"""
import argparse, pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
a = ap.parse_args()
n, size = 256, a.size
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
plan = _native.Plan(n, coords)
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 1)
st = plan.debug_stamps().astype(np.int64)
last = 13 if (st[:, 13] > st[:, 0]).all() else 12
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, last]
names = ["setup+gather", "S1 + X1", "S2 h0", "X2 fwd", "freq_a", "freq_b", "X2 inv", "X1 inv", "S1 inv", "stores issued", "drain / barrier"]
d = np.diff(st[:, idx], axis=1) * 0.01
print(f"kernel {ker[0]*1e3:.1f} us; patches {len(st)}; stamp 0 -> {last}: mean {(st[:,last]-st[:,0]).mean()*0.01:.1f} us")
for i, nm in enumerate(names):
    print(f"  {nm:18s} mean {d[:, i].mean():6.2f}  p10 {np.percentile(d[:, i],10):6.2f}  p90 {np.percentile(d[:, i],90):6.2f} us")
t0 = st[:, 0].min()
start = np.sort((st[:, 0] - t0) * 0.01)
end = np.sort((st[:, last] - t0) * 0.01)
# the i-th start after the first 256 follows the (i-256)-th end (any workgroup): the gap between patches on one CU
cap = 256 if len(st) > 256 else len(st)
gaps = start[cap:] - end[: len(start) - cap]
print(f"gap between an end stamp and the start stamp that takes its place: mean {gaps.mean():.2f}  p10 {np.percentile(gaps,10):.2f}  p90 {np.percentile(gaps,90):.2f} us")
print("last end", end[-1].round(1), "us")
if (st[:, 14] > 0).all():  # entry stamp (kernel's first instructions of the pass) -> stamp 0 (slot descriptor known, stagger done)
    e0 = (st[:, 0] - st[:, 14]) * 0.01
    print(f"entry -> stamp 0 (kernel arguments, queue draw of a first pass, slot descriptor): mean {e0.mean():.2f}  p10 {np.percentile(e0,10):.2f}  p50 {np.percentile(e0,50):.2f}  p90 {np.percentile(e0,90):.2f} us")
    # exact chains: stamp 15 = the workgroup (block index at dispatch); consecutive patches of one workgroup
    gaps2, e2e = [], []
    for wg in np.unique(st[:, 15]):
        rows = st[st[:, 15] == wg]
        rows = rows[np.argsort(rows[:, 0])]
        for a_, b_ in zip(rows[:-1], rows[1:]):
            gaps2.append((b_[14] - a_[last]) * 0.01)
            e2e.append((b_[0] - a_[0]) * 0.01)
    gaps2, e2e = np.array(gaps2), np.array(e2e)
    print(f"workgroups {len(np.unique(st[:, 15]))}; end stamp -> entry stamp of the same workgroup's next pass (barrier, jump, first instruction fetch): "
          f"mean {gaps2.mean():.2f}  p10 {np.percentile(gaps2,10):.2f}  p50 {np.percentile(gaps2,50):.2f}  p90 {np.percentile(gaps2,90):.2f} us")
    print(f"period of a workgroup (start stamp to start stamp): mean {e2e.mean():.2f}  p10 {np.percentile(e2e,10):.2f}  p90 {np.percentile(e2e,90):.2f} us")
# where the launch's time goes: the start of the first patch, the last patch's end and the kernel's own duration
print(f"first start 0.0, median start of the first 240: {np.percentile(start[:240], 50):.1f} us, last patch end {end[-1]:.1f} us, kernel {ker[0]*1e3:.1f} us "
      f"(launch -> first start and last end -> kernel end together: {ker[0]*1e3 - end[-1]:.1f} us)")
per = np.array([np.sum((start >= lo) & (start < lo + 10)) for lo in range(0, int(end[-1]) + 10, 10)])
print("patch starts per 10 us:", " ".join(str(int(x)) for x in per))
pe = np.array([np.sum((end >= lo) & (end < lo + 10)) for lo in range(0, int(end[-1]) + 10, 10)])
print("patch ends per 10 us:  ", " ".join(str(int(x)) for x in pe))
# rim patches (hang over an image edge: gathered through the np.pad index maps, stored through the clipped path) against interior ones
cc = np.array(coords)
rim = (cc[:, 0] < 0) | (cc[:, 1] < 0) | (cc[:, 0] + n > size) | (cc[:, 1] + n > size)
tot_p = (st[:, last] - st[:, 0]) * 0.01
print(f"rim patches {int(rim.sum())} of {len(rim)}: stamp 0 -> {last} mean {tot_p[rim].mean():.2f} us against {tot_p[~rim].mean():.2f} us for interior ones")
for i, nm in enumerate(names):
    print(f"  {nm:18s} rim {d[rim, i].mean():6.2f}  interior {d[~rim, i].mean():6.2f} us")
