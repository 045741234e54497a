"""Development aid: do small workgroups of a second kernel run beside the persistent patch kernel?  Needs a build with
-DRPSF_STAMPS -DRPSF_DEV_PROBE -DRPSF_VGPR_CAP=124 (patch kernel at 248 registers: 16 left per SIMD).
    RPSF_LIB=devlibs/probe.so python scripts/probe_coresidency.py"""
import ctypes, pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
n, size = 256, 4096
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
plan = _native.Plan(n, coords)
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
plan.set_transfer(k)
other = _native.Plan(16, [(0, 0)])  # only for its stream
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
lib = _native.lib()
lib.rpsf_dev_probe.restype = ctypes.c_int
lib.rpsf_dev_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
blocks = 256
d_probe = _native.DeviceBuffer(blocks * 16)
for order in ("patch first", "probe first"):
    for rep in range(3):
        plan.apply_device(d_img.ptr, d_out.ptr, geom); plan.synchronize()
        d_probe.upload(np.zeros(blocks * 2, np.uint64))
        if order == "probe first":
            _native.check(lib.rpsf_dev_probe(0, blocks, 150, d_probe.ptr, other.stream))
        plan.apply_device(d_img.ptr, d_out.ptr, geom)
        if order == "patch first":
            _native.check(lib.rpsf_dev_probe(0, blocks, 150, d_probe.ptr, other.stream))
        plan.synchronize()
        st = plan.debug_stamps().astype(np.int64)
        pr = d_probe.download((blocks, 2), np.uint64)
        t0, t1 = st[:, 0].min(), st[:, 12].max()
        start = pr[:, 0].astype(np.int64)
        inside = ((start > t0 + 500) & (start < t1)).sum()
        early = (start <= t0 + 500).sum()
        cu = {(int(v) >> 32, (int(v) >> 8) & 0xF, (int(v) >> 13) & 0x7) for v in pr[:, 1]}  # (xcc, cu_id, se_id)
        print(f"{order}: patch kernel {0.01*(t1-t0):.1f} us; probe workgroups started before it {early}, inside it {inside}, after it {blocks-inside-early}; "
              f"distinct (xcc, cu, se) {len(cu)}; probe starts rel. to kernel start (us): p10 {0.01*(np.percentile(start,10)-t0):.1f} p50 {0.01*(np.percentile(start,50)-t0):.1f} p90 {0.01*(np.percentile(start,90)-t0):.1f}")
