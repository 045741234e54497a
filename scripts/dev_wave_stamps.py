"""Development aid: per-wave phase timing of the N=256 patch kernel.  Needs a library built with -DRPSF_STAMPS -DRPSF_WAVE_STAMPS: STAMP()
records lane 0 of every wave at ((patch * 8 + wave) * 16 + i) and the stamp buffer is 8x larger.
    RPSF_LIB=devlibs/wave_stamps.so python scripts/dev_wave_stamps.py"""
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
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
geom = _native.Geometry.whole(size, size, 1)
plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 1)
out = np.zeros(len(coords) * 8 * 16, np.uint64)
_native.check(_native.lib().rpsf_plan_debug_stamps(plan._handle, out.ctypes.data_as(ctypes.c_void_p), out.size))
st = out.reshape(len(coords), 8, 16).astype(np.int64)
names = ["gather", "S1+X1", "S2h0", "X2fwd", "freq_a", "freq_b", "X2inv", "X1inv", "S1inv", "stores", "-", "drain", "-"]  # second-generation kernel's stamps 0 .. 12
print(f"kernel {ker[0]*1e3:.1f} us")
t0 = st[:, :, 0].min(axis=1, keepdims=True)
rel = (st[:, :, :14] - t0[:, :, None]) * 0.01
print("mean time (us since the patch's first wave started) at which each wave passes each stamp")
print("wave " + " ".join(f"{i:6d}" for i in range(14)))
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{rel[:, w, i].mean():6.2f}" for i in range(14)))
d = np.diff(st[:, :, :14], axis=2) * 0.01
print("mean phase duration per wave (us)")
print("wave " + " ".join(f"{nm:>8s}" for nm in names))
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{d[:, w, i].mean():8.2f}" for i in range(13)))
