import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import regularizepsf_amd as rp
from regularizepsf_amd import _native
size, n = 4096, 256
coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((size, size), n)]
rng = np.random.default_rng(0)
k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
plan = _native.Plan(n, coords); plan.set_transfer(k)
pad = _native.PAD_MODES["symmetric"]
img = (rng.standard_normal((size, size)) * 5 + 100).astype(np.float32)
out = np.empty(img.shape, np.float64)
import ctypes
lib = _native.lib()
def call(o, is64):
    _native.check(lib.rpsf_apply_host(plan._handle, img.ctypes.data_as(ctypes.c_void_p), 0, size, size, pad, 0.0, o.ctypes.data_as(ctypes.c_void_p), is64))
call(out, 1)
for _ in range(3):
    t0 = time.perf_counter(); call(out, 1); t1 = time.perf_counter()
    fresh = np.empty(img.shape, np.float64); t2 = time.perf_counter(); call(fresh, 1); t3 = time.perf_counter()
    o32 = np.empty(img.shape, np.float32); o32[:] = 0; t4 = time.perf_counter(); call(o32, 0); t5 = time.perf_counter()
    del fresh; t6 = time.perf_counter()
    print(f"warm f64 out {1e3*(t1-t0):.2f} ms | fresh f64 out {1e3*(t3-t2):.2f} ms | warm f32 out {1e3*(t5-t4):.2f} ms | free {1e3*(t6-t5):.2f} ms")
