"""Development aid: device time of every row band of BASELINE config 4 (8192^2 / 256) at world 1/2/4/8, one band at a time on
one GPU (no seam transport: the local apply only), and of 3968^2 vs 4096^2 frames (whole rounds vs a partial last round).
    python scripts/band_times.py [reserved CUs]   (as ShardedApply reserves them for the seam exchange when it has a communicator)"""
import json, pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
from regularizepsf_amd.sharding import ShardedApply

rng = np.random.default_rng(0)
n = 256
for size in (3968, 4096):
    coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
    plan = _native.Plan(n, coords)
    k = (rng.standard_normal((len(coords), n, n), dtype=np.float32) + 1j * rng.standard_normal((len(coords), n, n), dtype=np.float32)).astype(np.complex64)
    plan.set_transfer(k)
    img = rng.standard_normal((size, size), dtype=np.float32)
    d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
    geom = _native.Geometry.whole(size, size, 1)
    plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 5)
    tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 40)
    print(json.dumps({"frame": size, "patches": len(coords), "apply_us": round(float(np.median(ker)) * 1e3, 1)}))
    plan.close(); d_img.free(); d_out.free()

h = w = 8192
coords = [tuple(int(v) for v in t) for t in calculate_covering((h, w), n)]
kk = (rng.standard_normal((585, n, n), dtype=np.float32) + 1j * rng.standard_normal((585, n, n), dtype=np.float32)).astype(np.complex64)
image = rng.standard_normal((h, w), dtype=np.float32)
base = None
for world in (1, 2, 4, 8):
    times = []
    for rank in range(world):
        sh = ShardedApply(coords, lambda idx: np.resize(kk, (len(idx), n, n)), n, h, w, rank, world, 0, None)
        if len(sys.argv) > 1:
            sh.plan.set_reserved_cus(int(sys.argv[1]))
        b = sh.band
        sh.upload_rows(image[b.image_row0:b.image_row0 + b.image_rows])
        sh.plan.apply_device_timed(sh.d_img.ptr, sh.d_out.ptr, sh.geometry, 3)
        tot, ker = sh.plan.apply_device_timed(sh.d_img.ptr, sh.d_out.ptr, sh.geometry, 20)
        times.append((len(b.patch_index), round(float(np.median(ker)) * 1e3, 1)))
        sh.plan.close(); sh.d_img.free(); sh.d_out.free(); sh.d_recv.free()
    slowest = max(t for _, t in times)
    base = base or slowest
    print(json.dumps({"world": world, "bands (patches, us)": times, "slowest_us": slowest, "speedup_vs_1": round(base / slowest, 2)}))
