"""First contact of the third-generation (sweep) kernel with the GPU: goldens against the reference's outputs, then timings
against the colour-plane path at the sizes VERDICT round 5 names."""
import json
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402
from tests.helpers import APPLY_CASES, load_apply_case, rel_errors  # noqa: E402

for case in [c for c in APPLY_CASES if c[3] <= 64 and c[7] in _native.PAD_MODES]:
    fx, coords, k = load_apply_case(case[0])
    image = np.ascontiguousarray(fx["image"], np.float32)
    outs = {}
    for mode in ("sweep", "planes"):
        plan = _native.Plan(case[3], coords)
        plan.set_transfer(k)
        plan.set_overlap_mode(mode)
        outs[mode] = plan.apply(image, _native.PAD_MODES[str(fx["pad_mode"])])
        again = plan.apply(image, _native.PAD_MODES[str(fx["pad_mode"])])
        print(case[0], mode, rel_errors(outs[mode], fx["expected"]), "repeat identical:", bool(np.array_equal(again, outs[mode])), flush=True)

rng = np.random.default_rng(0)
for n, size in ((32, 512), (64, 512), (16, 512), (32, 2048), (32, 4096), (64, 4096), (16, 4096), (64, 2048)):
    coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
    k = np.empty((len(coords), n, n), np.complex64)
    k.real = rng.standard_normal(k.shape, dtype=np.float32)
    k.imag = rng.standard_normal(k.shape, dtype=np.float32)
    img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
    d_img = _native.DeviceBuffer(img.nbytes).upload(img)
    d_out = _native.DeviceBuffer(img.nbytes)
    geom = _native.Geometry.whole(size, size, 1)
    res = {}
    for mode in ("sweep", "planes"):
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        plan.set_overlap_mode(mode)
        plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 3)
        tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 20)
        res[mode] = d_out.download((size, size)).copy()
        alg = plan.transfer_bytes + 2 * img.nbytes
        print(json.dumps({"n": n, "size": size, "mode": mode, "patches": len(coords), "total_ms_med": round(float(np.median(tot)), 4),
                          "total_ms_min": round(float(tot.min()), 4), "frac": round(float(alg / np.median(tot) / 1e6 / 8000), 4)}), flush=True)
    d = np.abs(res["sweep"].astype(np.float64) - res["planes"]).max() / np.abs(res["planes"]).max()
    print("  sweep vs planes, max rel:", d, flush=True)
