"""Step-by-step first run of the sweep kernel (development aid)."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
t0 = time.time()
def say(*a):
    print(f"[{time.time() - t0:6.2f}]", *a, flush=True)
say("start")
from regularizepsf_amd import _native, calculate_covering  # noqa: E402
from tests.helpers import load_apply_case, rel_errors  # noqa: E402
say("imported")
case = sys.argv[1] if len(sys.argv) > 1 else "n32_sym"
fx, coords, k = load_apply_case(case)
n = k.shape[1]
image = np.ascontiguousarray(fx["image"], np.float32)
h, w = image.shape
plan = _native.Plan(n, coords)
say("plan created")
plan.set_transfer(k)
say("transfer set")
plan.set_overlap_mode("sweep")
d_img = _native.DeviceBuffer(image.nbytes).upload(image)
d_out = _native.DeviceBuffer(image.nbytes)
geom = _native.Geometry.whole(h, w, _native.PAD_MODES[str(fx["pad_mode"])])
say("buffers up")
plan.apply_device(d_img.ptr, d_out.ptr, geom)
say("launched")
plan.synchronize()
say("synchronised")
out = d_out.download((h, w))
say("device apply:", rel_errors(out, fx["expected"]))
out2 = plan.apply(image, _native.PAD_MODES[str(fx["pad_mode"])])
say("host apply:", rel_errors(out2, fx["expected"]))
