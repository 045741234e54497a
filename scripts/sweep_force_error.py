"""Development aid (library built with -DRPSF3_ABL_FORCE_ERR, RPSF_LIB pointing at it): the sweep kernel's "a wait ran out" report must surface as an error of the
apply that has just been waited for, once, and the plan must be usable afterwards."""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402

n, shape = 32, (256, 320)
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in c) for c in calculate_covering(shape, n)]
k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
img = rng.standard_normal(shape).astype(np.float32)
plan = _native.Plan(n, coords)
plan.set_transfer(k)
for attempt in range(2):
    try:
        plan.apply_host(img, _native.PAD_MODES["symmetric"])
        print("attempt", attempt, ": no error reported")
    except _native.NativeError as e:
        print("attempt", attempt, ": reported:", e)
plan.set_overlap_mode("planes")
out = plan.apply_host(img, _native.PAD_MODES["symmetric"])
print("planes path afterwards: finite", bool(np.isfinite(out).all()))
