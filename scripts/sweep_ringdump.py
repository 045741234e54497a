"""Development aid: the region's LDS ring after every job of region 0 (build with -DRPSF3_DUMP -DRPSF3_RINGDUMP -DRPSF3_ONEWAVE) against the emulator's."""
import ctypes
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from regularizepsf_amd import _native  # noqa: E402
from regularizepsf_amd._native import lib, check, _ptr  # noqa: E402
from tests.helpers import load_apply_case  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "n32_sym"
fx, coords, k = load_apply_case(case)
n = k.shape[1]
image = np.ascontiguousarray(fx["image"], np.float32)
h, w = image.shape
os.environ["EMU3_RING_DUMP"] = "/tmp/emu3_ring.bin"
emu = ctypes.CDLL(str(ROOT / "tests" / "emu" / "libemu3.so"))
c = np.ascontiguousarray(np.array(coords, np.int32))
kk = np.ascontiguousarray(k, np.complex64)
out = np.zeros((h, w), np.float32)
vp = ctypes.c_void_p
modes = {"constant": 0, "symmetric": 1, "reflect": 2, "edge": 3, "wrap": 4}
emu.emu3_apply(n, len(coords), c.ctypes.data_as(vp), h, w, modes[str(fx["pad_mode"])], ctypes.c_float(0.0), image.ctypes.data_as(vp),
               kk.ctypes.data_as(vp), out.ctypes.data_as(vp), 256, 0, 1, None)
rp = {16: 4 * 128 + 8 + 4, 32: 2 * 128 + 16 + 4, 64: 2 * 128 + 32 + 4}[n]
ref = np.fromfile("/tmp/emu3_ring.bin", np.float32).reshape(-1, n, rp)
plan = _native.Plan(n, coords)
plan.set_transfer(k)
plan.set_overlap_mode("sweep")
plan.apply(image, _native.PAD_MODES[str(fx["pad_mode"])])
buf = np.zeros(512 * 8 * 8 * 16, np.uint64)
check(lib().rpsf_plan_debug_stamps(plan._handle, _ptr(buf), buf.size))
got = buf.view(np.float32)[65536:65536 + ref.size].reshape(ref.shape)
for j in range(ref.shape[0]):
    known = ~np.isnan(ref[j])
    d = np.where(known, np.abs(got[j] - np.nan_to_num(ref[j])), 0.0)
    scale = np.abs(np.nan_to_num(ref[j])).max()
    badrows = np.where(d.max(axis=1) > 1e-4 * scale)[0]
    print("after job", j, "known words", int(known.sum()), "max diff", float(d.max()), "scale", float(scale), "bad ring rows", badrows[:20])
    if len(badrows):
        r = badrows[0]
        cols = np.where(d[r] > 1e-4 * scale)[0]
        print("    row", r, "bad cols", cols[:12], "...", cols[-4:], "gpu", got[j][r, cols[:4]], "emu", ref[j][r, cols[:4]])
img_out = plan.apply(image, _native.PAD_MODES[str(fx["pad_mode"])])
# after job 3 (B of lattice row 1, ring half 1 = image rows 0..15; ring column x = image column x - 32 for N = 32 / n32_sym)
if n == 32:
    band = got[3][16:32]
    for r in (0, 1, 2, 3):
        print("image row", r, "cols 0..7:", img_out[r, :8], "\n   ring:", band[r, 32:40], "\n   expected:", fx["expected"][r, :8])
    d = np.abs(img_out[:16, :64] - band[:, 32:96])
    print("rows 0..15, cols 0..63: image vs ring max diff", float(d.max()))
    print("per-row max diff", d.max(axis=1))
    print("per-col max diff (first 32)", d.max(axis=0)[:32])
