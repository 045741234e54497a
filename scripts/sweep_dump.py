"""Development aid: the values of every lane after each phase of job 0 of region 0 - GPU (build with -DRPSF3_DUMP) against the emulator."""
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
# emulator
os.environ["EMU3_DUMP"] = "/tmp/emu3_dump.bin"
os.environ["EMU3_DUMP_JOB"] = sys.argv[2] if len(sys.argv) > 2 else "0"
emu = ctypes.CDLL(str(ROOT / "tests" / "emu" / "libemu3.so"))
c = np.ascontiguousarray(np.array(coords, np.int32))
kk = np.ascontiguousarray(k, np.complex64)
out = np.zeros((h, w), np.float32)
vp = ctypes.c_void_p
modes = {"constant": 0, "symmetric": 1, "reflect": 2, "edge": 3, "wrap": 4}
rc = emu.emu3_apply(n, len(coords), c.ctypes.data_as(vp), h, w, modes[str(fx["pad_mode"])], ctypes.c_float(0.0), image.ctypes.data_as(vp),
                    kk.ctypes.data_as(vp), out.ctypes.data_as(vp), 256, 0, 1, None)
ref = np.fromfile("/tmp/emu3_dump.bin", np.float32).reshape(-1, 64, 8, 2)
print("emulator rc", rc, "phases", ref.shape[0])
plan = _native.Plan(n, coords)
plan.set_transfer(k)
plan.set_overlap_mode("sweep")
got_img = plan.apply(image, _native.PAD_MODES[str(fx["pad_mode"])])
buf = np.zeros(16 * 64 * 8, np.uint64)
check(lib().rpsf_plan_debug_stamps(plan._handle, _ptr(buf), buf.size))
got = buf.view(np.float32).reshape(16, 64, 8, 2)
for e, kph, name in ((0, 3, "after T0"), (1, 5, "after T1"), (2, 9, "after T2"), (3, 10, "after the second window")):
    d = np.abs(got[kph] - ref[e])
    scale = np.abs(ref[e]).max()
    print(name, "max diff", d.max(), "scale", scale, "bad lanes", np.where(d.max(axis=(1, 2)) > 1e-4 * scale)[0][:16])
    if d.max() > 1e-4 * scale:
        l = int(np.where(d.max(axis=(1, 2)) > 1e-4 * scale)[0][0])
        print("   lane", l, "gpu", got[kph][l, :4].ravel(), "emu", ref[e][l, :4].ravel())
for ph in range(16):
    print("gpu phase", ph, "absmax", float(np.abs(got[ph]).max()), "first", got[ph][0, :2].ravel())
print("phase-0 slots (w_re, w_im, win[0], win[1], lane, flags, kslot, ring_col) lanes 0..3, 17:")
for l in (0, 1, 2, 3, 17):
    print("  ", got[0].reshape(64, 16)[l, :8])
meta = got[0].reshape(64, 16)[0, 8:16]
print("job 3: row0 col0 oc0 oc1 flags hs ring_col dep0 =", meta)
row0, col0 = int(meta[0]), int(meta[1])
fu = got[11:15].reshape(-1)[: 64 * 32].reshape(64, 8, 4)  # [lane][unit][4]
exp = fx["expected"]
bad = 0
for lane_ in range(64):
    u, hf = lane_ & 31, lane_ >> 5
    for i in range(8):
        r, cc = row0 + 2 * i + hf, col0 + 4 * u
        if 0 <= r < h and 0 <= cc and cc + 4 <= w and int(meta[2]) <= cc and cc + 4 <= int(meta[3]):
            d = np.abs(fu[lane_, i] - exp[r, cc:cc + 4]).max()
            if d > 1e-3 * np.abs(exp).max():
                bad += 1
                if bad < 6:
                    print("  flush value differs at row", r, "col", cc, "gpu", fu[lane_, i], "expected", exp[r, cc:cc + 4], "host-visible result", got_img[r, cc:cc + 4])
print("flush units that differ from the expected output:", bad)
from tests.helpers import rel_errors  # noqa: E402
print("image against the golden:", rel_errors(got_img, fx["expected"]))
d = np.abs(got_img.astype(np.float64) - fx["expected"])
badpix = d > 1e-4 * np.abs(fx["expected"]).max()
print("bad pixels", int(badpix.sum()), "of", badpix.size)
if badpix.any():
    rows = np.where(badpix.any(axis=1))[0]; cols = np.where(badpix.any(axis=0))[0]
    print(" rows", rows.min(), rows.max(), "cols", cols.min(), cols.max())
    hh = n // 2
    hb, wb = h // hh, w // hh
    blk = badpix[: hb * hh, : wb * hh].reshape(hb, hh, wb, hh).any(axis=(1, 3))
    for r in range(hb):
        print("   ", "".join("#" if x else "." for x in blk[r]))
print("job-3 flush map (rows x units; # = differs, . = ok, blank = outside the image / not owned):")
for rr in range(16):
    line = ""
    for u in range(32):
        lane_ = (rr & 1) * 32 + u
        i = rr >> 1
        r, cc = row0 + rr, col0 + 4 * u
        if 0 <= r < h and 0 <= cc and cc + 4 <= w and int(meta[2]) <= cc and cc + 4 <= int(meta[3]):
            d = np.abs(fu[lane_, i] - exp[r, cc:cc + 4]).max()
            line += "#" if d > 1e-3 * np.abs(exp).max() else "."
        else:
            line += " "
    print("   ", line)

# ring read-back: phase 12 = v before the accumulate (imaginary parts = lower half), phase 13 = what the lane reads back from its ring row
v12 = got[12]  # [lane][8][2]
rb = got[13]
print("read-back of the lane's own lower-half row: max |ring - v.y| =", float(np.abs(rb[:, :, 0] - v12[:, :, 1]).max()), " ring offsets lane 0:", rb[0, :, 1], "lane 1:", rb[1, :2, 1], "lane 9:", rb[9, :2, 1])
ft = got[15].reshape(-1)[: 64 * 16].reshape(64, 4, 4)
H2 = n // 2
bad = 0
for lane_ in range(64):
    u, hf = lane_ & 31, lane_ >> 5
    for i in range(min(4, H2 // 2)):
        row = 2 * i + hf
        for d_ in range(4):
            col = 4 * u + d_
            q_, c_ = col // n, col % n
            src_lane = q_ * H2 + row
            if c_ < 8:
                if abs(ft[lane_, i, d_] - v12[src_lane, c_, 1]) > 1e-4 * np.abs(v12).max():
                    bad += 1
                    if bad < 6:
                        print("  flush-layout read differs: row", row, "col", col, "got", ft[lane_, i, d_], "lane", src_lane, "stored", v12[src_lane, c_, 1])
print("flush-layout reads of the lower band that differ from what the lanes stored:", bad)
