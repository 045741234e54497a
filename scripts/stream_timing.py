"""Development aid: end-to-end (host arrays in and out) timing of the streamed batch path against the frame loop and
against what PCIe gives (rpsf_pcie_probe).  python scripts/stream_timing.py [--size 2048] [--patch 128] [--frames 8,32]"""
import argparse
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from oracle import regpsf_oracle as orc  # noqa: E402  (synthetic inputs only)
from regularizepsf_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--patch", type=int, default=128)
ap.add_argument("--frames", default="8,32")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-bind", action="store_true", help="do not bind the process to the GPU's NUMA node")
a = ap.parse_args()
if not a.no_bind:
    print("bound to NUMA node", _native.bind_to_device_node(0))
h = w = a.size
n = a.patch
coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((h, w), n)]
rng = np.random.default_rng(5)
k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64) * 0.1
plan = _native.Plan(n, coords)
plan.set_transfer(k)
pad = _native.PAD_MODES["symmetric"]
print(f"host pool: {_native.host_threads()} threads; frame {h}x{w} f32 = {h*w*4/1e6:.1f} MB")
for nbytes in (h * w * 4, h * w * 8):
    pr = _native.pcie_probe(nbytes, 5)
    print(f"pcie {nbytes/1e6:.1f} MB: h2d {pr['h2d_ms']:.3f} ms ({nbytes/pr['h2d_ms']/1e6:.1f} GB/s), d2h {pr['d2h_ms']:.3f} ms "
          f"({nbytes/pr['d2h_ms']/1e6:.1f} GB/s), both at once {pr['duplex_ms']:.3f} ms")
floor = _native.pcie_probe(h * w * 4, 5)["duplex_ms"]
for frames in [int(v) for v in a.frames.split(",")]:
    base = (rng.standard_normal((frames, h, w)) * 5 + 100).astype(np.float32)
    for in_dt, out_dt in ((np.float32, np.float32), (np.float32, np.float64), (np.float64, np.float64)):
        imgs = base.astype(in_dt)
        out = np.zeros((frames, h, w), out_dt)  # warm pages
        loop = np.zeros((frames, h, w), out_dt)
        best_loop = best_stream = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter()
            for f in range(frames):
                plan.apply_host(imgs[f], pad, out=loop[f])
            best_loop = min(best_loop, time.perf_counter() - t0)
            t0 = time.perf_counter()
            plan.apply_frames_host(imgs, pad, out=out)
            best_stream = min(best_stream, time.perf_counter() - t0)
        same = np.array_equal(out, loop)
        print(f"{frames:3d} frames {np.dtype(in_dt).name}->{np.dtype(out_dt).name}: loop {1e3*best_loop/frames:.3f} ms/frame, "
              f"streamed {1e3*best_stream/frames:.3f} ms/frame = {1e3*best_stream/frames/floor:.2f} x pcie floor ({floor:.3f} ms), "
              f"{frames*h*w/best_stream/1e6:.0f} Mpx/s, identical: {same}")
