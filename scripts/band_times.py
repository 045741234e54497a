"""Strong-scaling projection of BASELINE config 4 (one 8192^2 frame, 256-px patches) from ONE GPU: every row band of a world of
1 / 2 / 4 / 8 ranks runs the product's own sharded step - ShardedApply.step(): the seam plan (the band's last lattice row, on
its stream, computed ONCE: its upper rows come back to the band through K4) beside the main plan (every other patch of the band, 8 CUs
left free, as with a communicator attached), the 128 spill rows moved on the seam stream, the receiver's stream waiting for them, K4
adding them - one band at a time, back to back, after a 100 ms prewarm.
The link is replaced by a device-to-device hipMemcpyAsync of the same 4 MiB on the seam stream (`LocalLink`; what arrives is not
what a neighbour would send - this script times, it does not verify).  Time per step = wall clock over the steps between two device
synchronisations.  speedup = (world 1, same process, same clocks) / (slowest band).

    python scripts/band_times.py [--steps 60] [--no-overlap] [--seam recompute]
"""
import argparse
import ctypes
import json
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering  # noqa: E402
from regularizepsf_amd.sharding import ShardedApply  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--no-overlap", action="store_true")
ap.add_argument("--pipeline", action="store_true", help="exchange mode: the send / recv / add of step k beside the launch of step k + 1")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--seam", choices=["exchange", "recompute"], default="exchange")
ap.add_argument("--ranks", default="", help="development sweeps: only these ranks of every world (comma-separated; default all)")
ap.add_argument("--link-us", type=float, default=0.0,
                help="model the link: the stand-in's send + receive of the seam rows together take about this long (narrow kernels on as few workgroups as it "
                     "takes; 27 = 4 MiB over one xGMI link at 153 GB/s).  0: a plain device-to-device copy (3 us)")
args = ap.parse_args()

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipMemcpyAsync.restype = ctypes.c_int


class LocalLink:
    """Timing stand-in for _native.Comm: the seam rows are copied device-to-device on the stream the exchange would use."""

    def __init__(self, floats):
        self.scratch = _native.DeviceBuffer(max(1, floats) * 4)
        self._plan = _native.Plan(16, [(0, 0)])  # owns a stream for bands that have no seam plan
        self.stream = self._plan.stream

    narrow = False  # --pipeline: the stand-in moves the rows with narrow grid-stride kernels, as RCCL's few send / recv channels do
    workgroups = 64  # of those kernels (--link-us: calibrated below so that send + receive take the link's time)

    def calibrate(self, floats, target_us):
        """Fewest-workgroup setting whose send + receive of `floats` take at least `target_us` (measured here, on an idle GPU)."""
        src = _native.DeviceBuffer(floats * 4)
        best = None
        for g in (64, 48, 32, 24, 16, 12, 8, 6, 4, 3, 2, 1):
            self.workgroups = g
            for _ in range(3):
                self.seam_exchange(src.ptr, floats, src.ptr, floats)
            self._plan.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                self.seam_exchange(src.ptr, floats, src.ptr, floats)
            self._plan.synchronize()
            us = 1e6 * (time.perf_counter() - t0) / 20
            best = (g, us)
            if us >= target_us:
                break
        self.workgroups = best[0]
        return best

    def seam_exchange(self, send_ptr, send_count, recv_ptr, recv_count, stream=None):
        st = stream if stream is not None else self.stream
        if self.narrow:  # (adds instead of copies: the same bytes through the same few CUs)
            if send_count:
                _native.add_rows(self.scratch.ptr, send_ptr, send_count, 0, st, max_workgroups=self.workgroups)
            if recv_count:
                _native.add_rows(recv_ptr, self.scratch.ptr, recv_count, 0, st, max_workgroups=self.workgroups)
            return
        if send_count:
            assert hip.hipMemcpyAsync(self.scratch.ptr, send_ptr, send_count * 4, 3, st) == 0
        if recv_count:
            assert hip.hipMemcpyAsync(recv_ptr, self.scratch.ptr, recv_count * 4, 3, st) == 0

    def seam_exchange_add(self, send_ptr, send_count, recv_ptr, recv_count, accum_ptr, stream=None):
        self.seam_exchange(send_ptr, send_count, recv_ptr, recv_count, stream)
        if recv_count:
            _native.add_rows(accum_ptr, recv_ptr, recv_count, 0, stream)


rng = np.random.default_rng(0)
n, h, w = 256, 8192, 8192
coords = [tuple(int(v) for v in t) for t in calculate_covering((h, w), n)]
kk = (rng.standard_normal((585, n, n), dtype=np.float32) + 1j * rng.standard_normal((585, n, n), dtype=np.float32)).astype(np.complex64)
image = rng.standard_normal((h, w), dtype=np.float32)
link = LocalLink(128 * w)
link.narrow = args.pipeline or args.link_us > 0
if args.link_us > 0:
    g, us = link.calibrate(128 * w, args.link_us)
    print(json.dumps({"link_model": f"send + receive of {128 * w * 4 >> 20} MiB by narrow kernels on {g} workgroups: {us:.1f} us per exchange on an idle GPU (asked: {args.link_us})"}), flush=True)
base = None
for world in [int(v) for v in args.worlds.split(",")]:
    times = []
    for rank in ([r for r in (int(v) for v in args.ranks.split(",")) if r < world] if args.ranks else range(world)):
        sh = ShardedApply(coords, lambda idx: np.resize(kk, (len(idx), n, n)), n, h, w, rank, world, 0, link if world > 1 else None,
                          seam=args.seam, overlap="pipeline" if args.pipeline else not args.no_overlap)
        b = sh.band
        sh.upload_rows(image[b.image_row0:b.image_row0 + b.image_rows])
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:  # prewarm: steady clocks
            for _ in range(8):
                sh.step()
            sh.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sh.step()
        sh.synchronize()
        us = (time.perf_counter() - t0) / args.steps * 1e6
        times.append((len(b.patch_index), sh.seam_plan.n_patches if sh.seam_plan is not None else 0, round(us, 1)))
        if sh.seam_plan is not None:
            sh.seam_plan.close()
            for buf in ([sh.d_spill] if sh.d_spill is not None else sh.d_seam):
                buf.free()
        sh.plan.close(); sh.d_img.free(); sh.d_recv.free()
        for buf in (sh.d_outs if sh.pipeline else [sh.d_out]):
            buf.free()
    slowest = max(t for _, _, t in times)
    base = base or slowest
    print(json.dumps({"world": world, "seam": args.seam, "overlap": ("pipeline" if args.pipeline else not args.no_overlap) if args.seam == "exchange" else False, "bands (patches, seam patches, us per step)": times,
                      "slowest_us": slowest, "speedup_vs_world_1": round(base / slowest, 2)}), flush=True)
