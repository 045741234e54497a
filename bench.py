"""Benchmark of the hot path: ArrayPSFTransform.apply on a device-resident synthetic starfield.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--no-cpu]

One "step" = one full device-resident apply (fused patch kernel + overlap-add) with image, output and packed transfer
kernel resident in HBM.
  N = 1: the headline workload, BASELINE.json configs[2]: 4096x4096 image, 256x256 patches, 1089-patch lattice, coma PSF
         grid -> Gaussian target, alpha=3, eps=0.1.
  N > 1 (launched by torch.distributed.run, one process per GPU): BASELINE.json configs[3], ONE 8192x8192 frame cut into
         N row bands of the patch lattice (strong scaling).  `--seam recompute` (default): both neighbours compute the lattice row
         on their seam, no data-path collective; `--seam exchange`: the rows a band's last lattice row spills into the next band
         are sent to that rank with RCCL send/recv and added there, inside the timed region.
         `--weak` keeps the per-GPU work fixed instead (a (4096 N) x 4096 image of config 3's recipe).
torch is used here only for the launcher's rendezvous: a gloo group broadcasts the 128-byte RCCL unique id; the compute
path, the barrier/max-reduction around the timed region and the seam exchange are ctypes -> librpsf_hip.so (RCCL is loaded
by the library itself).

Before the W warm-up steps every rank keeps its GPU busy for `--prewarm-ms` (default 100 ms, untimed, local applies
only): a freshly woken MI355X spends its first millisecond of work below steady clocks.  The timed region is unchanged:
barrier + synchronise, exactly K steps, barrier + synchronise, maximum over ranks.

Prints ONE JSON line on rank 0 (see the "Measurement" section of DESIGN.md for every field).
"""

from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
MEASURED_COPY_GBS = 6290.0  # the same guide's measured device-copy ceiling (SURVEY.md 8d asks for both)

CONFIGS = {  # name -> (height, width, patch, starfield seed)
    1: (512, 512, 32, 1),      # BASELINE.json configs[0], the plumbing configuration: constant Gaussian PSF 1.8 -> 1.5 (its point is the cpu_baseline leg)
    2: (2048, 2048, 128, 2),
    3: (4096, 4096, 256, 3),
    4: (8192, 8192, 256, 4),   # BASELINE.json configs[3]: one 8192^2 frame, row bands over the ranks (strong scaling)
    5: (2048, 2048, 128, 100),  # batch of frames sharing config 2's transfer kernel (BASELINE.json configs[4])
    6: (4096, 4096, 64, 6),    # not a BASELINE config: the patch size of the reference's own example (docs/source/example.ipynb: psf_size = 64) at
                               # the frame size of the headline; since round 6 the third-generation (sweep) kernel
    7: (4096, 4096, 32, 7),    # the same for 32-pixel patches (BASELINE config 1's patch size at the headline's frame size)
    8: (4096, 4096, 16, 8),    # and for 16-pixel patches
}


_REAL_STDOUT = None


def quiet_stdout() -> None:
    """stdout carries exactly one JSON line.  Libraries print there as well (gloo's connection messages, RCCL's
    banner with NCCL_DEBUG=VERSION - flushed at exit, i.e. after our line), so file descriptor 1 is pointed at
    stderr for the duration of the run and the JSON line is written to the saved original descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_json(line: dict) -> None:
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or "unknown"


def cpu_baseline(image, coords, k, budget_s: float = 25.0):
    """Time the CPU oracle (bit-identical restatement of the reference) on the GPU box's host cores.

    Two legs, each best of 3 after one warm-up run, on the same inputs as the GPU step:
      * `workers = os.cpu_count()`: scipy.fft on every core; the sample is the whole frame if a probe says it fits the
        budget, otherwise a top band of lattice rows (the first lattice rows of patches and the image rows they cover);
      * `workers = None`, the reference's default (single-threaded FFTs), same rule.
    """
    import numpy
    import scipy

    from oracle import regpsf_oracle as orc

    n = k.shape[1]
    cores = os.cpu_count() or 1
    rows = sorted({r for r, _ in coords})

    def band(n_rows):
        last_row = rows[n_rows - 1]
        sel = [i for i, (r, _) in enumerate(coords) if r <= last_row]
        h = min(image.shape[0], last_row + n)
        done_rows = h if n_rows == len(rows) else last_row + n // 2  # rows fully covered by the sampled lattice rows
        return sel, h, done_rows

    def leg(workers):
        sel, h, _ = band(2)
        t0 = time.perf_counter()
        orc.apply_transfer(image[:h], [coords[i] for i in sel], k[sel], workers=workers)
        per_row = (time.perf_counter() - t0) / 2
        n_rows = int(max(2, min(len(rows), (budget_s / 4) / max(per_row, 1e-3))))  # 1 warm-up + 3 timed runs in the budget
        sel, h, done_rows = band(n_rows)
        times = []
        for _ in range(4):
            t0 = time.perf_counter()
            orc.apply_transfer(image[:h], [coords[i] for i in sel], k[sel], workers=workers)
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        return done_rows * image.shape[1] / 1e6 / best, best, done_rows, len(sel)

    v_all, t_all, rows_all, n_all = leg(cores)
    v_one, t_one, rows_one, n_one = leg(None)
    return {
        "value": round(v_all, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "sample": f"top {rows_all} of {image.shape[0]} image rows ({n_all} of {len(coords)} patches), float64 NumPy/SciPy oracle, "
                  f"scipy.fft workers={cores}, best of 3 after 1 warm-up, {t_all:.2f} s",
        "single_thread_value": round(v_one, 3),
        "single_thread_sample": f"workers=None (the reference's default) on the top {rows_one} rows ({n_one} patches), "
                                f"best of 3 after 1 warm-up, {t_one:.2f} s",
        "cpu_model": cpu_model(), "numpy": numpy.__version__, "scipy": scipy.__version__,
    }


def e2e_host_frames(plan, images, pad_mode, out_dtype=np.float64, reps=5, pinned=False):
    """End to end, host arrays in and out ("report end-to-end (H2D + kernel + D2H) separately"): the path
    ArrayPSFTransform.apply / apply_batch take.  `images`: (frames, H, W) host stack; the result stack is allocated (and its pages
    touched) once, outside the clock, as a caller that reuses its buffers would.  Returns best-of-`reps` milliseconds for the
    frame-by-frame loop (rpsf_apply_host per frame) and for the streamed call (rpsf_apply_frames_host), and the PCIe floor:
    the time PCIe needs for one frame's float32 bytes in each direction at once (rpsf_pcie_probe, pinned memory, two streams)."""
    from regularizepsf_amd import _native

    frames, h, w = images.shape
    if pinned:  # the caller keeps frames and results in page-locked arrays (regularizepsf_amd.pinned_empty): no staging copy where no conversion is due
        src = _native.pinned_empty(images.shape, images.dtype, plan.device)
        src[...] = images
        images = src
        out = _native.pinned_empty((frames, h, w), out_dtype, plan.device)
        out[...] = 0
    else:
        out = np.empty((frames, h, w), out_dtype)
        out.fill(0)  # (np.zeros maps its pages lazily: the first apply would take the page faults inside the clock)
    loop_ms = stream_ms = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        for f in range(frames):
            plan.apply_host(images[f], pad_mode, out=out[f])
        loop_ms = min(loop_ms, 1e3 * (time.perf_counter() - t0))
        if frames > 1:
            t0 = time.perf_counter()
            plan.apply_frames_host(images, pad_mode, out=out)
            stream_ms = min(stream_ms, 1e3 * (time.perf_counter() - t0))
    probe = _native.pcie_probe(h * w * 4, 5, plan.device)
    return loop_ms / frames, (stream_ms / frames if frames > 1 else None), probe


class GlooSeam:
    """Debug stand-in for regularizepsf_amd._native.Comm: same calls, seam rows travel through the host over gloo."""

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = rank, world, device

    def seam_exchange_add(self, send_ptr, send_count, recv_ptr, recv_count, accum_ptr, stream=None):
        import ctypes
        import torch
        import torch.distributed as dist
        from regularizepsf_amd import _native

        lib = _native.lib()
        _native.check(lib.rpsf_device_synchronize(self.device))
        req = None
        if self.rank + 1 < self.world and send_count:
            out = np.empty(send_count, np.float32)
            _native.check(lib.rpsf_memcpy_d2h(self.device, out.ctypes.data_as(ctypes.c_void_p), send_ptr, out.nbytes))
            req = dist.isend(torch.from_numpy(out), self.rank + 1)
        if self.rank > 0 and recv_count:
            got = torch.empty(recv_count, dtype=torch.float32)
            dist.recv(got, self.rank - 1)
            acc = np.empty(recv_count, np.float32)
            _native.check(lib.rpsf_memcpy_d2h(self.device, acc.ctypes.data_as(ctypes.c_void_p), accum_ptr, acc.nbytes))
            acc += got.numpy()
            _native.check(lib.rpsf_memcpy_h2d(self.device, accum_ptr, acc.ctypes.data_as(ctypes.c_void_p), acc.nbytes))
        if req is not None:
            req.wait()

    def seam_exchange(self, send_ptr, send_count, recv_ptr, recv_count, stream=None):
        """The transfer alone (synchronous here): spill rows to rank + 1, rows from rank - 1 into recv_ptr."""
        import ctypes
        import torch
        import torch.distributed as dist
        from regularizepsf_amd import _native

        lib = _native.lib()
        _native.check(lib.rpsf_device_synchronize(self.device))
        req = None
        if self.rank + 1 < self.world and send_count:
            out = np.empty(send_count, np.float32)
            _native.check(lib.rpsf_memcpy_d2h(self.device, out.ctypes.data_as(ctypes.c_void_p), send_ptr, out.nbytes))
            req = dist.isend(torch.from_numpy(out), self.rank + 1)
        if self.rank > 0 and recv_count:
            got = torch.empty(recv_count, dtype=torch.float32)
            dist.recv(got, self.rank - 1)
            _native.check(lib.rpsf_memcpy_h2d(self.device, recv_ptr, got.numpy().ctypes.data_as(ctypes.c_void_p), got.numpy().nbytes))
        if req is not None:
            req.wait()

    stream = None

    def barrier(self, stream=None):
        import torch.distributed as dist

        dist.barrier()

    def allreduce_max(self, value):
        import torch
        import torch.distributed as dist

        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def close(self):
        pass


def prewarm(run_step, synchronize, budget_ms: float) -> None:
    """Untimed steps until the device has been busy for `budget_ms`: a freshly woken GPU runs its first ~millisecond
    of work below its steady clocks (on MI355X the first 50 steps after plan creation take 0.9 ms longer than the next
    50), and the W warm-up steps of the contract cover only 1-2 ms.  Happens before the W warm-up steps."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < budget_ms:
        for _ in range(8):
            run_step()
        synchronize()


def run_batch(args, rank, world, device, comm):
    """Config 5: every rank corrects its own `--frames` frames of 2048^2 with the shared 128-px transfer kernel
    (replicas: no data-path collective; weak scaling - 8 frames per GPU is BASELINE's 64 frames on 8 GPUs)."""
    from oracle import regpsf_oracle as orc
    from regularizepsf_amd import _native

    h, w, n, seed = CONFIGS[5]
    frames = args.frames
    # one process per GPU, next to it: host arrays allocated from here on live on the GPU's NUMA node (the streamed leg copies them)
    bound_node = _native.bind_to_device_node(device) if args.streamed and not args.no_bind else -1
    coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((h, w), n)]
    src = np.stack([orc.coma_psf(n, r, c, h, w) for r, c in coords])
    tgt = orc.psf_fft(orc.gaussian_psf(n, 1.8))[None]
    s_fft = orc.psf_fft(src, workers=-1)
    k = orc.construct_transfer(s_fft, np.broadcast_to(tgt, s_fft.shape), 3.0, 0.1).astype(np.complex64)
    if not np.isfinite(k).all():
        raise ValueError("synthetic transfer kernel is not finite")
    images = np.stack([orc.starfield(h, w, seed + rank * frames + i) for i in range(frames)])
    plan = _native.Plan(n, coords, device)
    plan.set_transfer(k)
    d_in = _native.DeviceBuffer(images.nbytes, device).upload(images)
    d_out = _native.DeviceBuffer(images.nbytes, device)
    geom = _native.Geometry.whole(h, w, _native.PAD_MODES["symmetric"])
    stride = h * w

    def run_step():
        plan.apply_batch_device(d_in.ptr, d_out.ptr, frames, stride, stride, geom)

    def barrier():
        plan.synchronize()  # (device-wide)
        if comm is not None:
            comm.barrier()
            plan.synchronize()

    prewarm(run_step, plan.synchronize, args.prewarm_ms)
    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)
    ms_per_step = 1e3 * elapsed / args.steps
    timed_region_ms = 1e3 * elapsed
    iters = max(10, min(args.steps, 100))
    total_ms, kernel_ms = plan.apply_batch_device_timed(d_in.ptr, d_out.ptr, frames, stride, stride, geom, iters)
    kern_avg_ms = float(np.mean(kernel_ms))
    alg_bytes = plan.transfer_bytes + frames * 2 * h * w * 4  # K once per batch + every frame read and written once
    # the whole device-resident batch step (patch kernel + plane sum, fused or not), as for the single frames; the patch kernel
    # alone is kept as a second figure (with the fused sum they coincide)
    step_ms = ms_per_step if world == 1 else float(np.mean(total_ms))
    achieved = alg_bytes / (step_ms * 1e-3) / 1e9
    achieved_kernel = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
    parity = None
    if args.verify:
        out = d_out.download((frames, h, w)).astype(np.float64)
        worst = [0.0, 0.0]
        for f in (0, frames - 1):
            ref = orc.apply_transfer(images[f], coords, k, workers=-1)
            err = float(np.abs(out[f] - ref).max() / np.abs(ref).max())
            err2 = float(np.linalg.norm(out[f] - ref) / np.linalg.norm(ref))
            print(f"[verify] rank {rank} frame {f}: max|d|/max|ref| = {err:.3e}, rel L2 = {err2:.3e}", file=sys.stderr, flush=True)
            if max(err, err2) > 1e-5:
                raise SystemExit(f"verification failed on rank {rank}")
            worst = [max(worst[0], err), max(worst[1], err2)]
        parity = {"max_rel": float(f"{worst[0]:.3e}"), "l2_rel": float(f"{worst[1]:.3e}"), "bound": 1e-5,
                  "against": "float64 NumPy/SciPy oracle (bit-identical restatement of the reference), first and last frame of the batch"}
    if comm is not None:
        import torch.distributed as dist

        barrier()
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    value = world * frames * h * w / (ms_per_step * 1e-3) / 1e6
    cus, name = _native.device_info(device)
    line = {
        "metric": "corrected Mpixels/sec + fraction of HBM roofline, batch of 2048^2 frames / 128-patch, shared transfer array",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms,
        "ms_per_step": round(ms_per_step, 4), "timed_region_ms": round(timed_region_ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{frames} frames per GPU of {h}x{w} starfields, {n}x{n} patches ({len(coords)} per frame), one "
                        f"transfer kernel shared by the batch (coma PSF grid -> Gaussian target, alpha=3 eps=0.1), "
                        f"{'one GPU' if world == 1 else f'{world} replicas, no data-path collective'}",
            "image": [h, w], "patch": n, "patches": len(coords), "frames_per_gpu": frames, "device": name,
            "compute_units": cus, "resident": "frames, outputs and packed transfer kernel in HBM before the timed region",
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "patch_kernel2_256p" if n == 256 else "patch_kernel2_128p" if n == 128 else "patch_kernel",
            "whole_step_ms": round(step_ms, 4), "frac_patch_kernel_only": round(achieved_kernel / HBM_PEAK_GBS, 4),
            "kernel_avg_ms": round(kern_avg_ms, 4), "kernel_launches": iters, "algorithmic_bytes": int(alg_bytes),
            "bytes_model": "packed folded K read once per batch + every frame read once + every output written once",
            "apply_avg_ms_events": round(float(np.mean(total_ms)), 4),
        },
    }
    if parity:
        line["parity"] = parity
    if args.streamed and world == 1:
        # The same frames as host arrays, in and out, through the streamed entry point (three streams, persistent host pool), against
        # the frame loop and against what PCIe gives a frame on this box; float32 frames in, float64 out is what the reference's
        # users get (FITS frames are float32; transform.py:174-177 returns float64), float32 out skips the widening.
        e2e = {}
        for label, dt in (("f32_to_f64", np.float64), ("f32_to_f32", np.float32)):
            loop_ms, stream_ms, probe = e2e_host_frames(plan, images, _native.PAD_MODES["symmetric"], dt)
            e2e[label] = {"e2e_ms_per_frame": round(stream_ms, 4), "frame_loop_ms_per_frame": round(loop_ms, 4),
                          "over_pcie_floor": round(stream_ms / probe["duplex_ms"], 3),
                          "mpixels_per_s": round(h * w / stream_ms / 1e3, 1)}
        loop_ms, stream_ms, _ = e2e_host_frames(plan, images, _native.PAD_MODES["symmetric"], np.float32, pinned=True)
        e2e["f32_to_f32_pinned_arrays"] = {"e2e_ms_per_frame": round(stream_ms, 4), "frame_loop_ms_per_frame": round(loop_ms, 4),
                                           "over_pcie_floor": round(stream_ms / probe["duplex_ms"], 3), "mpixels_per_s": round(h * w / stream_ms / 1e3, 1)}
        long_images = np.stack([images[i % frames] for i in range(4 * frames)])  # the same path in steady state (start-up and drain amortised)
        _, long_ms, _ = e2e_host_frames(plan, long_images, _native.PAD_MODES["symmetric"], np.float64, reps=3)
        line["e2e_ms_per_frame"] = e2e["f32_to_f64"]["e2e_ms_per_frame"]
        line["pcie_floor_ms"] = round(probe["duplex_ms"], 4)
        line["e2e"] = {
            "what": f"{frames} host frames float32 in, host frames out, rpsf_apply_frames_host (H2D || shared-K launch || D2H + widening on 3 streams, "
                    f"persistent pool of {_native.host_threads()} host threads), best of 5; the floor = one frame's float32 bytes each way at once over PCIe "
                    "(rpsf_pcie_probe: pinned memory, two streams)",
            **e2e, "steady_state_ms_per_frame_f32_to_f64": round(long_ms, 4), "steady_state_frames": 4 * frames,
            "steady_state_over_pcie_floor": round(long_ms / probe["duplex_ms"], 3),
            "pcie": {k: round(v, 4) for k, v in probe.items()}, "frame_bytes": h * w * 4,
            "process_bound_to_numa_node": bound_node, "device_numa_node": _native.device_numa_node(device),
        }
    if not args.no_cpu and world == 1:
        line["cpu_baseline"] = cpu_baseline(images[0], coords, k)
    emit_json(line)


def run_construct(args, device):
    """`--config construct`: the transfer-array build at the headline sizes (1089 patches of 256 x 256; ArrayPSF(device=) x 2 ->
    ArrayPSFTransform.construct, regularizepsf/psf.py:216-219 and transform.py:78-82), every stage timed by itself (best of `--steps`
    runs, wall clock between device synchronisations; every stage is hundreds of microseconds or more) and priced on its algorithmic
    bytes.  The rocprofv3 kernel rows of the same command are under profiles/."""
    from oracle import regpsf_oracle as orc
    from regularizepsf_amd import _native

    h, w, n, _ = CONFIGS[3]
    coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((h, w), n)]
    count, per = len(coords), n * n
    rng = np.random.default_rng(3)
    src = np.stack([orc.coma_psf(n, r, c, h, w) for r, c in coords]).astype(np.float32)
    tgt = np.broadcast_to(orc.gaussian_psf(n, 1.8).astype(np.float32), src.shape).copy()
    lib = _native.lib()
    reps = max(3, min(args.steps, 10))

    def best(fn):
        times = []
        for _ in range(reps):
            _native.check(lib.rpsf_device_synchronize(device))
            t0 = time.perf_counter()
            fn()
            _native.check(lib.rpsf_device_synchronize(device))
            times.append(1e3 * (time.perf_counter() - t0))
        return min(times)

    s_dev = _native.psf_fft_device(src, device)
    t_dev = _native.psf_fft_device(tgt, device)
    k_dev = _native.DeviceBuffer(count * per * 8, device)
    plan = _native.Plan(n, coords, device)
    full_b, packed_b = count * per * 8, None
    ms_fft_host = best(lambda: _native.psf_fft_device(src, device).free())
    ms_k2 = best(lambda: _native.check(lib.rpsf_build_transfer_device(device, count * per, s_dev.ptr, t_dev.ptr, 0, 3.0, 0.1, k_dev.ptr, None)))
    ms_pack = best(lambda: plan.set_transfer_device(k_dev.ptr))
    packed_b = plan.transfer_bytes
    ms_fused = best(lambda: plan.set_transfer_spectra_device(s_dev.ptr, t_dev.ptr, 3.0, 0.1))
    params = np.zeros((count, _native.MODEL_PARAMS))
    params[:, 0], params[:, 1], params[:, 2], params[:, 3], params[:, 4] = 1.0, n / 2, n / 2, 1.5, 1.9
    ms_model = best(lambda: [b.free() for b in _native.psf_model_fft_device("elliptical_gaussian", n, params, True, device) if b is not None])
    # parity of K2: the GPU's K against the oracle's construct on the SAME complex64 spectra (NumPy evaluates complex64 input in
    # float32, as the reference does for float32 PSFs), first 64 patches: identical non-finite bins (float32 spectra of smooth PSFs are exactly 0 in the
    # (N/2, N/2) bin, 0 / 0 = NaN there on both sides), max|K_gpu - K_ref| <= 1e-5 max|K_ref| on the rest
    _native.check(lib.rpsf_build_transfer_device(device, count * per, s_dev.ptr, t_dev.ptr, 0, 3.0, 0.1, k_dev.ptr, None))
    k_gpu = k_dev.download((count, n, n), np.complex64)[:64]
    s64 = s_dev.download((count, n, n), np.complex64)[:64]
    t64 = t_dev.download((count, n, n), np.complex64)[:64]
    with np.errstate(all="ignore"):
        k_ref = orc.construct_transfer(s64, t64, 3.0, 0.1)
    finite = np.isfinite(k_ref)
    same_pattern = bool(np.array_equal(np.isfinite(k_gpu), finite))
    k_err = float(np.abs(k_gpu[finite] - k_ref[finite]).max() / np.abs(k_ref[finite]).max())
    def gbs(nbytes, ms):
        return round(nbytes / (ms * 1e-3) / 1e9, 1)

    fused_bytes = 2 * full_b + packed_b
    cus, name = _native.device_info(device)
    line = {
        "metric": "transfer-array build (ArrayPSFTransform.construct on device-resident spectra): milliseconds + fraction of HBM roofline, 1089 patches of 256x256",
        "value": round(ms_fused, 4), "unit": "ms", "n_gpus": 1, "steps": reps, "warmup": 0, "ms_per_step": round(ms_fused, 4),
        "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{count} patches of {n}x{n}: coma PSF grid -> Gaussian target, alpha=3 eps=0.1; spectra resident in HBM (complex64), packed folded K out",
                   "patch": n, "patches": count, "device": name, "compute_units": cus},
        "roofline": {"bound": "hbm", "achieved": gbs(fused_bytes, ms_fused), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs(fused_bytes, ms_fused) / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "pack_spectra_kernel2",
                     "algorithmic_bytes": int(fused_bytes),
                     "bytes_model": "source and target spectra read once (2 x n N^2 x 8 B) + packed folded K written once (n N (N/2+1) x 8 B)"},
        "stages": {
            "construct_one_pass_ms": round(ms_fused, 4),
            "construct_two_pass_ms": round(ms_k2 + ms_pack, 4),
            "k2_build_transfer": {"ms": round(ms_k2, 4), "bytes": 3 * full_b, "gb_per_s": gbs(3 * full_b, ms_k2), "frac": round(gbs(3 * full_b, ms_k2) / HBM_PEAK_GBS, 4)},
            "pack_kernel2": {"ms": round(ms_pack, 4), "bytes": full_b + packed_b, "gb_per_s": gbs(full_b + packed_b, ms_pack),
                             "frac": round(gbs(full_b + packed_b, ms_pack) / HBM_PEAK_GBS, 4)},
            "k3_psf_fft_from_host_samples": {"ms": round(ms_fft_host, 4), "what": "ArrayPSF(device=): float32 samples over PCIe (pageable) + spectrum kernel, spectra stay on the device",
                                             "h2d_bytes": int(src.nbytes), "spectra_bytes": full_b},
            "k6_k3_model_to_spectra": {"ms": round(ms_model, 4), "what": "rasterise an elliptical Gaussian per patch on the device (float64 evaluation) + spectrum kernel; parameters only cross PCIe",
                                       "bytes": count * per * 4 + full_b},
        },
        "parity": {"k_max_rel": float(f"{k_err:.3e}"), "bound": 1e-5, "non_finite_bins": int((~finite).sum()), "non_finite_pattern_identical": same_pattern,
                   "against": "the oracle's construct (NumPy, float32 arithmetic for complex64 input like the reference) on the same complex64 spectra, first 64 patches "
                              "(SURVEY 8d: max|K_gpu - K_ref| <= 1e-5 max|K_ref| over the finite bins)"},
    }
    emit_json(line)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=[str(c) for c in sorted(CONFIGS)] + ["construct"],
                    help="default: 3 (BASELINE headline) on one GPU, 4 (one 8192^2 frame, row bands, strong scaling) on several")
    ap.add_argument("--patch", type=int, default=None,
                    help="N = 1: a 4096^2 frame with patches of this size instead of a numbered config - any size the reference takes (transform.py:151-155); "
                         "sizes other than 16, 32, 64, 128, 256 run the hipFFT fallback")
    ap.add_argument("--weak", action="store_true", help="N > 1: grow the image with the ranks instead of cutting one frame")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="untimed device activity before the W warm-up steps (clock ramp); 0 = off")
    ap.add_argument("--comm", choices=["rccl", "gloo"], default="rccl",
                    help="seam transport for N > 1; 'gloo' is a debug stand-in (host copies) used to exercise the "
                         "multi-rank flow on a box with fewer GPUs than ranks")
    ap.add_argument("--verify", action="store_true", default=None,
                    help="check every rank's owned rows against the CPU oracle on one step outside the timed loops (default: on)")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--second-leg-timeout", type=float, default=300.0,
                    help="N > 1: seconds the second seam leg may take before the headline line is emitted without it")
    ap.add_argument("--one-seam", action="store_true",
                    help="N > 1: time only the --seam mode (default: the other seam mode is timed as a second leg in the same process)")
    ap.add_argument("--frames", type=int, default=8, help="config 5: frames per GPU in one batch")
    ap.add_argument("--streamed", action="store_true",
                    help="config 5, N = 1: also time the frames as HOST arrays in and out through the streamed entry point "
                         "(e2e_ms_per_frame, pcie_floor_ms in the line)")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the process to the GPU's NUMA node for the end-to-end legs")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the end-to-end (host arrays in and out) leg")
    ap.add_argument("--rotate", type=int, default=1,
                    help="N = 1 only: this many DIFFERENT device-resident frames (and output buffers) are corrected in rotation.  "
                         "Default 1: the same frame every step, as in every earlier round - it then stays in the 256 MB Infinity "
                         "Cache between steps, which a stream of new frames would not (DESIGN.md, Measurement)")
    ap.add_argument("--new-frames", type=int, default=8,
                    help="N = 1: after the timed loop, a second timed loop of K steps over this many different resident frames and "
                         "outputs in rotation; reported as roofline.frac_new_frames / ms_per_step_new_frames next to the headline "
                         "(0 or 1: skip)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="N = 1 only: this many plans (own stream, planes and output; the same K) take turns, so that the head of "
                         "one apply overlaps the tail of the previous one - the throughput of a frame pipeline.  Default 1: one "
                         "apply after the other, which is also what roofline.frac is always computed on")
    ap.add_argument("--no-overlap", action="store_true",
                    help="--seam exchange: plain apply -> send/recv -> add on one stream (same as --exchange-mode plain)")
    ap.add_argument("--exchange-mode", choices=["pipeline", "two-plans", "plain"], default="plain",
                    help="how the seam exchange is run: 'plain' (default: the fewest moving parts for the first run on a real multi-GPU node) - "
                         "apply -> ncclSend / ncclRecv -> K4 add in sequence on the plan's stream; 'pipeline' - one launch per step, the send / recv / add "
                         "of step k on a CU-masked exchange stream beside the launch of step k + 1 (projected 5.2x against 4.5x with a real link's "
                         "latency, DESIGN.md 7); 'two-plans' - the band's last lattice row as a plan of its own, its rows sent beside the rest of the "
                         "same step's band")
    ap.add_argument("--seam", choices=["recompute", "exchange"], default="recompute",
                    help="N > 1: 'recompute' (default) - every band also runs the lattice row above it that reaches into its rows and "
                         "the bands are cut so that own + recomputed patches balance (585 per rank at eight bands of the 8192-wide "
                         "frame): no data-path collective, RCCL carries only the barrier and the max-reduction of the timed region; "
                         "'exchange' - the spill rows of a band are sent to the next rank with RCCL send/recv and added there "
                         "(BASELINE's halo reduce).  Measured band by band on one GPU with the product's own step (scripts/band_times.py, "
                         "profiles/r03e_*): 5.8x against 5.1x at eight bands, which is why recompute is the default (DESIGN.md 7)")
    args = ap.parse_args()
    quiet_stdout()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.config == "construct":
        from regularizepsf_amd import _native as _n

        return run_construct(args, local_rank % max(1, _n.device_count()))
    if args.config is not None:
        args.config = int(args.config)
    if args.config is None:
        args.config = 3 if world == 1 or args.weak else 4
    if args.verify is None:
        args.verify = True

    from oracle import regpsf_oracle as orc  # synthetic inputs + cpu_baseline only
    from regularizepsf_amd import _native
    from regularizepsf_amd.sharding import ShardedApply

    if args.patch is not None:
        if world > 1:
            raise SystemExit("--patch is a one-GPU line")
        CONFIGS[9] = (4096, 4096, args.patch, 9)
        args.config = 9
    h1, w, n, seed = CONFIGS[args.config]
    compiled = n in (16, 32, 64, 128, 256)  # a hand-written plan; everything else: gather -> hipFFT -> x K -> hipFFT -> overlap-add
    strong = args.config == 4 or not args.weak  # one frame cut into `world` bands; --weak: the image grows with the ranks
    height = h1 if strong else h1 * world
    device = local_rank % max(1, _native.device_count())  # (one process per GPU; on a box with fewer GPUs than ranks - debugging - ranks share devices)
    pad = "symmetric"

    # ---------------- inputs: synthetic, same recipe on every rank, each rank builds only its band ----------
    coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((height, w), n)]

    k_cache = {}

    def kernel_for(index):
        """Transfer kernels of the given patches; the PSF field of the h1 x w frame repeats down the tall image."""
        key = (len(index), index[0], index[-1])
        if key in k_cache:
            return k_cache[key]
        out = k_cache[key] = np.empty((len(index), n, n), np.complex64)
        tgt = orc.psf_fft(orc.gaussian_psf(n, 1.5 if args.config == 1 else 1.8))[None]
        for first in range(0, len(index), 128):
            part = index[first:first + 128]
            if args.config == 1:  # constant Gaussian 1.8 -> 1.5
                src = np.broadcast_to(orc.gaussian_psf(n, 1.8), (len(part), n, n))
            else:
                src = np.stack([orc.coma_psf(n, coords[i][0] % h1 if world > 1 and not strong else coords[i][0], coords[i][1], h1, w)
                                for i in part])
            s_fft = orc.psf_fft(src, workers=-1)
            with np.errstate(all="ignore"):
                kk = orc.construct_transfer(s_fft, np.broadcast_to(tgt, s_fft.shape), 3.0, 0.1)
            if not np.isfinite(kk).all():
                raise ValueError("synthetic transfer kernel is not finite")
            out[first:first + len(part)] = kk
        return out

    def image_rows(lo, hi):
        """Rows [lo, hi) of the tall image = `world` independent starfields stacked vertically."""
        if strong:
            return orc.starfield(h1, w, seed)[lo:hi]
        parts = []
        for b in range(world):
            a0, a1 = max(lo, b * h1), min(hi, (b + 1) * h1)
            if a0 < a1:
                parts.append(orc.starfield(h1, w, seed + 100 * b)[a0 - b * h1:a1 - b * h1])
        return np.concatenate(parts)

    comm = None
    if world > 1:
        # stdout carries exactly one JSON line: RCCL's own banner / debug output (NCCL_DEBUG=VERSION is set on
        # the GPU boxes) goes to stderr instead
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
            os.environ["NCCL_DEBUG"] = "WARN"  # the VERSION banner is printf'ed to stdout at process exit, after the JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        import torch.distributed as dist  # launcher plumbing only: a gloo group carries the 128-byte RCCL unique id

        dist.init_process_group(backend="gloo")  # env:// rendezvous from torch.distributed.run (works with its agent store)
        if args.comm == "rccl":
            try:
                box = [_native.Comm.unique_id() if rank == 0 else None]
            except _native.NativeError as e:  # librccl missing: every rank must take the same branch
                box = [None]
                print(f"[bench] RCCL unavailable on rank 0 ({e})", file=sys.stderr, flush=True)
            dist.broadcast_object_list(box, src=0)
            err = None
            if box[0] is not None:
                try:
                    comm = _native.Comm(device, rank, world, bytes(box[0]))
                except _native.NativeError as e:
                    err = str(e)
            else:
                err = "no unique id"
            # every rank must end up on the same transport: agree on whether RCCL came up everywhere
            flags = [None] * world
            dist.all_gather_object(flags, err)
            bad = [f"rank {r}: {f}" for r, f in enumerate(flags) if f is not None]
            if bad:
                if comm is not None:
                    comm.close()
                if rank == 0:  # (a box with fewer GPUs than ranks: RCCL refuses two ranks on one device)
                    print(f"[bench] RCCL did not come up ({'; '.join(bad)}): barrier, max-reduction and seam rows over gloo (host copies)",
                          file=sys.stderr, flush=True)
                comm = GlooSeam(rank, world, device)
        else:
            comm = GlooSeam(rank, world, device)
    if args.config == 5:
        return run_batch(args, rank, world, device, comm)
    overlap = False if args.no_overlap else {"pipeline": "pipeline", "two-plans": True, "plain": False}[args.exchange_mode]
    shard = ShardedApply(coords, kernel_for, n, height, w, rank, world, device, comm, pad_mode=pad, seam=args.seam,
                         overlap=overlap)
    # N > 1: BOTH seam modes run in this process - the --seam one is the headline (timed first), the other a second leg with its own
    # plans, K and buffers - so that the first run on a real multi-GPU node exercises the RCCL send / recv of the halo rows AND the
    # collective-free recompute, whichever of them is the default
    other_seam = {"recompute": "exchange", "exchange": "recompute"}[args.seam]
    shard2 = None
    if world > 1 and not args.one_seam:
        shard2 = ShardedApply(coords, kernel_for, n, height, w, rank, world, device, comm, pad_mode=pad, seam=other_seam,
                              overlap=overlap)
    band = shard.band
    band_image = image_rows(band.image_row0, band.image_row0 + band.image_rows)
    shard.upload_rows(band_image)
    plan, geom, d_img, d_out = shard.plan, shard.geometry, shard.d_img, shard.d_out
    run_step = shard.step
    rotation = []  # --rotate > 1: further frames and outputs, corrected in turn by the same plan (world 1 only)
    if world == 1 and args.rotate > 1 and args.in_flight <= 1:
        for i in range(1, args.rotate):
            frame = orc.starfield(h1, w, seed + 1000 * i)[band.image_row0:band.image_row0 + band.image_rows]
            rotation.append((_native.DeviceBuffer(frame.nbytes, device).upload(np.ascontiguousarray(frame, np.float32)),
                             _native.DeviceBuffer(band.out_rows * w * 4, device)))
        pairs = [(d_img, d_out)] + rotation
        spin = [0]

        def run_step():  # noqa: F811
            a, b = pairs[spin[0] % len(pairs)]
            spin[0] += 1
            plan.apply_device(a.ptr, b.ptr, geom)
    pipeline = []  # --in-flight > 1: further plans with their own streams, planes and outputs (world 1 only)
    if world == 1 and args.in_flight > 1:
        for _ in range(args.in_flight - 1):
            extra = _native.Plan(n, [coords[i] for i in band.patch_index], device=device)
            extra.set_transfer(kernel_for(band.patch_index))
            pipeline.append((extra, _native.DeviceBuffer(band.out_rows * w * 4, device)))
        turn = [0]
        plans = [(plan, d_out)] + pipeline

        def run_step():  # noqa: F811
            q, out = plans[turn[0] % len(plans)]
            turn[0] += 1
            q.apply_device(d_img.ptr, out.ptr, geom)

    def barrier():
        plan.synchronize()
        if comm is not None:
            comm.barrier()
            plan.synchronize()

    # (local applies only: the ranks run different numbers of them, so no seam exchange in here)
    prewarm(lambda: plan.apply_device(d_img.ptr, d_out.ptr, geom), plan.synchronize, args.prewarm_ms)
    for _ in range(args.warmup):
        run_step()
    barrier()
    # N = 1, one plan, one frame: the K steps are launched back to back by ONE library call that also brackets them with a HIP event pair on the
    # plan's stream - so the device time of the timed region itself (roofline.kernel_avg_ms) lies inside the wall-clock step by construction
    timed_loop_event_ms = None
    t0 = time.perf_counter()
    if world == 1 and not rotation and not pipeline:
        timed_loop_event_ms = plan.apply_device_loop_ms(d_img.ptr, d_out.ptr, geom, args.steps)
    else:
        for _ in range(args.steps):
            run_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)  # whole-job time = slowest rank
    ms_per_step = 1e3 * elapsed / args.steps
    timed_region_ms = 1e3 * elapsed

    def verify_owned_rows(sh, label):
        """One step of `sh`, then every rank checks the rows it owns against the float64 oracle on the same inputs: the patches that
        touch those rows on the image rows they read (a band's worth of work per rank, not the whole frame)."""
        sh.step()
        barrier()
        b = sh.band
        lo, hi = b.out_row0, b.out_row0 + b.own_rows
        idx = [i for i, (r, _) in enumerate(coords) if r < hi and r + n > lo]
        r_lo = max(0, min(coords[i][0] for i in idx))
        r_hi = min(height, max(coords[i][0] for i in idx) + n)
        # (cropping the image is exact: a patch hangs over the crop only where the crop edge is the image edge)
        sub = image_rows(r_lo, r_hi)
        ref = orc.apply_transfer(sub, [(coords[i][0] - r_lo, coords[i][1]) for i in idx], kernel_for(idx), workers=-1)
        ref_own = ref[lo - r_lo:hi - r_lo]
        own = sh.owned_rows().astype(np.float64)
        scale = float(np.abs(ref).max())
        err = float(np.abs(own - ref_own).max() / scale)
        err2 = float(np.linalg.norm(own - ref_own) / np.linalg.norm(ref_own))
        print(f"[verify] rank {rank} seam={label}: rows {lo}..{hi}, max|d|/max|ref| = {err:.3e}, rel L2 = {err2:.3e}", file=sys.stderr, flush=True)
        worst = comm.allreduce_max(max(err, err2)) if comm is not None else max(err, err2)
        if worst > 1e-5:
            raise SystemExit(f"verification failed (seam={label}): {worst:.3e}")
        parity[label] = {"max_rel": float(f"{err:.3e}"), "l2_rel": float(f"{err2:.3e}")}  # (rank 0's own rows)
        return worst

    verified, parity = {}, {}
    if args.verify:
        verified[args.seam if world > 1 else "single"] = verify_owned_rows(shard, args.seam if world > 1 else "single")

    # ---------------- the same plan on a stream of NEW frames (N = 1): --new-frames different resident starfields and outputs in
    # rotation.  The headline loop above corrects ONE frame K times (SURVEY.md 8d's timed region); that frame and part of its colour
    # planes then sit in the 256 MB Infinity Cache from step to step, which a production stream of frames does not have.
    new_frames_ms = new_frames_prefetch_ms = None
    if world == 1 and not rotation and not pipeline and args.new_frames > 1:
        extra = []
        for i in range(1, args.new_frames):
            frame = orc.starfield(h1, w, seed + 1000 * i)[band.image_row0:band.image_row0 + band.image_rows]
            extra.append((_native.DeviceBuffer(frame.nbytes, device).upload(np.ascontiguousarray(frame, np.float32)),
                          _native.DeviceBuffer(band.out_rows * w * 4, device)))
        ring = [(d_img, d_out)] + extra
        # (building the frames on the host took seconds: the GPU has idled back to low clocks)
        prewarm(lambda: plan.apply_device(d_img.ptr, d_out.ptr, geom), plan.synchronize, args.prewarm_ms)
        for i in range(max(args.warmup, len(ring))):
            plan.apply_device(ring[i % len(ring)][0].ptr, ring[i % len(ring)][1].ptr, geom)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            plan.apply_device(ring[i % len(ring)][0].ptr, ring[i % len(ring)][1].ptr, geom)
        barrier()
        new_frames_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        if n == 256:  # the same stream of frames with the opt-in image prefetch (rpsf_plan_set_image_prefetch), then back to the default
            plan.set_image_prefetch(True)
            for i in range(len(ring)):
                plan.apply_device(ring[i][0].ptr, ring[i][1].ptr, geom)
            barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                plan.apply_device(ring[i % len(ring)][0].ptr, ring[i % len(ring)][1].ptr, geom)
            barrier()
            new_frames_prefetch_ms = 1e3 * (time.perf_counter() - t0) / args.steps
            plan.set_image_prefetch(False)
        del extra, ring

    # ---------------- roofline: HIP events on the plan's stream, live ----------------
    # Algorithmic bytes: folded K read once (n N (N/2+1) complex64) + image read once + output written
    # once, over the whole device-resident apply of this rank (patch kernel + everything the overlap-add needs).  With
    # the plane sum fused into the patch launch (N = 256) the apply IS one launch of the dominant kernel.
    iters = max(20, min(args.steps, 200))
    # (the verification and the new-frames set-up above kept the host busy for seconds: the GPU has idled back to low clocks)
    prewarm(lambda: plan.apply_device(d_img.ptr, d_out.ptr, geom), plan.synchronize, args.prewarm_ms)
    loop_ms = plan.apply_device_loop_ms(d_img.ptr, d_out.ptr, geom, iters)
    total_ms, kernel_ms = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, iters)
    kern_avg_ms, apply_avg_ms = float(np.mean(kernel_ms)), float(np.mean(total_ms))
    # (each per-apply event pair puts a marker packet between two launches, ~8 us of the 184 here: the loop as the device sees it is one pair of
    # events around `iters` back-to-back applies; with the plane sum fused into the patch launch - N = 128, 256 - an apply IS one launch of the kernel)
    sweep = plan.sweep_info() if n <= 64 else {"regions": 0}
    one_launch = compiled and (n >= 128 or sweep["regions"] > 0)  # the whole apply is one launch of the dominant kernel
    if one_launch:
        kern_avg_ms = timed_loop_event_ms if timed_loop_event_ms is not None else loop_ms
    my_patches = plan.n_patches
    alg_bytes = my_patches * n * (n // 2 + 1) * 8 + band.image_rows * w * 4 + band.out_rows * w * 4
    # one rank's apply; at N = 1 the wall-clock step itself (with --in-flight > 1 the steps overlap: the single apply by events)
    step_ms = ms_per_step if world == 1 and not pipeline else apply_avg_ms
    achieved = alg_bytes / (step_ms * 1e-3) / 1e9
    achieved_kernel = alg_bytes / (kern_avg_ms * 1e-3) / 1e9

    rccl_ranks = None
    if comm is not None and not isinstance(comm, GlooSeam):
        try:
            rccl_ranks = comm.ranks()
        except Exception as e:  # noqa: BLE001 - a diagnostic field must not take the run down
            print(f"[bench] ncclCommCount: {e}", file=sys.stderr, flush=True)
    total_pixels = height * w
    value = total_pixels / (ms_per_step * 1e-3) / 1e6
    cus, name = _native.device_info(device)
    line = {
        "metric": f"corrected Mpixels/sec + fraction of HBM roofline, {h1}^2 image / {n}-patch",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms,
        "ms_per_step": round(ms_per_step, 4), "timed_region_ms": round(timed_region_ms, 3), "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{height}x{w} starfield, {n}x{n} patches, {len(coords)} patches "
                        f"({'whole image on one GPU' if world == 1 else f'{world} row bands, seam rows ' + ('recomputed by both neighbours (no data-path collective)' if args.seam == 'recompute' else f'exchanged over {args.comm.upper()}')}), "
                        f"{'constant Gaussian PSF 1.8 -> 1.5' if args.config == 1 else 'coma PSF grid -> Gaussian target'}, alpha=3 eps=0.1, pad symmetric",
            "image": [height, w], "patch": n, "patches": len(coords), "device": name, "compute_units": cus,
            "resident": "image, output and packed transfer kernel in HBM before the timed region",
            "in_flight": 1 + len(pipeline), "frames_in_rotation": 1 + len(rotation),
            "sync": "none (one rank)" if comm is None else ("gloo" if isinstance(comm, GlooSeam) else "rccl"),
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "frac_of_measured_copy_ceiling": round(achieved / MEASURED_COPY_GBS, 4),
            "kernel": ("hipFFT fallback: generic_gather + hipFFT C2C forward + generic_multiply + hipFFT C2C inverse + 4 x generic_scatter (colour classes) per chunk"
                       if not compiled else
                       "patch_kernel2_256p" if n == 256 else "patch_kernel2_128p" if n == 128 else "sweep_kernel" if sweep["regions"] else "patch_kernel"),
            "whole_apply_ms": round(step_ms, 4), "kernel_avg_ms": round(kern_avg_ms, 4), "kernel_launches": iters,
            "frac_patch_kernel_only": round(achieved_kernel / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": int(alg_bytes), "packed_k_bytes": int(plan.transfer_bytes),
            "bytes_model": "SURVEY 8d: folded K (n N (N/2+1) complex64) read once + image read once + output written once "
                           "(rank 0's band), divided by the whole device-resident apply",
            "apply_avg_ms_events": round(apply_avg_ms, 4), "apply_avg_ms_event_loop": round(loop_ms, 4), "patches_this_rank": my_patches,
            "kernel_avg_ms_from": ("one HIP event pair around the K back-to-back launches of the timed region itself (apply = one launch)" if one_launch and timed_loop_event_ms is not None else
                                   "one HIP event pair around back-to-back launches after the timed loop (apply = one launch)" if one_launch else
                                   "HIP event pairs around every patch-kernel launch"),
        },
    }
    if not compiled:  # the path the reference allows for every N: what it costs, with the bytes it really moves beside the algorithmic ones
        per = n * n * 8
        line["roofline"]["fallback"] = {
            "what": "no hand-written plan for this patch size: full complex transforms through hipFFT on a staging buffer, the caller's unfolded K",
            "bytes_moved_model": int(band.image_rows * w * 4 * 4 + my_patches * per * (1 + 2 + 3 + 2 + 1) + band.out_rows * w * 4 * (1 + 2 * 4)),
            "bytes_moved_what": "gather (4 x image read, buffer write) + forward FFT (read + write) + multiply (buffer read + write, K read) + inverse FFT "
                                "(read + write) + scatter (buffer read, 4 x output read-modify-write) + the output memset, if nothing stayed in a cache",
            "overlap_add": "colour classes, fixed order (bit-reproducible)"}
    if sweep["regions"]:  # third generation: how the lattice was cut (patches on region borders are computed by both neighbours)
        line["config"]["sweep"] = {"regions": sweep["regions"], "jobs": sweep["jobs"], "slabs_per_phase": sweep["slabs_per_phase"],
                                   "recompute_factor": round(sweep["patch_slots"] / max(1, my_patches), 3)}
    if n == 256 and world == 1:  # what this kernel STRUCTURE can reach, as measured: not a tuning target but a bound
        line["roofline"]["structure_ceiling"] = {
            "frac": 0.32, "ms": 0.165,
            "floors_ms": {"memory_system_alone": [0.145, 0.158], "on_chip_chain_alone": [0.153, 0.165], "without_lock_step_barriers": 0.167},
            "what": "one 256-pixel patch per CU with four colour planes: the apply's 980 MB on the memory system alone, the on-chip chain alone, "
                    "the product without its exchange barriers; a perfectly overlapped version of this structure lands at ~0.165 ms",
            "profiles": ["profiles/r04a_persistent_kernel_decomposition.log", "profiles/r04an_cost_of_each_exchange_barrier.log",
                         "profiles/r04b_split_patch_skeleton_upper_bound.log", "profiles/r05s_split_skeleton_audit.log"]}
    # ---------------- N > 1: the other seam mode, same steps, same barriers - AFTER the headline line exists, under a watchdog: a hang or an
    # error in this leg (RCCL send / recv between GPUs has never run before the first multi-GPU node this is launched on) must not cost the
    # headline its JSON line ----------------
    other_ms, second_leg_note = None, None
    if shard2 is not None:
        import threading

        def give_up():
            if rank == 0:
                line["config"]["seam"] = args.seam
                line["sync"] = "gloo" if isinstance(comm, GlooSeam) else "rccl"
                line["rccl_ranks"] = rccl_ranks
                line[f"scaling_{args.seam}_ms"] = round(ms_per_step, 4)
                line[f"scaling_{other_seam}_ms"] = None
                line["second_leg"] = f"seam={other_seam}: no result after {args.second_leg_timeout:.0f} s (watchdog); the headline leg is complete"
                line["verify"] = {"max_error": max(verified.values()) if verified else None, "legs": sorted(verified), "bound": 1e-5,
                                  "failed_legs": [other_seam]}
                emit_json(line)
            # kernels or collectives of the second leg may never return: no orderly shutdown - and NOT exit code 0: the headline line is out,
            # but a harness that looks at the return code must see that a leg hung
            os._exit(3)

        dog = threading.Timer(args.second_leg_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            band2 = shard2.band
            shard2.upload_rows(image_rows(band2.image_row0, band2.image_row0 + band2.image_rows))
            for _ in range(max(args.warmup, 2)):
                shard2.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                shard2.step()
            barrier()
            other_ms = 1e3 * comm.allreduce_max(time.perf_counter() - t0) / args.steps
            if args.verify:
                verified[other_seam] = verify_owned_rows(shard2, other_seam)
        except (Exception, SystemExit) as e:  # noqa: BLE001 - reported in the line (a failed verification too); peers that wait for this rank are ended by their watchdogs
            second_leg_note = f"seam={other_seam}: {type(e).__name__}: {e}"
            print(f"[bench] second leg failed on rank {rank}: {second_leg_note}", file=sys.stderr, flush=True)
        dog.cancel()
    if comm is not None and second_leg_note is None:  # orderly shutdown: nobody tears RCCL down while a peer is still in a collective
        import torch.distributed as dist

        barrier()
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        if second_leg_note is not None:  # (no orderly shutdown after a failed leg, and not exit code 0)
            os._exit(3)
        return
    if world > 1:  # both seam modes of this run (the headline is config.seam), the transport, and what RCCL says about its communicator
        line["config"]["seam"] = args.seam
        line["config"]["exchange_mode"] = "plain" if overlap is False else "two-plans" if overlap is True else "pipeline"
        line["sync"] = "gloo" if isinstance(comm, GlooSeam) else "rccl"
        line["rccl_ranks"] = rccl_ranks
        line[f"scaling_{args.seam}_ms"] = round(ms_per_step, 4)
        if second_leg_note is not None:
            line["second_leg"] = second_leg_note
        if other_ms is not None:
            line[f"scaling_{other_seam}_ms"] = round(other_ms, 4)
            line["seam_rows_transport"] = ("gloo (host copies: debugging stand-in)" if isinstance(comm, GlooSeam) else
                                           "RCCL ncclSend / ncclRecv in the timed region")
    if verified or second_leg_note is not None:
        line["verify"] = {"max_error": max(verified.values()) if verified else None, "legs": sorted(verified), "bound": 1e-5,
                          "failed_legs": [other_seam] if second_leg_note is not None else [],
                          "what": "every rank's own output rows of one step against the float64 oracle (max|d|/max|ref| and relative L2)"}
        first = parity.get(args.seam if world > 1 else "single")
        if first:  # SURVEY 8d's parity metric on the very inputs of the timed loop: max|d| / max|ref| and ||d||2 / ||ref||2, bound 1e-5
            line["parity"] = {**first, "bound": 1e-5, "against": "float64 NumPy/SciPy oracle (bit-identical restatement of the reference), same image and complex64 K"}
    if new_frames_ms is not None:  # same bytes, same plan, every step a frame the caches have not seen for args.new_frames - 1 applies
        line["roofline"]["ms_per_step_new_frames"] = round(new_frames_ms, 4)
        line["roofline"]["frac_new_frames"] = round(alg_bytes / (new_frames_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # (the production figure - a stream of frames - beside the repeated-frame one, in the metric string itself)
        line["metric"] += f" [roofline frac {line['roofline']['frac']} on a repeated frame, {line['roofline']['frac_new_frames']} on a stream of new frames]"
        line["value_new_frames"] = round(total_pixels / (new_frames_ms * 1e-3) / 1e6, 1)
        line["roofline"]["new_frames_in_rotation"] = args.new_frames
        if new_frames_prefetch_ms is not None:  # opt-in, not the default: it costs frames that are larger than the cache
            line["roofline"]["ms_per_step_new_frames_with_image_prefetch"] = round(new_frames_prefetch_ms, 4)
    if pipeline:  # the overlapped steps priced on the same bytes (not roofline.frac: SURVEY 8d's t is one device-resident apply)
        line["roofline"]["frac_steps_in_flight"] = round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if world == 1 and not args.no_e2e:
        affinity = os.sched_getaffinity(0)
        try:  # (an auxiliary leg: whatever goes wrong in it must not cost the run its headline line)
            # one host-array apply(), end to end: what every caller of the class API gets (float32 frame in, float64 out as the reference
            # returns it; and float32 out); PCIe floor = the frame's float32 bytes in, then out (a single frame cannot overlap its own directions
            # without being cut into row bands)
            # (one process per GPU runs next to it: for this leg the thread is bound to the GPU's NUMA node, so that the arrays it allocates are
            # first touched there; the affinity is restored afterwards - the cpu_baseline leg uses every core of the box)
            bound_node = -1 if args.no_bind else _native.bind_to_device_node(device)
            host_image = np.array(band_image, np.float32)[None]
            e2e = {}
            for label, dt in (("f32_to_f64", np.float64), ("f32_to_f32", np.float32)):
                ms, _, probe = e2e_host_frames(plan, host_image, _native.PAD_MODES[pad], dt)
                e2e[label] = round(ms, 4)
            e2e["f32_to_f32_pinned_arrays"] = round(e2e_host_frames(plan, host_image, _native.PAD_MODES[pad], np.float32, pinned=True)[0], 4)
            line["e2e_ms"] = e2e["f32_to_f64"]
            # Floors: `serial` = the frame's float32 bytes in, then out (what an uncut frame pays); `duplex` = both directions at once (rpsf_pcie_probe
            # on two streams) - a frame cut into B row bands cannot have less than duplex x (1 + 1/B): band 0 must be in before anything can go out.
            bands = plan.host_bands()
            duplex_floor = probe["duplex_ms"] * (1 + 1 / bands) if bands >= 2 else probe["h2d_ms"] + probe["d2h_ms"]
            line["e2e"] = {"what": "one ArrayPSFTransform.apply-style call on host arrays (rpsf_apply_host: the frame cut into row bands; one job of the persistent host "
                                   "pool stages the rows and widens landed bands while H2D, the bands' patch launches and D2H run), best of 5, result buffer reused",
                           "ms": e2e, "pcie": {k: round(v, 4) for k, v in probe.items()}, "row_bands": bands,
                           "pcie_floor_ms": round(duplex_floor, 4), "pcie_floor_serial_ms": round(probe["h2d_ms"] + probe["d2h_ms"], 4),
                           "pcie_floor_what": "duplex_ms x (1 + 1/row_bands): both directions at once, plus the first band that nothing can overlap",
                           "host_threads": _native.host_threads(),
                           "over_pcie_floor": round(e2e["f32_to_f64"] / duplex_floor, 3),
                           "over_pcie_floor_pinned_arrays": round(e2e["f32_to_f32_pinned_arrays"] / duplex_floor, 3),
                           "process_bound_to_numa_node": bound_node, "device_numa_node": _native.device_numa_node(device)}
            del host_image
        except Exception as e:  # noqa: BLE001
            line["e2e_error"] = f"{type(e).__name__}: {e}"
            print(f"[bench] end-to-end leg failed: {line['e2e_error']}", file=sys.stderr, flush=True)
        finally:
            os.sched_setaffinity(0, affinity)
    traffic_file = ROOT / "profiles" / ("traffic_latest.json" if args.config == 3 else f"traffic_config{args.config}.json")
    if world == 1 and traffic_file.exists():  # PMC counters cannot be read from inside the process
        tr = json.loads(traffic_file.read_text())
        line["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
        line["roofline"]["traffic_source"] = tr["source"]
    if not args.no_cpu and world == 1:
        line["cpu_baseline"] = cpu_baseline(band_image, coords, kernel_for(list(range(len(coords)))))
    emit_json(line)
    if second_leg_note is not None:  # the headline leg is complete and reported; the failed second leg must still show in the exit code
        sys.stdout.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
