"""The streamed host path (rpsf_apply_frames_host / ArrayPSFTransform.apply_batch): host frames in, host frames out on three
streams and the persistent worker pool - bit-identical to the frame-by-frame loop a user of the reference writes
(`[transform.apply(image) for image in images]`, regularizepsf/transform.py:85-177), and against the oracle."""

import pathlib
import threading

import numpy as np
import pytest

import regularizepsf_amd as rp
from oracle import regpsf_oracle as orc
from regularizepsf_amd import _native

ROOT = pathlib.Path(__file__).resolve().parent.parent

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: 1e-5 relative float32 (max|d| <= TOL max|ref| and rel. L2 <= TOL)


def check(out, ref):
    d = out.astype(np.float64) - ref
    assert np.abs(d).max() <= TOL * np.abs(ref).max()
    assert np.linalg.norm(d) <= TOL * np.linalg.norm(ref)


def _case(n, shape, frames, seed):
    rng = np.random.default_rng(seed)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = ((rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))) * 0.2).astype(np.complex64)
    images = (rng.standard_normal((frames, *shape)) * 10 + 100).astype(np.float32)
    return coords, k, images


def test_config5_eight_2048_frames_streamed_equals_the_loop_and_the_oracle():
    """BASELINE config 5 (one GPU's share): 8 starfields of 2048^2, 128-px patches, one coma K - streamed == loop bit for bit,
    first and last frame against the float64 oracle."""
    h = w = 2048
    n, frames = 128, 8
    coords, k = orc.synthetic_transfer(h, w, n, alpha=3.0, epsilon=0.1)
    images = np.stack([orc.starfield(h, w, 100 + i) for i in range(frames)])
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    streamed = t.apply_batch(images)
    assert streamed.dtype == np.float64 and streamed.shape == images.shape
    for f in range(frames):
        assert np.array_equal(streamed[f], t.apply(images[f])), f
    for f in (0, frames - 1):
        check(streamed[f], orc.apply_transfer(images[f], coords, k, workers=-1))


@pytest.mark.parametrize(("n", "shape", "frames", "group", "depth"), [
    (32, (96, 80), 11, 1, 4), (32, (96, 80), 11, 3, 2), (64, (200, 256), 7, 2, 3), (128, (384, 512), 9, 1, 1),
    (128, (384, 512), 9, 4, 4), (256, (512, 768), 5, 1, 4), (256, (512, 768), 5, 2, 2), (20, (50, 60), 6, 4, 3)])
def test_streamed_groups_depths_and_dtypes(n, shape, frames, group, depth):
    """Every group size / pipeline depth (short last group, slots reused several times), float32 and float64 on either
    side, stacks and lists of arrays: the streamed result is the loop's, bit for bit."""
    coords, k, images = _case(n, shape, frames, 13 * n + frames)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    t._device_plan().set_option("stream_group", group)  # (plan options, include/rpsf.h: the shipped library reads no such thing from the environment)
    t._device_plan().set_option("stream_depth", depth)
    loop = np.stack([t.apply(im) for im in images])
    if n in _native.SUPPORTED_PATCH_SIZES:
        same = np.array_equal
    else:  # the hipFFT fallback adds with float atomics: the order of the four contributions differs from run to run
        def same(a, b):
            return np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    assert same(t.apply_batch(images), loop)
    assert same(t.apply_batch(images.astype(np.float64)), loop)  # float64 frames are narrowed inside the library
    out32 = t.apply_batch(list(images), dtype=np.float32)
    assert out32.dtype == np.float32 and same(out32, loop.astype(np.float32))
    into = np.full(loop.shape, -1.0)
    assert t.apply_batch([im for im in images], out=into) is into and same(into, loop)
    check(loop[0], orc.apply_transfer(images[0], coords, k))
    # a single apply between two batches on the same plan, and a batch of one
    assert same(t.apply(images[1]), loop[1])
    assert same(t.apply_batch(images[:1]), loop[:1])
    assert same(t.apply_batch(images[2:]), loop[2:])


def test_no_thread_is_created_per_call():
    """The conversions run on the library's persistent pool: the thread count of the process stays where the first call left it."""
    def threads():
        for line in open("/proc/self/status"):
            if line.startswith("Threads:"):
                return int(line.split()[1])
        return -1

    coords, k, images = _case(64, (256, 256), 6, 3)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    t.apply(images[0])
    t.apply_batch(images)
    before = threads()
    for _ in range(10):
        t.apply(images[0])
        t.apply_batch(images)
    assert threads() == before
    assert 1 <= _native.host_threads() <= 256


def test_two_plans_stream_from_two_threads():
    """Distinct plans from distinct threads share the worker pool: jobs interleave, results stay the loop's."""
    cases = [_case(64, (192, 256), 6, 5), _case(128, (256, 384), 5, 6)]
    transforms = [rp.ArrayPSFTransform(rp.IndexedCube(c, k)) for c, k, _ in cases]
    loops = [np.stack([t.apply(im) for im in case[2]]) for t, case in zip(transforms, cases)]
    errors = []

    def work(i):
        try:
            for _ in range(8):
                if not np.array_equal(transforms[i].apply_batch(cases[i][2]), loops[i]):
                    errors.append(f"plan {i}: streamed result differs")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    pool = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    assert not errors, errors


def test_page_locked_arrays_skip_the_staging_copy_and_change_nothing():
    """Frames and results in rp.pinned_empty arrays: float32 sides go over PCIe in place (no pool job), float64 sides still convert;
    every combination gives the loop's pixels."""
    coords, k, images = _case(128, (384, 512), 7, 17)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    loop = np.stack([t.apply(im) for im in images])
    pin_in = rp.pinned_empty(images.shape, np.float32)
    pin_in[...] = images
    pin_out32 = rp.pinned_empty(images.shape, np.float32)
    pin_out64 = rp.pinned_empty(images.shape, np.float64)
    pin_out32[...] = -1
    pin_out64[...] = -1
    assert t.apply_batch(pin_in, out=pin_out32) is pin_out32 and np.array_equal(pin_out32, loop.astype(np.float32))
    assert np.array_equal(t.apply_batch(pin_in, out=pin_out64), loop)
    assert np.array_equal(t.apply_batch(images, out=pin_out32), loop.astype(np.float32))  # pageable in, pinned out
    assert np.array_equal(t.apply_batch(pin_in), loop)  # pinned in, pageable float64 out
    assert np.array_equal(t.apply_batch([pin_in[0], images[1], pin_in[2]], dtype=np.float32), loop[:3].astype(np.float32))  # mixed: staged
    single = rp.pinned_empty(images.shape[1:], np.float32)
    plan = t._device_plan()
    plan.apply_host(pin_in[3], _native.PAD_MODES["symmetric"], out=single)
    assert np.array_equal(single, loop[3].astype(np.float32))
    view = pin_in[1:5]  # a view keeps the allocation alive after the owner's name is gone
    del pin_in
    assert np.array_equal(t.apply_batch(view, dtype=np.float32), loop[1:5].astype(np.float32))


def test_pcie_probe_reports_plausible_rates():
    pr = _native.pcie_probe(16 << 20, 3)
    for key in ("h2d_ms", "d2h_ms", "duplex_ms"):
        assert 0.05 < pr[key] < 50.0, pr  # 16 MiB: 0.3 ms at 55 GB/s
    assert pr["duplex_ms"] >= 0.8 * max(pr["h2d_ms"], pr["d2h_ms"])


def test_streamed_errors_are_reported():
    coords, k, images = _case(32, (64, 64), 3, 9)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    with pytest.raises(ValueError):
        t.apply_batch([images[0], images[1][:32]])
    with pytest.raises(ValueError):
        t.apply_batch(images[0])
    plan = _native.Plan(32, coords)
    with pytest.raises(_native.NativeError):  # no transfer kernel installed
        plan.apply_frames_host(images, _native.PAD_MODES["symmetric"])


@pytest.mark.parametrize(("n", "shape", "pad_mode"), [(256, (3072, 2048), "symmetric"), (128, (2560, 2304), "reflect"),
                                                      (64, (2200, 3000), "constant"), (256, (4096, 4096), "symmetric")])
def test_a_large_frame_cut_into_row_bands_is_bit_identical_to_the_whole_frame(n, shape, pad_mode):
    """Host frames of 24 MiB and more are cut into row bands (views of the plan that share its K) so that upload, patches and
    download of ONE frame overlap; the bands take their plane colours from the whole lattice, so the result is the whole-frame
    apply's bit for bit - with two, four or eight bands, odd shapes, and from page-locked arrays."""
    coords, k, images = _case(n, shape, 1, 5 * n)
    image = images[0]
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    t._device_plan().set_option("host_bands", 0)
    whole = t.apply(image, pad_mode=pad_mode)
    check(whole, orc.apply_transfer(image, coords, k, pad_mode=pad_mode, workers=-1))
    for bands in (2, 4, 8):
        t._device_plan().set_option("host_bands", bands)  # (setting an option drops the bands built for the setting before)
        assert np.array_equal(t.apply(image, pad_mode=pad_mode), whole), bands
        assert np.array_equal(t.apply(image.astype(np.float64), pad_mode=pad_mode), whole), bands
    t._device_plan().set_option("host_bands", -1)
    assert np.array_equal(t.apply(image, pad_mode=pad_mode), whole)  # the default choice for this size
    pin = rp.pinned_empty(shape, np.float32)
    pin[...] = image
    pout = rp.pinned_empty(shape, np.float32)
    plan = t._device_plan()
    plan.apply_host(pin, _native.PAD_MODES[pad_mode], out=pout)
    assert np.array_equal(pout, whole.astype(np.float32))
    # the same plan on another frame shape afterwards (the bands are rebuilt for it: the corner list reaches beyond the smaller frame,
    # which the reference allows as long as the corners stay inside its 2N pad), then the first shape again
    small = np.ascontiguousarray(image[: shape[0] - n // 2, : shape[1] - n // 2])
    check(t.apply(small, pad_mode=pad_mode), orc.apply_transfer(small, coords, k, pad_mode=pad_mode, workers=-1))
    assert np.array_equal(t.apply(image, pad_mode=pad_mode), whole)


@pytest.mark.parametrize("threads", ["1", "2", "5"])
def test_a_banded_frame_with_a_narrow_host_pool(threads):
    """RPSF_HOST_THREADS = 1 leaves the host pool without workers: the calling thread stages, conducts and widens by itself (a pool job of one part runs its
    `meanwhile` hook BEFORE the part - the conductor of a banded frame would wait for staging that has not begun); two and five threads take the pool's path
    with an odd number of parts.  The pool is created once per process, so each width runs in a process of its own."""
    import os
    import subprocess
    import sys

    code = """
import numpy as np, regularizepsf_amd as rp
rng = np.random.default_rng(3)
n, shape = 64, (1500, 1300)
coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
k = ((rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))) * 0.2).astype(np.complex64)
image = (rng.standard_normal(shape) * 10 + 100).astype(np.float32)
t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
t._device_plan().set_option("host_bands", 0)
whole = t.apply(image)
for bands in (2, 5, 8):
    t._device_plan().set_option("host_bands", bands)
    assert np.array_equal(t.apply(image), whole), bands
    assert np.array_equal(t.apply(image.astype(np.float64)), whole), bands
    assert t._device_plan().host_bands() == bands
pin, pout = rp.pinned_empty(shape, np.float32), rp.pinned_empty(shape, np.float32)
pin[...] = image
from regularizepsf_amd import _native
t._device_plan().apply_host(pin, _native.PAD_MODES["symmetric"], out=pout)
assert np.array_equal(pout, whole.astype(np.float32))
print("ok", _native.host_threads())
"""
    env = dict(os.environ, RPSF_HOST_THREADS=threads)
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert done.returncode == 0, done.stderr[-2000:]
    assert done.stdout.strip().endswith(f"ok {threads}"), done.stdout


def test_long_sequences_of_large_frames_go_two_by_two_in_the_middle():
    """From twelve frames of more than 8 MiB on, the groups in the middle of the sequence hold two frames (a staging job and an enqueue per two
    frames), the first and last ones a single frame: 13 frames of 1536^2 (groups 1 1 2 2 2 2 1 1 1), float32 and float64 out - the loop's pixels,
    bit for bit, every frame in its place."""
    coords, k, images = _case(128, (1536, 1536), 13, 77)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    loop = np.stack([t.apply(im) for im in images])
    assert np.array_equal(t.apply_batch(images), loop)
    assert np.array_equal(t.apply_batch(list(images), dtype=np.float32), loop.astype(np.float32))
    check(loop[12], orc.apply_transfer(images[12], coords, k, workers=-1))
