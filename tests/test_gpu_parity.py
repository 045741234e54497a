"""Parity of the HIP path (through the C ABI) against the reference-generated goldens and the CPU oracle.

Tolerance (BASELINE.json north_star "within 1e-5 relative float32"): the kernel
computes in float32, the reference in float64, so the bar is
    max|d| <= 1e-5 * max|ref|   and   ||d||_2 <= 1e-5 * ||ref||_2 .
"""

import numpy as np
import pytest

import regularizepsf_amd as rp
from oracle import regpsf_oracle as orc
from tests.helpers import APPLY_CASES, GOLDEN, load_apply_case, load_c128_case, make_psfs, rel_errors

pytestmark = pytest.mark.gpu
TOL = 1e-5


def check(out, ref, tol=TOL):
    assert out.shape == ref.shape and out.dtype == np.float64
    rel_max, rel_l2 = rel_errors(out, ref)
    assert rel_max <= tol and rel_l2 <= tol, (rel_max, rel_l2)
    return rel_max, rel_l2


@pytest.mark.parametrize("case", [c[0] for c in APPLY_CASES])
def test_apply_matches_reference_golden(case):
    fx, coords, k = load_apply_case(case)
    transform = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    image = fx["image"].copy()
    out = transform.apply(image, pad_mode=str(fx["pad_mode"]))
    assert np.array_equal(image, fx["image"])  # input never mutated (transform.py:116-117)
    check(out, fx["expected"])


def test_upstream_identity_pin_full_size():
    """The reference's own end-to-end pin, tests/test_transform.py:29-49, at its full 2048^2 / 256 size."""
    size = 256
    x = np.arange(size, dtype=float)
    gauss = np.exp(-4 * np.log(2) * ((x[None, :] - size // 2) ** 2 + (x[:, None] - size // 2) ** 2) / 3**2)
    covering = [tuple(int(v) for v in t) for t in rp.calculate_covering((2048, 2048), size)]
    values = np.zeros((len(covering), size, size), np.float32)
    values[:] = gauss / np.sum(gauss)
    source = rp.ArrayPSF(rp.IndexedCube(covering, values))
    t = rp.ArrayPSFTransform.construct(source, source, 3.0, 0.1)
    image = np.zeros((2048, 2048), np.float32)
    image[500:1000, 200:400] = 5
    out = t.apply(image)
    assert np.allclose(image, out, atol=1e-3)
    assert abs(np.abs(out - image).max() - 5 * (1 - 1 / (1 + 0.1**4))) < 1e-5


def test_saturation_matches_reference_golden():
    fx = np.load(GOLDEN / "apply_saturation.npz")
    coords = [tuple(int(v) for v in t) for t in fx["coords"]]
    src, _ = make_psfs("identity", coords, 64, 192, 192)
    s_fft = orc.psf_fft(src)
    k = orc.construct_transfer(s_fft, s_fft, 3.0, 0.1)
    transform = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    for dil, nbw in ((1, 7), (2, 5), (0, 7)):
        out = transform.apply(fx["image"], saturation_threshold=10, saturation_dilation=dil, neighborhood_width=nbw)
        ref = fx[f"expected_d{dil}_w{nbw}"]
        assert np.array_equal(np.isnan(out), np.isnan(ref))
        good = ~np.isnan(ref)
        assert np.abs(out[good] - ref[good]).max() <= 1e-5 * np.abs(ref[good]).max()
        assert out[80, 80] == 100  # restored raw value (tests/test_transform.py:73-74)


@pytest.mark.parametrize("pad_mode", ["mean", "maximum", "linear_ramp"])
def test_host_padded_modes_match_oracle(pad_mode):
    fx, coords, k = load_apply_case("n32_sym")
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply(fx["image"], pad_mode=pad_mode)
    check(out, orc.apply_transfer(fx["image"], coords, k, pad_mode=pad_mode))


def test_construct_matches_reference_golden():
    fx = np.load(GOLDEN / "construct.npz")
    coords = [(i, 2 * i) for i in range(6)]
    for dt in ("float32", "float64"):
        s = rp.ArrayPSF(rp.IndexedCube(coords, fx["rand_values_s"].astype(dt)))
        t = rp.ArrayPSF(rp.IndexedCube(coords, fx["rand_values_t"].astype(dt)))
        assert np.array_equal(s.fft_evaluations, fx[f"rand_fft_{dt}"])  # host scipy path: bit exact
        for alpha in (0.5, 1.0, 2.0, 3.0):
            for eps in (0.1, 0.01):
                k = rp.ArrayPSFTransform.construct(s, t, alpha, eps)._transfer_kernel.values
                ref = fx[f"rand_{dt}_a{alpha}_e{eps}"]
                assert k.dtype == ref.dtype
                assert np.abs(k - ref).max() <= 1e-5 * np.abs(ref).max()


def test_construct_degenerate_bins_follow_reference():
    """NaN / 0 / Inf pattern of the known-answer table, float32 and float64."""
    from regularizepsf_amd import _native

    fx = np.load(GOLDEN / "construct.npz")
    for dt, cdt in (("float32", np.complex64), ("float64", np.complex128)):
        for alpha in (0.5, 1.0, 2.0, 3.0):
            for eps in (0.1, 0.01):
                k = _native.build_transfer(fx["table_s"].astype(cdt), fx["table_t"].astype(cdt), alpha, eps)
                ref = fx[f"table_{dt}_a{alpha}_e{eps}"]
                assert np.array_equal(np.isnan(k), np.isnan(ref)), (dt, alpha, eps, k, ref)
                good = ~np.isnan(ref)
                assert np.allclose(k[good], ref[good], rtol=2e-6 if dt == "float32" else 1e-12, atol=0)


def test_gpu_psf_fft_matches_scipy():
    rng = np.random.default_rng(3)
    for n in (16, 32, 64, 128, 256):
        vals = rng.random((5, n, n)).astype(np.float32) ** 3
        coords = [(i, i) for i in range(5)]
        got = rp.ArrayPSF(rp.IndexedCube(coords, vals), device=0).fft_evaluations
        ref = orc.psf_fft(vals.astype(np.float64))
        assert got.dtype == np.complex64
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_config1_512_n32_through_the_class_api():
    """BASELINE.json configs[0] (the plumbing configuration: 512 x 512 starfield seed 1, 32-px patches, 1089 of them, constant
    Gaussian PSF 1.8 -> 1.5) through ArrayPSF / ArrayPSFTransform.construct / apply on the GPU, against the reference's own output
    (every 8th pixel stored in tests/golden/config1_512_n32.npz; the whole frame through the oracle, whose SHA-256 the CPU suite
    pins to the reference's)."""
    fx = np.load(GOLDEN / "config1_512_n32.npz")
    h, w, n = (int(v) for v in fx["meta"])
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((h, w), n)]
    assert len(coords) == 1089
    src, tgt = make_psfs("gauss", coords, n, h, w)
    transform = rp.ArrayPSFTransform.construct(rp.ArrayPSF(rp.IndexedCube(coords, src)), rp.ArrayPSF(rp.IndexedCube(coords, tgt)),
                                               float(fx["alpha"]), float(fx["eps"]))
    image = orc.starfield(h, w, int(fx["seed"]))
    out = transform.apply(image)
    assert out.dtype == np.float64 and out.shape == (h, w)
    _, k = orc.synthetic_transfer(h, w, n, alpha=float(fx["alpha"]), epsilon=float(fx["eps"]), kind="gauss")
    ref = orc.apply_transfer(image, coords, k)
    check(out, ref)
    assert np.abs(out[::8, ::8] - fx["sample"]).max() <= TOL * np.abs(ref).max()
    assert np.array_equal(transform.apply(2 * image), 2 * out)  # linear; exactly so for a power of two


def test_config2_2048_n128_against_oracle():
    """BASELINE.json configs[1]: 2048^2, 128-pixel patches, slowly varying coma PSF grid."""
    coords, k = orc.synthetic_transfer(2048, 2048, 128, alpha=3.0, epsilon=0.1)
    image = orc.starfield(2048, 2048, seed=2)
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply(image)
    check(out, orc.apply_transfer(image, coords, k, workers=-1))


def test_config3_4096_n256_against_oracle_and_properties():
    """BASELINE.json configs[2] (headline) at full size: oracle parity, linearity, run-to-run stability."""
    coords, k = orc.synthetic_transfer(4096, 4096, 256, alpha=3.0, epsilon=0.1)
    image = orc.starfield(4096, 4096, seed=3)
    transform = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    out = transform.apply(image)
    check(out, orc.apply_transfer(image, coords, k, workers=-1))
    # linearity: apply(a*x + y) == a*apply(x) + apply(y) to float32 round-off
    other = orc.starfield(4096, 4096, seed=33)
    lhs = transform.apply(0.5 * image + other)
    rhs = 0.5 * out + transform.apply(other)
    assert np.linalg.norm(lhs - rhs) <= 2e-6 * np.linalg.norm(rhs)
    # the overlap-add runs in a fixed order (direct accumulation in colour order): bit-reproducible
    assert np.array_equal(transform.apply(image), out)


def test_sparse_and_offlattice_coordinates():
    """Arbitrary corner lists: uncovered pixels stay 0, odd offsets work (no lattice assumption)."""
    n = 32
    rng = np.random.default_rng(9)
    coords = [(-17, 3), (5, 40), (31, -9), (60, 61)]
    k = (rng.standard_normal((4, n, n)) + 1j * rng.standard_normal((4, n, n))).astype(np.complex64)
    image = rng.standard_normal((90, 100)).astype(np.float32)
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply(image)
    ref = orc.apply_transfer(image, coords, k)
    check(out, ref)
    assert np.all(out[ref == 0] == 0)


def test_errors_follow_reference_conventions():
    n = 32
    k = np.ones((1, n, n), np.complex64)
    with pytest.raises(ValueError):  # ragged np.stack in the reference
        rp.ArrayPSFTransform(rp.IndexedCube([(500, 0)], k)).apply(np.zeros((64, 64), np.float32))
    with pytest.raises(TypeError):  # float slice index in the reference
        rp.ArrayPSFTransform(rp.IndexedCube([(0.5, 0)], k)).apply(np.zeros((64, 64), np.float32))
    with pytest.raises(ValueError):
        rp.ArrayPSFTransform(rp.IndexedCube([(0, 0)], k)).apply(np.zeros((4, 64, 64), np.float32))
    out = rp.ArrayPSFTransform(rp.IndexedCube([(0, 0)], k)).apply(np.arange(64 * 64).reshape(64, 64))  # ints ok
    assert out.dtype == np.float64


@pytest.mark.parametrize("overlap", [False, True, "pipeline"])
@pytest.mark.parametrize("world", [2, 3])
def test_row_bands_on_one_gpu(world, overlap):
    """The multi-GPU band geometry (resident row windows, owned/spill rows) run band by band on ONE GPU;
    the seam add is done on the host here, RCCL does it in production (regularizepsf_amd/sharding.py).
    overlap=True: the spill rows come from a plan of their own (the band's last lattice row, launched first on its own
    stream so that the transfer can run beside the rest of the band); overlap=False: one plan, spill rows at the end of its buffer."""
    from regularizepsf_amd.sharding import ShardedApply

    h, w, n = 1024, 768, 128
    coords, k = orc.synthetic_transfer(h, w, n, alpha=1.0, epsilon=0.1)
    image = orc.starfield(h, w, seed=5)
    ref = orc.apply_transfer(image, coords, k, workers=-1)
    own, spill, bands = [], [], []
    for rank in range(world):
        sh = ShardedApply(coords, lambda idx: k[idx], n, h, w, rank, world, 0, None, overlap=overlap)
        b = sh.band
        assert sh.overlap == (overlap is True) and (sh.seam_plan is not None) == (overlap is True and b.send_rows > 0)
        assert sh.pipeline == (overlap == "pipeline" and b.send_rows + b.recv_rows > 0)
        sh.upload_rows(image[b.image_row0 : b.image_row0 + b.image_rows])
        sh.step()
        own.append(sh.owned_rows().astype(np.float64))
        spill.append(sh.spill_rows().astype(np.float64))
        bands.append(b)
    for g in range(1, world):
        own[g][: bands[g].recv_rows] += spill[g - 1]
    check(np.concatenate(own), ref)
    if overlap is not False:
        return
    # seam="recompute": every band also runs the patches above it that reach into its rows; nothing to add afterwards
    parts = []
    for rank in range(world):
        sh = ShardedApply(coords, lambda idx: k[idx], n, h, w, rank, world, 0, None, seam="recompute")
        b = sh.band
        assert b.send_rows == 0 and b.recv_rows == 0 and b.out_rows == b.own_rows
        sh.upload_rows(image[b.image_row0 : b.image_row0 + b.image_rows])
        sh.step()
        parts.append(sh.owned_rows().astype(np.float64))
    check(np.concatenate(parts), ref)


def test_rccl_wrapper_single_rank():
    """The dlopen'ed RCCL path of the C ABI (rpsf_comm_*) with world size 1: id, init, all-reduce, no-op seam."""
    from regularizepsf_amd import _native

    uid = _native.Comm.unique_id()
    assert len(uid) == 128
    comm = _native.Comm(0, 0, 1, uid)
    assert comm.allreduce_max(3.25) == 3.25
    assert comm.ranks() == 1  # ncclCommCount through rpsf_comm_ranks: what bench.py --gpus N prints as rccl_ranks
    comm.barrier()
    buf = _native.DeviceBuffer(4096)
    comm.seam_exchange_add(buf.ptr, 1024, buf.ptr, 0, buf.ptr)  # no neighbour: nothing to send or receive
    comm.close()


def test_forced_atomic_mode_matches_planes():
    """Both overlap-add strategies (colour planes + sum kernel, float atomics) give the same image."""
    from regularizepsf_amd import _native

    fx, coords, k = load_apply_case("n64_sym")
    image = np.ascontiguousarray(fx["image"], np.float32)
    outs = {}
    for mode in ("planes", "atomic"):
        plan = _native.Plan(64, coords)
        plan.set_transfer(k)
        plan.set_overlap_mode(mode)
        outs[mode] = plan.apply(image, _native.PAD_MODES["symmetric"]).astype(np.float64)
        check(outs[mode], fx["expected"])
    assert np.abs(outs["planes"] - outs["atomic"]).max() <= 2e-6 * np.abs(fx["expected"]).max()


@pytest.mark.parametrize(("n", "shape"), [(256, (2048, 2048)), (128, (2048, 2048)), (128, (700, 1000)), (256, (900, 520))])
def test_direct_overlap_add_matches_planes_and_is_reproducible(n, shape, monkeypatch):
    """The three overlap-add strategies on a lattice - direct accumulation through the XCD's L2 (default for 128/256-px
    patches), colour planes + plane sum, float atomics - agree to round-off; the direct one is bit-reproducible, and
    so is its run-time demotion path (RPSF_OPT_DEBUG_ORPHAN makes every third workgroup behave as if it had been placed on
    a foreign XCD: its tiles go through the colour planes and the fix-up kernel)."""
    from regularizepsf_amd import _native

    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    rng = np.random.default_rng(n + shape[0])
    k = (rng.standard_normal((len(coords), n, n), dtype=np.float32)
         + 1j * rng.standard_normal((len(coords), n, n), dtype=np.float32)).astype(np.complex64)
    image = rng.standard_normal(shape, dtype=np.float32)
    pad = _native.PAD_MODES["reflect"]
    outs = {}
    for mode in ("direct", "planes", "atomic", "orphans"):
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        plan.set_overlap_mode("direct" if mode == "orphans" else mode)
        if mode == "orphans":
            plan.set_option("debug_orphan", 3)
        outs[mode] = plan.apply(image, pad)
        if mode in ("direct", "orphans"):
            for _ in range(3):
                assert np.array_equal(plan.apply(image, pad), outs[mode]), mode
    scale = np.abs(outs["planes"]).max()
    assert np.isfinite(outs["direct"]).all()
    for mode in ("direct", "atomic", "orphans"):
        assert np.abs(outs[mode] - outs["planes"]).max() <= 2e-6 * scale, mode


@pytest.mark.parametrize("pad_mode", ["symmetric", "reflect", "edge", "wrap", "constant"])
@pytest.mark.parametrize("shape", [(5, 7), (1, 1), (33, 9)])
def test_images_smaller_than_a_patch(shape, pad_mode):
    """np.pad reflects repeatedly when the pad is wider than the image; the in-kernel index maps must agree."""
    n = 16
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = rng.standard_normal(shape).astype(np.float32) + 3
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply(image, pad_mode=pad_mode)
    check(out, orc.apply_transfer(image, coords, k, pad_mode=pad_mode))


def test_input_layouts_and_dtypes():
    fx, coords, k = load_apply_case("n32_sym")
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    base = fx["image"]
    ref = fx["expected"]
    wide = np.zeros((base.shape[0], 2 * base.shape[1]), np.float32)
    wide[:, ::2] = base
    check(t.apply(wide[:, ::2]), ref)                 # non-contiguous view
    check(t.apply(base.astype(np.float64)), ref)      # float64 input
    check(t.apply(np.asfortranarray(base)), ref)      # Fortran order
    ints = np.round(base).astype(np.int32)
    check(t.apply(ints), orc.apply_transfer(ints, coords, k))  # integer input is promoted like the reference does
    k128 = k.astype(np.complex128)                     # complex128 kernels are accepted (rounded to complex64 on upload)
    check(rp.ArrayPSFTransform(rp.IndexedCube(coords, k128)).apply(base), ref)


def test_editing_the_kernel_invalidates_the_device_copy():
    fx, coords, k = load_apply_case("n32_sym")
    cube = rp.IndexedCube(coords, k.copy())
    t = rp.ArrayPSFTransform(cube)
    first = t.apply(fx["image"])
    cube[coords[0]] = np.zeros((32, 32), np.complex64)       # __setitem__ is tracked
    second = t.apply(fx["image"])
    k2 = k.copy()
    k2[0] = 0
    check(second, orc.apply_transfer(fx["image"], coords, k2))
    assert np.abs(first - second).max() > 0


def test_kernel_fft_agrees_with_hipfft():
    """rocFFT/hipFFT as an independent on-device cross-check of the hand-written FFT (never used by the product)."""
    import ctypes

    from regularizepsf_amd import _native

    try:
        hipfft = ctypes.CDLL("libhipfft.so")
    except OSError:
        pytest.skip("libhipfft.so not available")
    n, count = 128, 3
    rng = np.random.default_rng(11)
    vals = rng.standard_normal((count, n, n)).astype(np.float32)
    ours = _native.psf_fft(vals)
    z = vals.astype(np.complex64)
    d_in = _native.DeviceBuffer(z.nbytes).upload(z)
    d_out = _native.DeviceBuffer(z.nbytes)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    hipfft.hipfftPlanMany.restype = ci  # hipfftHandle is a pointer: declare every prototype explicitly
    hipfft.hipfftPlanMany.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ci, ci,
                                      ctypes.POINTER(ci), ci, ci, ci, ci]
    hipfft.hipfftExecC2C.restype = ci
    hipfft.hipfftExecC2C.argtypes = [vp, vp, vp, ci]
    hipfft.hipfftDestroy.restype = ci
    hipfft.hipfftDestroy.argtypes = [vp]
    plan = vp()
    HIPFFT_C2C, HIPFFT_FORWARD = 0x29, -1
    dims = (ci * 2)(n, n)
    rc = hipfft.hipfftPlanMany(ctypes.byref(plan), 2, dims, None, 1, n * n, None, 1, n * n, HIPFFT_C2C, count)
    assert rc == 0
    assert hipfft.hipfftExecC2C(plan, d_in.ptr, d_out.ptr, HIPFFT_FORWARD) == 0
    _native.check(_native.lib().rpsf_device_synchronize(0))
    theirs = d_out.download((count, n, n), np.complex64)
    hipfft.hipfftDestroy(plan)
    assert np.abs(ours - theirs).max() <= 2e-6 * np.abs(theirs).max()


# ------------------------------------------------------------------ batches of frames sharing one transfer kernel
def _random_case(n, shape, seed, frames):
    rng = np.random.default_rng(seed)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    images = (rng.standard_normal((frames, *shape)) * 10 + 100).astype(np.float32)
    return coords, k, images


@pytest.mark.parametrize(("n", "shape", "frames"), [(16, (70, 90), 5), (32, (128, 100), 3), (64, (200, 256), 4),
                                                     (128, (300, 420), 3), (256, (512, 512), 2)])
def test_batch_equals_frame_by_frame(n, shape, frames):
    """apply_batch is the Python loop over apply, bit for bit (same kernels, same plane-sum order), and the
    first frame is pinned against the CPU oracle."""
    coords, k, images = _random_case(n, shape, 7 * n + frames, frames)
    transform = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    batch = transform.apply_batch(images, pad_mode="reflect")
    assert batch.shape == images.shape and batch.dtype == np.float64
    for f in range(frames):
        assert np.array_equal(batch[f], transform.apply(images[f], pad_mode="reflect")), f
    check(batch[0], orc.apply_transfer(images[0], coords, k, pad_mode="reflect"))
    assert transform.apply_batch(images, dtype=np.float32).dtype == np.float32


def test_batch_config5_shape_against_oracle():
    """Config 5 of BASELINE.json scaled down: starfield frames, 128-px patches, coma K shared by the batch."""
    n, size, frames = 128, 512, 6
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((size, size), n)]
    src, tgt = make_psfs("coma", coords, n, size, size)
    k = orc.construct_transfer(orc.psf_fft(src), orc.psf_fft(tgt), 3.0, 0.1).astype(np.complex64)
    images = np.stack([orc.starfield(size, size, 100 + i) for i in range(frames)])
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply_batch(images)
    for f in range(frames):
        check(out[f], orc.apply_transfer(images[f], coords, k))


def test_batch_device_strides_modes_and_errors():
    from regularizepsf_amd import _native

    n, shape, frames = 32, (96, 80), 4
    coords, k, images = _random_case(n, shape, 3, frames)
    h, w = shape
    pad = _native.PAD_MODES["symmetric"]
    ref = None
    for mode in ("planes", "atomic"):
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        plan.set_overlap_mode(mode)
        # device variant with padded frame strides (frames are not back to back)
        stride = h * w + 64
        host = np.zeros((frames, stride), np.float32)
        host[:, : h * w] = images.reshape(frames, -1)
        d_in = _native.DeviceBuffer(host.nbytes).upload(host)
        d_out = _native.DeviceBuffer(host.nbytes)
        plan.apply_batch_device(d_in.ptr, d_out.ptr, frames, stride, stride, _native.Geometry.whole(h, w, pad))
        plan.synchronize()
        got = d_out.download((frames, stride))[:, : h * w].reshape(frames, h, w)
        single = np.stack([plan.apply(images[f], pad) for f in range(frames)])
        if mode == "planes":
            assert np.array_equal(got, single)
            ref = got
        else:  # float atomics: order of the four contributions differs run to run
            assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
        assert np.array_equal(plan.apply_batch(images, pad), single) or mode == "atomic"
        with pytest.raises(_native.NativeError):
            plan.apply_batch_device(d_in.ptr, d_out.ptr, frames, h * w - 1, stride, _native.Geometry.whole(h, w, pad))
        with pytest.raises(_native.NativeError):
            plan.apply_batch_device(d_in.ptr, d_out.ptr, 0, stride, stride, _native.Geometry.whole(h, w, pad))
        d_in.free()
        d_out.free()
    transform = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    with pytest.raises(ValueError):
        transform.apply_batch(images[0])
    # unsupported-in-kernel pad mode and the saturation branch fall back to the per-frame path
    out = transform.apply_batch(images[:2], pad_mode="mean", saturation_threshold=125.0)
    for f in range(2):
        assert np.array_equal(out[f], transform.apply(images[f], pad_mode="mean", saturation_threshold=125.0))


def test_construct_keeps_the_kernel_on_the_device():
    """complex64 spectra + supported patch size: K2 -> pack run on the device; same K and same correction as the
    host-pointer route, and the first apply does not upload K again."""
    from regularizepsf_amd import _native

    n, shape = 32, (96, 128)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    src, tgt = make_psfs("coma", coords, n, *shape)
    s = rp.ArrayPSF(rp.IndexedCube(coords, src.astype(np.float32)))
    t = rp.ArrayPSF(rp.IndexedCube(coords, tgt.astype(np.float32)))
    tr = rp.ArrayPSFTransform.construct(s, t, 3.0, 0.1)
    k = tr._transfer_kernel.values
    assert k.dtype == np.complex64 and tr._plan is not None
    assert np.array_equal(k, _native.build_transfer(s.fft_evaluations, t.fft_evaluations, 3.0, 0.1), equal_nan=True)
    ref = orc.construct_transfer(s.fft_evaluations, t.fft_evaluations, 3.0, 0.1)
    assert np.abs(k - ref).max() <= 1e-5 * np.abs(ref).max()
    plan_before = tr._plan
    image = orc.starfield(*shape, 5)
    out = tr.apply(image)
    assert tr._plan is plan_before  # no re-upload
    assert np.array_equal(out, rp.ArrayPSFTransform(rp.IndexedCube(coords, k.copy())).apply(image))
    check(out, orc.apply_transfer(image, coords, k))
    tr._transfer_kernel[coords[0]] = np.zeros((n, n), np.complex64)  # editing K still invalidates the device copy
    assert not np.array_equal(tr.apply(image), out)


def test_apply_host_converts_dtypes_inside_the_library():
    """float32 / float64 / integer images in, float64 (reference) or float32 out: same numbers as the float32 entry
    point, for a frame large enough to be cut into several chunks and threads and for a tiny one."""
    from regularizepsf_amd import _native

    pad = _native.PAD_MODES["symmetric"]
    for n, shape in ((64, (1100, 1300)), (16, (9, 7))):
        coords, k, images = _random_case(n, shape, 11 + n, 1)
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        ref = plan.apply(images[0], pad)
        for dt in (np.float32, np.float64, np.int16, ">f4"):
            img = images[0].astype(dt)
            expect = ref if dt != np.int16 else plan.apply(img.astype(np.float32), pad)
            out64 = plan.apply_host(img, pad)
            assert out64.dtype == np.float64 and np.array_equal(out64, expect.astype(np.float64)), (n, dt)
            out32 = plan.apply_host(img, pad, out_dtype=np.float32)
            assert out32.dtype == np.float32 and np.array_equal(out32, expect), (n, dt)
        strided = np.asfortranarray(images[0].astype(np.float64))
        assert np.array_equal(plan.apply_host(strided, pad), ref.astype(np.float64))


def test_distinct_plans_from_distinct_threads():
    """include/rpsf.h: a plan is not re-entrant, but distinct plans may be driven from distinct threads
    (ctypes releases the GIL during the calls, so these really overlap)."""
    import threading

    cases = []
    for i, (n, shape) in enumerate([(32, (200, 180)), (64, (256, 320)), (128, (384, 256))]):
        coords, k, images = _random_case(n, shape, 40 + i, 1)
        t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
        cases.append((t, images[0], t.apply(images[0])))
    errors = []

    def work(t, image, expect):
        try:
            for _ in range(8):
                if not np.array_equal(t.apply(image), expect):
                    errors.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=c) for c in cases]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("seed", range(24))
def test_randomized_shapes_pads_and_corner_lists(seed):
    """Seeded sweep over patch size, image shape (not a multiple of anything), pad mode and corner list
    (full lattice, thinned lattice, lattice shifted off the half-patch grid) against the CPU oracle."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([16, 32, 64, 128]))
    h, w = (int(v) for v in rng.integers(n // 2 + 1, 3 * n + 17, size=2))
    pad_mode = str(rng.choice(["symmetric", "reflect", "edge", "wrap", "constant"]))
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((h, w), n)]
    variant = seed % 3
    if variant == 1:  # thinned: still on the lattice, holes in the coverage
        keep = rng.random(len(coords)) < 0.7
        keep[0] = True
        coords = [c for c, k_ in zip(coords, keep) if k_]
    elif variant == 2:  # off-lattice corners: overlap-add goes through float atomics
        dr, dc = (int(v) for v in rng.integers(-n // 4, n // 4 + 1, size=2))
        coords = [(r + dr + int(rng.integers(0, 3)), c + dc) for r, c in coords]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = (rng.standard_normal((h, w)) * 20 + 50).astype(np.float32)
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply(image, pad_mode=pad_mode)
    check(out, orc.apply_transfer(image, coords, k, pad_mode=pad_mode))


@pytest.mark.parametrize(("n", "shape", "pad_mode"), [(24, (100, 130), "symmetric"), (9, (40, 33), "reflect"),
                                                        (100, (260, 300), "wrap"), (48, (96, 96), "constant"),
                                                        (7, (20, 20), "mean")])
def test_any_patch_size_goes_through_the_hipfft_fallback(n, shape, pad_mode):
    """The reference takes every square patch size, odd ones included (transform.py:151-164); sizes without a
    hand-written plan run gather -> hipFFT -> x K -> hipFFT -> overlap-add and meet the same tolerance."""
    rng = np.random.default_rng(n)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = (rng.standard_normal(shape) * 10 + 40).astype(np.float32)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    out = t.apply(image, pad_mode=pad_mode)
    check(out, orc.apply_transfer(image, coords, k, pad_mode=pad_mode))
    stack = t.apply_batch(np.stack([image, image[::-1].copy()]), pad_mode="symmetric")
    check(stack[1], orc.apply_transfer(image[::-1], coords, k))
    assert t.apply(image, saturation_threshold=55.0).shape == shape  # host-padded route (shifted origin) also works


@pytest.mark.parametrize(("n", "shape"), [(96, (700, 900)), (45, (300, 260))])
def test_the_fallback_adds_in_a_fixed_order_on_a_covering(n, shape):
    """On the corners calculate_covering lays (util.py:10-53; odd sizes too) the fallback's overlap-add goes colour class by colour class with plain
    adds - four passes per chunk, patches of one class do not overlap - so its results are bit-reproducible like the compiled sizes'; float atomics
    (mode "atomic", and every corner list without that structure) meet the tolerance but not the bits."""
    from regularizepsf_amd import _native

    rng = np.random.default_rng(n)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = (rng.standard_normal(shape) * 10 + 40).astype(np.float32)
    ref = orc.apply_transfer(image, coords, k)
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    pad = _native.PAD_MODES["symmetric"]
    first = plan.apply(image, pad)
    check(first.astype(np.float64), ref)
    for _ in range(6):
        assert np.array_equal(plan.apply(image, pad), first)
    # the same patches listed in another order: other chunks, the same colour order inside every pixel only if one chunk holds them all
    plan.set_overlap_mode("atomic")
    check(plan.apply(image, pad).astype(np.float64), ref)
    plan.set_overlap_mode("planes")
    assert np.array_equal(plan.apply(image, pad), first)
    # corners off any covering: atomics, within the tolerance
    moved = [(r + (3 if i % 5 == 0 else 0), c) for i, (r, c) in enumerate(coords)]
    other = _native.Plan(n, moved)
    other.set_transfer(k)
    with pytest.raises(_native.NativeError, match="lattice"):
        other.set_overlap_mode("planes")
    check(other.apply(image, pad).astype(np.float64), orc.apply_transfer(image, moved, k))


def test_apply_into_a_callers_array():
    """`apply(..., out=)` (keyword only; not in the reference): the result lands in the caller's array - float64 or float32 - on the library's default
    path (a small frame through the sweep kernel: the kernel's stores go straight to the page-locked result rows), through the saturation branch and
    through a host-padded mode; the array comes back, and a wrong shape is refused."""
    n, shape = 32, (200, 264)
    rng = np.random.default_rng(11)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = rng.standard_normal(shape) * 10 + 100
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    plain = t.apply(image)
    check(plain, orc.apply_transfer(image, coords, k))
    for dt in (np.float64, np.float32):
        out = np.full(shape, np.nan, dt)
        assert t.apply(image, out=out) is out
        assert np.array_equal(out, plain.astype(dt))
    hot = image.copy()
    hot[50, 60] = 1e6
    out = np.empty(shape)
    assert t.apply(hot, saturation_threshold=5e5, out=out) is out
    assert np.array_equal(out, t.apply(hot, saturation_threshold=5e5))
    assert t.apply(image, pad_mode="linear_ramp", out=out) is out
    assert np.array_equal(out, t.apply(image, pad_mode="linear_ramp"))
    with pytest.raises(ValueError):
        t.apply(image, out=np.empty((shape[0], shape[1] + 1)))


def test_integration_stub_call_sequence():
    """The reference-side binding shown in INTEGRATION.md, call for call (raw ctypes, no helper layer)."""
    import ctypes

    from regularizepsf_amd import build

    fx, coords, k = load_apply_case("n32_sym")
    lib = ctypes.CDLL(str(build.TARGET))
    lib.rpsf_last_error.restype = ctypes.c_char_p
    handle = ctypes.c_void_p()
    c = np.ascontiguousarray(np.array(coords, np.int32))
    assert lib.rpsf_plan_create(ctypes.byref(handle), 0, 32, len(c), c.ctypes.data_as(ctypes.c_void_p)) == 0, lib.rpsf_last_error()
    kk = np.ascontiguousarray(k, np.complex64)
    assert lib.rpsf_plan_set_transfer(handle, kk.ctypes.data_as(ctypes.c_void_p)) == 0, lib.rpsf_last_error()
    img = np.ascontiguousarray(fx["image"], float)
    out = np.empty_like(img)
    pad = {"constant": 0, "symmetric": 1, "reflect": 2, "edge": 3, "wrap": 4}[str(fx["pad_mode"])]
    rc = lib.rpsf_apply_host(handle, img.ctypes.data_as(ctypes.c_void_p), 1, img.shape[0], img.shape[1], pad,
                             ctypes.c_float(0.0), out.ctypes.data_as(ctypes.c_void_p), 1)
    assert rc == 0, lib.rpsf_last_error()
    check(out, fx["expected"])
    # ... the saturation branch in one call (apply_saturated of the stub) ...
    hot = img.copy()
    hot[10, 12] = hot[40, 3] = 1.0e6
    sat = np.empty_like(hot)
    rc = lib.rpsf_apply_host_saturated(handle, hot.ctypes.data_as(ctypes.c_void_p), 1, hot.shape[0], hot.shape[1], pad, ctypes.c_double(5.0e5), 1, 7,
                                       sat.ctypes.data_as(ctypes.c_void_p), 1)
    assert rc == 0, lib.rpsf_last_error()
    check(sat, orc.apply_transfer(hot, coords, k, pad_mode=str(fx["pad_mode"]), saturation_threshold=5.0e5))
    assert sat[10, 12] == 1.0e6
    # ... and the streamed loop over frames (apply_many of the stub)
    frames = [img, img[::-1].copy(), img * 2.0]
    stack = np.empty((3, *img.shape))
    ins = (ctypes.c_void_p * 3)(*[f.ctypes.data for f in frames])
    outs = (ctypes.c_void_p * 3)(*[stack[i].ctypes.data for i in range(3)])
    rc = lib.rpsf_apply_frames_host(handle, ins, 1, 3, img.shape[0], img.shape[1], pad, ctypes.c_float(0.0), outs, 1)
    assert rc == 0, lib.rpsf_last_error()
    assert np.array_equal(stack[0], out)
    check(stack[1], orc.apply_transfer(frames[1], coords, k, pad_mode=str(fx["pad_mode"])))
    lib.rpsf_plan_destroy(handle)


# ------------------------------------------------------------------ round 2: configs at full size, precision, seams, files
def test_complex128_kernel_against_the_reference_in_complex128():
    """float64 PSFs give a complex128 K and the reference multiplies in complex128 (transform.py:164); the GPU path rounds
    K to complex64 when it uploads it.  The golden was produced by the reference WITHOUT that rounding."""
    fx, coords, k = load_c128_case()
    out = rp.ArrayPSFTransform(rp.IndexedCube(coords, k)).apply(fx["image"])
    check(out, fx["expected"])
    # and through construct(): float64 PSFs in, complex128 K out (K2 in double), same result
    h, w, n = (int(v) for v in fx["meta"])
    src, tgt = make_psfs("coma", coords, n, h, w)
    tr = rp.ArrayPSFTransform.construct(rp.ArrayPSF(rp.IndexedCube(coords, src)), rp.ArrayPSF(rp.IndexedCube(coords, tgt)), 3.0, 0.1)
    assert tr._transfer_kernel.values.dtype == np.complex128
    check(tr.apply(fx["image"]), fx["expected"])


def test_config4_8192_n256_eight_bands_both_seam_modes():
    """BASELINE.json configs[3]: one 8192^2 frame, 256-px patches, cut into 8 row bands.  The bands run one at a time on
    this GPU (what eight ranks do concurrently); the seam rows are added with the library's own K4 entry point
    (rpsf_add_rows - the add RCCL's receiver runs), then checked against the CPU oracle; `recompute` needs no exchange."""
    from regularizepsf_amd import _native
    from regularizepsf_amd.sharding import ShardedApply, make_band_plans

    h = w = 8192
    n, world = 256, 8
    coords, k = orc.synthetic_transfer(h, w, n, alpha=3.0, epsilon=0.1)
    image = orc.starfield(h, w, seed=4)
    ref = orc.apply_transfer(image, coords, k, workers=-1)
    plans = make_band_plans(coords, n, h, world)
    assert sorted(len(b.patch_index) for b in plans) == [520] * 7 + [585]
    assert all(b.send_rows == 128 for b in plans[:-1]) and plans[-1].send_rows == 0
    out = np.empty((h, w), np.float64)
    prev = None  # (ShardedApply, band) of the rank above, kept alive for its spill rows
    for rank in range(world):
        sh = ShardedApply(coords, lambda idx: k[idx], n, h, w, rank, world, 0, None)
        b = sh.band
        sh.upload_rows(image[b.image_row0 : b.image_row0 + b.image_rows])
        sh.step()
        sh.synchronize()
        if prev is not None:  # what the receiving rank does with the rows RCCL delivers: K4 on its own output
            psh, pb = prev
            assert psh.overlap and psh.seam_plan is not None  # spill rows computed first, by a plan of their own
            assert psh.seam_once and psh.plan.n_patches + psh.seam_plan.n_patches == len(pb.patch_index)  # every patch runs once
            _native.add_rows(sh.d_out.ptr, psh.spill_ptr(), pb.send_rows * w)
            sh.synchronize()
        out[b.out_row0 : b.out_row0 + b.own_rows] = sh.d_out.download((b.own_rows, w))
        prev = (sh, b)
    check(out, ref)
    for rank in (0, 3, 7):  # `recompute`: a band also runs the lattice row above it; spot-check three bands
        sh = ShardedApply(coords, lambda idx: k[idx], n, h, w, rank, world, 0, None, seam="recompute")
        b = sh.band
        sh.upload_rows(image[b.image_row0 : b.image_row0 + b.image_rows])
        sh.step()
        got = sh.owned_rows().astype(np.float64)
        rows = ref[b.out_row0 : b.out_row0 + b.own_rows]
        assert np.abs(got - rows).max() <= TOL * np.abs(ref).max()


@pytest.mark.timeout(1500, method="thread")
def test_two_persistent_plans_on_two_streams():
    """What ShardedApply(overlap=True) asks of the device at eight bands of the 8192-wide configuration: two persistent launches of
    the 256-px plan in flight at once - the seam plan (65 patches: 72 workgroups) on its stream and the main plan (520 patches:
    8 summing + 240 patch workgroups, 8 CUs left free) on the other, 320 workgroups for 256 CUs - with a third stream running K4
    beside them (the stand-in for RCCL's send/recv kernels).  200 steps, two different frames; every step the same bits, the first
    against the oracle.  (Forward progress of concurrent persistent launches: rpsf.hip, launch_patches.)"""
    from regularizepsf_amd import _native
    from regularizepsf_amd.sharding import ShardedApply

    h = w = 8192
    n, world, rank = 256, 8, 3
    coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((h, w), n)]
    rng = np.random.default_rng(11)
    k_of = {}

    def kernel_for(index):
        for i in index:
            if i not in k_of:
                k_of[i] = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)
        return np.stack([k_of[i] for i in index])

    sh = ShardedApply(coords, kernel_for, n, h, w, rank, world, 0, None, overlap=True)
    b = sh.band
    assert sh.overlap and sh.seam_plan is not None and len(b.patch_index) == 520 and b.send_rows == 128
    sh.plan.set_reserved_cus(8)  # as with a communicator attached
    side = _native.Plan(16, [(0, 0)])  # a third stream
    count = 8 * 256 * 4
    d_a = _native.DeviceBuffer(count * 4).upload(np.zeros(count, np.float32))
    d_b = _native.DeviceBuffer(count * 4).upload(np.ones(count, np.float32))
    frames = [(rng.standard_normal((b.image_rows, w)) * 10 + 50 + 20 * f).astype(np.float32) for f in range(2)]
    first = {}
    steps = 0
    for f in (0, 1, 0, 1):
        sh.upload_rows(frames[f])
        for _ in range(2):
            for _ in range(25):
                sh.step()
                _native.add_rows(d_a.ptr, d_b.ptr, count, 0, side.stream)
                steps += 1
            got = (sh.owned_rows(), sh.spill_rows())
            if f not in first:
                first[f] = got
            assert np.array_equal(got[0], first[f][0]) and np.array_equal(got[1], first[f][1]), (f, steps)
    assert steps == 200
    _native.check(_native.lib().rpsf_device_synchronize(0))
    assert np.array_equal(d_a.download((count,)), np.full(count, float(steps), np.float32))
    # the band's patches alone (nothing arrives from the band above in this test) against the oracle, frame 0
    local = [(coords[i][0] - b.image_row0, coords[i][1]) for i in b.patch_index]
    ref = orc.apply_transfer(frames[0], local, kernel_for(b.patch_index), workers=-1)
    r0 = b.out_row0 - b.image_row0
    scale = np.abs(ref).max()
    assert np.abs(first[0][0] - ref[r0 : r0 + b.own_rows]).max() <= TOL * scale
    assert np.abs(first[0][1] - ref[r0 + b.own_rows : r0 + b.own_rows + b.send_rows]).max() <= TOL * scale


def test_config5_2048_frames_sharing_one_kernel():
    """BASELINE.json configs[4] at its real frame size: a batch of 2048^2 starfields, 128-px patches, one shared transfer
    kernel, corrected in one launch on the device; first and last frame against the CPU oracle."""
    from regularizepsf_amd import _native

    n, size, frames = 128, 2048, 8
    coords, k = orc.synthetic_transfer(size, size, n, alpha=3.0, epsilon=0.1)
    images = np.stack([orc.starfield(size, size, 100 + i) for i in range(frames)])
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    d_in = _native.DeviceBuffer(images.nbytes).upload(images)
    d_out = _native.DeviceBuffer(images.nbytes)
    plan.apply_batch_device(d_in.ptr, d_out.ptr, frames, size * size, size * size, _native.Geometry.whole(size, size, 1))
    plan.synchronize()
    out = d_out.download((frames, size, size))
    for f in (0, frames - 1):
        check(out[f].astype(np.float64), orc.apply_transfer(images[f], coords, k, workers=-1))
    assert np.array_equal(out[3], plan.apply(images[3], 1))  # the batch is the frame-by-frame loop, bit for bit


@pytest.mark.timeout(300, method="thread")
@pytest.mark.parametrize(("n", "size"), [(128, 768), (256, 1024)])
def test_batches_and_single_applies_in_any_order_on_one_plan(n, size):
    """The fused launches count finished patches on per-(frame, tile) counters that are never reset inside a run of equal
    launches (a tile is complete at epoch x contributors).  Launches of different frame counts on ONE plan - batch(8) ->
    apply -> batch(8), batch(8) -> batch(3) -> batch(8) - must restart the count (a launch advances only the counters of
    its own frames); every result bit-identical to the first batch."""
    from regularizepsf_amd import _native

    frames = 8
    rng = np.random.default_rng(n)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((size, size), n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    images = (rng.standard_normal((frames, size, size)) * 10 + 50).astype(np.float32)
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    d_in = _native.DeviceBuffer(images.nbytes).upload(images)
    d_out = _native.DeviceBuffer(images.nbytes)
    geom = _native.Geometry.whole(size, size, 1)

    def batch(count):
        d_out.upload(np.zeros_like(images))
        plan.apply_batch_device(d_in.ptr, d_out.ptr, count, size * size, size * size, geom)
        plan.synchronize()
        return d_out.download((frames, size, size))[:count]

    first = batch(frames)
    check(first[frames - 1].astype(np.float64), orc.apply_transfer(images[frames - 1], coords, k))
    for count in (1, frames, 3, frames, 1, 1, 5, frames):
        if count == 1:
            assert np.array_equal(plan.apply(images[2], 1), first[2])
        else:
            assert np.array_equal(batch(count), first[:count]), count


def test_seam_add_kernel_adds():
    """K4 (add_rows_kernel) on non-empty buffers, odd lengths included (vector body + scalar tail)."""
    from regularizepsf_amd import _native

    rng = np.random.default_rng(4)
    for count in (1, 3, 4, 1023, 128 * 8192 + 5):
        a = rng.standard_normal(count).astype(np.float32)
        b = rng.standard_normal(count).astype(np.float32)
        da = _native.DeviceBuffer(a.nbytes).upload(a)
        db = _native.DeviceBuffer(b.nbytes).upload(b)
        _native.add_rows(da.ptr, db.ptr, count)
        _native.check(_native.lib().rpsf_device_synchronize(0))
        assert np.array_equal(da.download((count,)), a + b)
        assert np.array_equal(db.download((count,)), b)


def test_reference_written_h5_file_runs_on_the_gpu():
    """a transform file written by the reference's own save() loads straight into the HIP path."""
    t = rp.ArrayPSFTransform.load(GOLDEN / "h5_transform_c64.h5")
    n = t.psf_shape[0]
    assert t._transfer_kernel.values.dtype == np.complex64 and len(t) > 0
    rng = np.random.default_rng(8)
    rows = max(r for r, _ in t.coordinates) + n // 2
    cols = max(c for _, c in t.coordinates) + n // 2
    image = (rng.standard_normal((rows, cols)) * 10 + 50).astype(np.float32)
    out = t.apply(image)
    check(out, orc.apply_transfer(image, t.coordinates, t._transfer_kernel.values))


def test_in_place_edits_of_the_kernel_are_loud_or_seen():
    """The reference reads `values` at every apply (transform.py:164).  Here a device copy is taken; while it exists the array is read-only,
    so an edit that bypasses IndexedCube.__setitem__ raises instead of leaving a stale copy behind - however small it is and wherever
    it lands (the fingerprint of rounds 2-5 sampled one 64-byte line per patch and missed the rest) - and every sanctioned way of
    editing gives the reference's answer."""
    fx, coords, k = load_apply_case("n32_sym")
    image = fx["image"]
    k = k.copy()
    alias = k[:]  # a view taken before the upload stays writeable: edits through it are left to the fingerprint
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    t.apply(image)
    assert not k.flags.writeable
    with pytest.raises(ValueError):
        k[3, 5, 7] = 0  # one element, off the fingerprint's line
    with pytest.raises(ValueError):
        t._transfer_kernel.values[3, 5, 7] = 0
    with pytest.raises(ValueError):
        k *= 2
    with t.edit() as kk:  # any edit, however small
        kk[3, 5, 7] = 1.5 - 2j
    check(t.apply(image), orc.apply_transfer(image, coords, k.copy()))
    assert not k.flags.writeable
    t.invalidate()  # drop the device copy: the array is the caller's again
    assert k.flags.writeable
    k[4, 1, 1] *= 3
    check(t.apply(image), orc.apply_transfer(image, coords, k.copy()))
    t._transfer_kernel[coords[1]] = np.zeros(k.shape[1:], np.complex64)  # __setitem__ is counted
    assert complex(k[1, 0, 0]) == 0
    check(t.apply(image), orc.apply_transfer(image, coords, k.copy()))
    before = t.apply(image)
    alias *= 2  # through memory that was never frozen: the fingerprint's job
    check(t.apply(image), 2 * before, tol=2e-6)


@pytest.mark.parametrize(("n", "shape"), [(16, (200, 264)), (32, (512, 512)), (32, (300, 500)), (64, (512, 384)), (64, (700, 1000))])
def test_sweep_kernel_against_the_other_overlap_adds_and_any_cut(n, shape):
    """The third-generation kernel (N <= 64: a workgroup owns a region of output pixels and adds the four contributions of a pixel in LDS,
    in a fixed order) against the oracle, against the colour planes and the float atomics, bit-reproducible, and bit-identical however the
    lattice is cut into regions (rpsf_plan_set_sweep_regions)."""
    from regularizepsf_amd import _native

    coords, k, images = _random_case(n, shape, 5 * n + shape[0], 1)
    image = images[0]
    pad = _native.PAD_MODES["symmetric"]
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    info = plan.sweep_info()
    assert info["regions"] > 0 and info["patch_slots"] >= len(coords)
    base = plan.apply(image, pad)  # automatic = the sweep kernel
    check(base.astype(np.float64), orc.apply_transfer(image, coords, k))
    for _ in range(3):
        assert np.array_equal(plan.apply(image, pad), base)
    for target in (1, 7, 64, 1000, 100000):
        plan.set_sweep_regions(target)
        assert np.array_equal(plan.apply(image, pad), base), (target, plan.sweep_info())
    scale = np.abs(base).max()
    for mode in ("planes", "atomic"):
        other = _native.Plan(n, coords)
        other.set_transfer(k)
        other.set_overlap_mode(mode)
        assert np.abs(other.apply(image, pad).astype(np.float64) - base).max() <= 2e-6 * scale, mode
    plan.set_overlap_mode("sweep")
    assert np.array_equal(plan.apply(image, pad), base)


def test_sweep_kernel_needs_a_complete_lattice():
    """Corner lists that are not a complete lattice keep the earlier paths (atomics / planes); asking for the sweep kernel says why not."""
    from regularizepsf_amd import _native

    coords, k, images = _random_case(32, (200, 200), 9, 1)
    sparse = coords[::3]
    plan = _native.Plan(32, sparse)
    plan.set_transfer(k[::3])
    assert plan.sweep_info()["regions"] == 0
    with pytest.raises(_native.NativeError):
        plan.set_overlap_mode("sweep")
    check(plan.apply(images[0], _native.PAD_MODES["symmetric"]).astype(np.float64), orc.apply_transfer(images[0], sparse, k[::3]))


@pytest.mark.parametrize("n", [16, 64])
def test_sweep_kernel_batches_and_row_windows(n):
    """Frames that share the transfer kernel ride along grid.y of the one launch: every frame equals its own single apply bit for bit."""
    from regularizepsf_amd import _native

    shape, frames = (256, 384), 5
    coords, k, images = _random_case(n, shape, 21 + n, frames)
    h, w = shape
    pad = _native.PAD_MODES["reflect"]
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    stack = np.ascontiguousarray(images, np.float32)
    d_in = _native.DeviceBuffer(stack.nbytes).upload(stack)
    d_out = _native.DeviceBuffer(stack.nbytes)
    plan.apply_batch_device(d_in.ptr, d_out.ptr, frames, h * w, h * w, _native.Geometry.whole(h, w, pad))
    plan.synchronize()
    got = d_out.download((frames, h, w))
    for f in range(frames):
        single = plan.apply(images[f], pad)
        assert np.array_equal(got[f], single), f
        check(single.astype(np.float64), orc.apply_transfer(images[f], coords, k, pad_mode="reflect"))


def test_one_transform_from_two_threads():
    """ArrayPSFTransform.apply may be called from several threads on ONE transform (the reference is plain NumPy and
    allows it); the device plan is not re-entrant, so the calls take turns."""
    import threading

    coords, k, images = _random_case(64, (300, 260), 77, 2)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    expect = [t.apply(images[i]) for i in range(2)]
    errors = []

    def work(i):
        for _ in range(6):
            if not np.array_equal(t.apply(images[i]), expect[i]):
                errors.append(i)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors


def test_fused_plane_sum_epochs_origins_and_fallback():
    """The 256-px plan sums its colour planes inside the patch launch when the geometry allows it (per-tile counters that are
    never reset: a tile is complete at epoch x contributors).  Same plan, many applies: an aligned frame (fused),
    a frame whose width is not a multiple of 32 (falls back to the separate sum kernel), a host-padded frame (shifted
    origin, fused again) - every result against the oracle, and the fused result bit-identical to the unfused one."""
    from regularizepsf_amd import _native

    n = 256
    rng = np.random.default_rng(21)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering((768, 1024), n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    big = (rng.standard_normal((768, 1024)) * 10 + 30).astype(np.float32)
    odd = big[:700, :1000].copy()       # width 1000: not a multiple of 32 -> unfused
    refs = {"big": orc.apply_transfer(big, coords, k), "odd": orc.apply_transfer(odd, coords, k)}
    first = t.apply(big)
    check(first, refs["big"])
    check(t.apply(odd), refs["odd"])
    for _ in range(3):
        assert np.array_equal(t.apply(big), first)
    check(t.apply(big, pad_mode="mean"), orc.apply_transfer(big, coords, k, pad_mode="mean"))  # host-padded: origin (2N, 2N)
    assert np.array_equal(t.apply(big), first)
    # fused == unfused, bit for bit (same colour order)
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    a = plan.apply(big, 1)
    plan2 = _native.Plan(n, coords)
    plan2.set_transfer(k)
    plan2.set_option("fuse", 0)
    b = plan2.apply(big, 1)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(768, 1024), (2048, 2304), (512, 512)])
def test_persistent_patch_workgroups_bit_identical(shape):
    """The fused launch of the 256-px plan keeps its patch workgroups resident: each draws the next slot of its XCD's chunk
    from a queue and jumps back to the kernel's first instruction (the queues and the tile counters are never reset, the
    previous patch is counted on its tiles from inside the next pass).  Repeated applies of one plan, more patches than the
    chip holds workgroups (2048 x 2304: 323 > 256 - 8), fewer patches than XCDs have slots (512^2: 25) - against the oracle
    and bit-identical to the one-patch-per-workgroup launch (RPSF_OPT_PERSIST = 0)."""
    import os

    from regularizepsf_amd import _native

    n = 256
    rng = np.random.default_rng(77)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = (rng.standard_normal(shape) * 10 + 30).astype(np.float32)
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    outs = [plan.apply(image, 1) for _ in range(4)]
    check(outs[0].astype(np.float64), orc.apply_transfer(image, coords, k))
    assert all(np.array_equal(o, outs[0]) for o in outs[1:])
    plan.set_reserved_cus(8)  # what the sharded apply asks for when RCCL kernels have to run beside the launch
    assert np.array_equal(plan.apply(image, 1), outs[0])
    plan.set_reserved_cus(0)
    plan.set_stagger(0)
    assert np.array_equal(plan.apply(image, 1), outs[0])
    plain = _native.Plan(n, coords)
    plain.set_transfer(k)
    plain.set_option("persist", 0)
    assert np.array_equal(plain.apply(image, 1), outs[0])


def test_image_prefetch_changes_nothing_but_the_timing():
    """rpsf_plan_set_image_prefetch (opt-in, for streams of new frames): the head summing workgroups of the persistent launch touch the
    image ahead of the gathers as a side job of their summing loop.  Same bits with and without, repeated applies, two frames in turn;
    ignored by plans it does not apply to (128-pixel patches)."""
    from regularizepsf_amd import _native

    rng = np.random.default_rng(8)
    for n, shape in ((256, (2048, 2304)), (256, (768, 1024)), (128, (768, 1024))):
        coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
        k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
        images = [(rng.standard_normal(shape) * 10 + 30 + 7 * f).astype(np.float32) for f in range(2)]
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        plain = [plan.apply(im, 1) for im in images]
        plan.set_image_prefetch(True)
        for _ in range(3):
            for im, ref in zip(images, plain):
                assert np.array_equal(plan.apply(im, 1), ref)
        plan.set_image_prefetch(False)
        assert np.array_equal(plan.apply(images[0], 1), plain[0])
        check(plain[1].astype(np.float64), orc.apply_transfer(images[1], coords, k, workers=-1))


@pytest.mark.parametrize("shape,frames", [((1024, 1280), 5), ((4096, 4096), 3)])
def test_batches_of_the_256_pixel_plan_bit_identical_to_the_loop(shape, frames):
    """Batches of the persistent 256-pixel plan: the frames of a patch slot side by side (small frames) or frame after frame in one
    launch (frames whose planes would not share the Infinity Cache) - either way the same bits as applying the frames one by one,
    run after run, and the oracle's result."""
    n = 256
    rng = np.random.default_rng(4)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    stack = (rng.standard_normal((frames, *shape)) * 10 + 30).astype(np.float32)
    out = t.apply_batch(stack)
    assert all(np.array_equal(out[i], t.apply(stack[i])) for i in range(frames))
    assert np.array_equal(out, t.apply_batch(stack))
    check(out[frames - 1], orc.apply_transfer(stack[frames - 1], coords, k, workers=-1))


def test_device_resident_psf_to_transform_chain():
    """ArrayPSF(device=0) leaves the spectra on the GPU, construct() builds and packs K there, apply() runs -
    nothing but the PSF samples and the image crosses PCIe.  The host copies appear only when somebody looks at them, and
    they are what the host route computes."""
    from regularizepsf_amd import _native

    n, shape = 64, (200, 260)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    src, tgt = make_psfs("coma", coords, n, *shape)
    s = rp.ArrayPSF(rp.IndexedCube(coords, src.astype(np.float32)), device=0)
    t = rp.ArrayPSF(rp.IndexedCube(coords, tgt.astype(np.float32)), device=0)
    assert s._fft_cube._loader is not None and s.sample_shape == (n, n) and len(s) == len(coords)
    image = orc.starfield(*shape, 9)
    sweep = []
    for alpha, eps in ((3.0, 0.1), (1.0, 0.05)):  # an alpha / epsilon sweep re-uses the resident spectra
        tr = rp.ArrayPSFTransform.construct(s, t, alpha, eps)
        assert tr._transfer_kernel._loader is not None and s._fft_cube._loader is not None  # still nothing downloaded
        out = tr.apply(image)
        assert tr._transfer_kernel._loader is not None  # apply() did not need K on the host either
        sweep.append((alpha, eps, tr, out))
    got_s, got_t = s.fft_evaluations, t.fft_evaluations  # fetched now
    assert got_s.dtype == np.complex64 and np.array_equal(got_s, _native.psf_fft(src.astype(np.float32)))
    ref_s = orc.psf_fft(src.astype(np.float32))
    assert np.abs(got_s - ref_s).max() <= 1e-5 * np.abs(ref_s).max()
    for alpha, eps, tr, out in sweep:
        plan_before = tr._plan
        k = tr._transfer_kernel.values  # fetched now: the same K the host route builds from the same spectra
        assert k.dtype == np.complex64 and np.array_equal(k, _native.build_transfer(got_s, got_t, alpha, eps), equal_nan=True)
        check(out, orc.apply_transfer(image, coords, k))
        assert np.array_equal(tr.apply(image), out) and tr._plan is plan_before  # looking at K did not cost a re-upload
        tr._transfer_kernel[coords[0]] = np.zeros((n, n), np.complex64)  # edits after the fetch are still noticed
        assert not np.array_equal(tr.apply(image), out)
    # The spectra have been fetched by now, so a caller may have edited them in place: construct reads them as they are
    # (transform.py:78-82) instead of the copy it left on the GPU.
    got_s *= 2
    edited = rp.ArrayPSFTransform.construct(s, t, 3.0, 0.1)
    assert np.array_equal(edited._transfer_kernel.values, _native.build_transfer(got_s, got_t, 3.0, 0.1), equal_nan=True)
    assert not np.array_equal(edited._transfer_kernel.values, sweep[0][2]._transfer_kernel.values)


def test_specialised_persistent_kernels_and_the_geometries_they_hand_back():
    """The persistent kernels are compiled for "fused colour planes, every 16-byte unit of a patch maps to four consecutive image columns or to the
    fill" (patch_body2's HOT instantiation: no pixel-by-pixel rim paths in the code); the launcher proves that per apply (hot_geometry, rpsf.hip) and
    hands everything else to the one-patch-per-workgroup kernel.  One plan per patch size, the same frame under every pad mode - 'constant',
    'symmetric', 'wrap' run the persistent kernel, 'reflect' and 'edge' tear units apart and must not - and through image / output views whose rows
    are unaligned (a one-float offset, an odd stride): all against the oracle, and the aligned result bit-identical to the one-patch-per-workgroup
    launch of the same plan."""
    import os

    from regularizepsf_amd import _native

    for n, shape in ((256, (768, 1024)), (128, (512, 640))):
        rng = np.random.default_rng(400 + n)
        coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
        k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
        image = (rng.standard_normal(shape) * 10 + 30).astype(np.float32)
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        outs = {}
        for mode in ("symmetric", "constant", "wrap", "reflect", "edge"):
            outs[mode] = plan.apply(image, _native.PAD_MODES[mode])
            check(outs[mode].astype(np.float64), orc.apply_transfer(image, coords, k, pad_mode=mode))
        plain = _native.Plan(n, coords)
        plain.set_transfer(k)
        plain.set_option("persist", 0)
        for mode in ("symmetric", "wrap", "reflect"):
            assert np.array_equal(plain.apply(image, _native.PAD_MODES[mode]), outs[mode]), mode
        # device-resident views: the image one float into its buffer (rows no longer 16-byte aligned) with an odd stride
        h, w = shape
        ld = w + 3
        host = np.zeros(h * ld + 1, np.float32)
        host[1:].reshape(h, ld)[:, :w] = image
        d_img = _native.DeviceBuffer(host.nbytes).upload(host)
        d_out = _native.DeviceBuffer(h * w * 4)
        geom = _native.Geometry(h, w, _native.PAD_MODES["symmetric"], 0.0, 0, 0, 0, h, ld, 0, h, w)
        plan.apply_device(d_img.at(4), d_out.ptr, geom)
        plan.synchronize()
        assert np.array_equal(d_out.download((h, w)), outs["symmetric"])


def test_fused_sum_of_tiles_the_image_clips_to_an_odd_width():
    """The tile sums inside the persistent launch index a tile's 16-byte groups by shift and mask; a last tile column that the image clips to a width
    that is no power of two (96 pixels = 24 groups at N = 256) goes pixel by pixel instead.  Width 608 = 4 x 128 + 96, fused (a multiple of 32)."""
    from regularizepsf_amd import _native

    n, shape = 256, (512, 608)
    rng = np.random.default_rng(608)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = (rng.standard_normal(shape) * 10 + 30).astype(np.float32)
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    out = plan.apply(image, _native.PAD_MODES["symmetric"])
    check(out.astype(np.float64), orc.apply_transfer(image, coords, k))
    assert np.array_equal(plan.apply(image, _native.PAD_MODES["symmetric"]), out)


def test_every_form_of_the_128_pixel_persistent_kernel():
    """The 128-pixel persistent kernel exists three times: with streaming loads of the pair words of K (patch_kernel2_128p), with plain ones
    (patch_kernel2_128pc, chosen at plan creation when K fits the Infinity Cache beside the planes; RPSF_OPT_K_CACHED overrides) and with streaming
    plane stores on top (patch_kernel2_128pcs, chosen per launch for large batches; RPSF_OPT_PLANE_NT overrides).  Same frame, same K, every form,
    single frame and a batch of three that share K: against the oracle, and bit-identical to each other."""
    import os

    from regularizepsf_amd import _native

    n, shape = 128, (640, 768)
    rng = np.random.default_rng(77)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    frames = (rng.standard_normal((3, *shape)) * 10 + 30).astype(np.float32)
    outs = {}
    for form in ("0", "1"):
        plan = _native.Plan(n, coords)
        plan.set_option("k_cached", int(form))
        plan.set_transfer(k)
        single = plan.apply(frames[0], _native.PAD_MODES["symmetric"])
        check(single.astype(np.float64), orc.apply_transfer(frames[0], coords, k, pad_mode="symmetric"))
        batch = plan.apply_batch(frames, _native.PAD_MODES["symmetric"])
        assert np.array_equal(batch[0], single)
        check(batch[2].astype(np.float64), orc.apply_transfer(frames[2], coords, k, pad_mode="symmetric"))
        outs[form] = (single, batch)
    assert np.array_equal(outs["0"][0], outs["1"][0]) and np.array_equal(outs["0"][1], outs["1"][1])
    # the third form (patch_kernel2_128pcs: streaming plane stores, taken by large batches) on the same batch
    plan.set_option("plane_nt", 1)
    streamed = plan.apply_batch(frames, _native.PAD_MODES["symmetric"])
    plan.set_option("plane_nt", -1)
    assert np.array_equal(streamed, outs["1"][1])


@pytest.mark.timeout(900, method="thread")
def test_two_persistent_128_pixel_batches_on_two_streams():
    """Forward progress with the 128-pixel plan's largest head: from 4096 patch-frames on a persistent launch carries 160 head summing
    workgroups (sum_first_for, rpsf.hip) that hold CU slots without progress of their own.  Two such launches (two plans, two streams:
    4 x 2048^2 frames each, 4356 patch-frames) in flight at once, 40 rounds: every round the same bits, frame 0 against the oracle."""
    from regularizepsf_amd import _native

    h = w = 2048
    n, frames = 128, 4
    coords, k = orc.synthetic_transfer(h, w, n, alpha=3.0, epsilon=0.1)
    geom = _native.Geometry.whole(h, w, _native.PAD_MODES["symmetric"])
    stacks = [np.stack([orc.starfield(h, w, 40 + 10 * j + f) for f in range(frames)]) for j in range(2)]
    plans, bufs = [], []
    for j in range(2):
        plan = _native.Plan(n, coords)
        plan.set_transfer(k)
        plans.append(plan)
        bufs.append((_native.DeviceBuffer(stacks[j].nbytes).upload(stacks[j]), _native.DeviceBuffer(stacks[j].nbytes)))
    first = [None, None]
    for _ in range(4):
        for _ in range(10):
            for j in range(2):
                plans[j].apply_batch_device(bufs[j][0].ptr, bufs[j][1].ptr, frames, h * w, h * w, geom)
        plans[0].synchronize()
        for j in range(2):
            got = bufs[j][1].download((frames, h, w))
            if first[j] is None:
                first[j] = got
            assert np.array_equal(got, first[j]), j
    ref = orc.apply_transfer(stacks[1][0], coords, k, workers=-1)
    assert np.abs(first[1][0] - ref).max() <= TOL * np.abs(ref).max()


@pytest.mark.parametrize(("n", "shape"), [(16, (64, 80)), (32, (96, 128)), (64, (192, 256)), (128, (384, 256)), (256, (512, 768)), (20, (60, 50))])
def test_construct_packs_k_straight_from_the_spectra(n, shape):
    """`construct` on device-resident spectra evaluates transform.py:78-82 where the packer reads K (one pass, the full K is never
    materialised): bit-identical to K2 followed by the pack kernel - in the packed stream (same apply, bit for bit) and in the values
    a caller sees when it looks at the transform's kernel."""
    from regularizepsf_amd import _native

    h, w = shape
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    src, tgt = make_psfs("coma", coords, n, h, w)
    # (float32 spectra of smooth PSFs are exactly zero in the (N/2, N/2) bin - the alternating sum cancels - and 0 / 0 is NaN there, in the
    # reference's float32 arithmetic as well: test_construct_degenerate_bins_follow_reference covers that; here a small spike keeps K finite)
    src[:, 0, 0] += 2e-3
    tgt[:, 0, 0] += 1e-3
    image = orc.starfield(h, w, seed=n)
    pad = _native.PAD_MODES["symmetric"]
    per = len(coords) * n * n
    if n in _native.SUPPORTED_PATCH_SIZES:
        s_dev = _native.psf_fft_device(src.astype(np.float32))
        t_dev = _native.psf_fft_device(tgt.astype(np.float32))
    else:  # (no spectrum kernel for this size: the spectra come from the host)
        s_dev = _native.DeviceBuffer(per * 8).upload(orc.psf_fft(src).astype(np.complex64))
        t_dev = _native.DeviceBuffer(per * 8).upload(orc.psf_fft(tgt).astype(np.complex64))
    fused = _native.Plan(n, coords)
    fused.set_transfer_spectra_device(s_dev.ptr, t_dev.ptr, 3.0, 0.1)
    k_dev = _native.DeviceBuffer(per * 8)
    _native.build_transfer_device(s_dev.ptr, t_dev.ptr, k_dev.ptr, per, False, 3.0, 0.1)
    two_pass = _native.Plan(n, coords)
    two_pass.set_transfer_device(k_dev.ptr)
    a, b = fused.apply(image, pad), two_pass.apply(image, pad)
    if n in _native.SUPPORTED_PATCH_SIZES:
        assert np.array_equal(a, b)
    else:  # float atomics in the fallback's overlap-add
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    k = k_dev.download((len(coords), n, n), np.complex64)
    assert np.isfinite(k).all()
    check(a.astype(np.float64), orc.apply_transfer(image, coords, k))
    if n in _native.SUPPORTED_PATCH_SIZES:  # the class API takes the same route and hands out the same K when asked
        ps = rp.ArrayPSF(rp.IndexedCube(coords, src.astype(np.float32)), device=0)
        pt = rp.ArrayPSF(rp.IndexedCube(coords, tgt.astype(np.float32)), device=0)
        t = rp.ArrayPSFTransform.construct(ps, pt, 3.0, 0.1)
        assert np.array_equal(t.apply(image), a.astype(np.float64))
        assert np.array_equal(t._transfer_kernel.values, k)
        assert np.array_equal(t.apply(image), a.astype(np.float64))  # looking at K did not change (or re-upload) anything
