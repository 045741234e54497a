"""Functional PSF models: the reference's API and checks (regularizepsf/psf.py:25-189, mirrored from its tests/test_psf.py:93-237)
on the host, and - on the GPU - the built-in device models rasterised by kernel K6 against the same formulas in NumPy."""

import numpy as np
import pytest

import regularizepsf_amd as rp
from regularizepsf_amd import InvalidFunctionError
from regularizepsf_amd.psf import SimpleFunctionalPSF, VariedFunctionalPSF, simple_functional_psf, varied_functional_psf


def test_simple_models():
    eqn = simple_functional_psf(lambda row, col: row + col)
    assert isinstance(eqn, SimpleFunctionalPSF) and eqn.parameters == set() and eqn(1, 2) == 3
    eqn = simple_functional_psf(lambda row, col, sigma=3, mu=4: row + col + sigma + mu)
    assert eqn.parameters == {"sigma", "mu"} and eqn(1, 2) == 10 and eqn(1, 2, mu=0) == 6
    for bad in (lambda: 1, lambda y, x: x + y, lambda x, sigma: x + sigma, lambda row, c: row):
        with pytest.raises(InvalidFunctionError):
            simple_functional_psf(bad)
    with pytest.raises(TypeError):
        simple_functional_psf(3)


def test_varied_models():
    base = simple_functional_psf(lambda row, col, sigma=5: row + col + sigma)
    my_psf = varied_functional_psf(base)(lambda row, col: {"sigma": 1})
    assert isinstance(my_psf, VariedFunctionalPSF) and my_psf.parameters == {"sigma"} and my_psf(0, 0) == 1
    two = simple_functional_psf(lambda row, col, sigma, mu: row + col)
    plain = simple_functional_psf(lambda row, col: row + col)
    for b, vary in ((two, lambda: {"sigma": 0.1}), (plain, lambda row, col, c: {"sigma": 0.1}), (plain, lambda c, col: {"sigma": 0.1}),
                    (plain, lambda row, c: {"sigma": 0.1}), (simple_functional_psf(lambda row, col, m: row), lambda row, col: {"n": 0, "m": 30})):
        with pytest.raises(InvalidFunctionError):
            varied_functional_psf(b)(vary)
    with pytest.raises(TypeError):
        varied_functional_psf()(lambda row, col: {"sigma": 0.2})
    with pytest.raises(TypeError):
        varied_functional_psf(None)
    with pytest.raises(TypeError):
        varied_functional_psf(lambda row, col: {"sigma": 0.1})  # used naked
    base_m = simple_functional_psf(lambda row, col, m: row + col)

    @varied_functional_psf(base_m)
    def drifting(row, col):
        return {"m": 30} if row == 0 and col == 0 else {"n": 100, "m": 30}

    with pytest.raises(InvalidFunctionError):
        drifting(10, 10)
    relaxed = varied_functional_psf(base_m)(check_at_call=False)(lambda row, col: {"m": 1})
    assert relaxed.validate_at_call is False and relaxed(2, 3) == 5


def test_evaluation_to_array_psf_follows_the_reference_grid():
    """as_array_psf samples on np.meshgrid(arange, arange): element [i, j] is the model at row = j, col = i."""
    functionalpsf = simple_functional_psf(lambda row, col, a=10: 100 * row + col + a)
    arraypsf = functionalpsf.as_array_psf([(0, 0), (1, 0)], 3)
    assert len(arraypsf) == 2 and arraypsf.sample_shape == (3, 3) and arraypsf.coordinates == [(0, 0), (1, 0)]
    rr, cc = np.meshgrid(np.arange(3), np.arange(3))
    assert np.array_equal(arraypsf[(1, 0)], 100 * rr + cc + 10) and arraypsf[(0, 0)][0, 2] == 210
    assert np.array_equal(functionalpsf.as_array_psf([(0, 0)], 3, a=0)[(0, 0)], 100 * rr + cc)
    base = simple_functional_psf(lambda row, col, sigma=5: row + col + sigma)
    varied = varied_functional_psf(base)(lambda row, col: {"sigma": row * col})
    arraypsf = varied.as_array_psf([(0, 0), (3, 4)], 3)
    assert arraypsf.coordinates == [(0, 0), (3, 4)] and np.array_equal(arraypsf[(3, 4)], rr + cc + 12)
    assert varied.simplify(3, 4)(1, 1) == 14


def test_built_in_models_on_the_host():
    """The built-in models are ordinary simple models on the host; pack() is the parameter layout of include/rpsf.h."""
    g = rp.elliptical_gaussian
    assert g.parameters == {"amplitude", "row0", "col0", "sigma_row", "sigma_col", "theta", "background"}
    assert g(2.0, 3.0, row0=2.0, col0=3.0, amplitude=4.0, background=1.0) == 5.0
    tilted = g(np.array([1.0]), np.array([0.0]), sigma_row=1.0, sigma_col=2.0, theta=np.pi / 2)  # rotated: row offsets see sigma_col
    assert np.allclose(tilted, np.exp(-0.5 / 4.0))
    assert rp.moffat(0.0, 2.0, alpha=2.0, beta=1.0) == 0.5
    table = g.pack([{"row0": 7, "sigma_row": 2.0}, {}])
    assert table.shape == (2, 8) and table[0].tolist() == [1.0, 7.0, 0.0, 2.0, 1.0, 0.0, 0.0, 0.0] and table[1][3] == 1.0
    with pytest.raises(InvalidFunctionError):
        g.pack([{"sigma": 1.0}])
    normal = g.as_array_psf([(0, 0), (8, 8)], 9, row0=4, col0=4, sigma_row=1.5, sigma_col=1.5, normalize=True)
    assert np.allclose(normal.values.sum(axis=(1, 2)), 1.0) and normal.values.dtype == np.float64


def _golden():
    import pathlib

    from tests.golden import make_functional_golden as gen

    return np.load(pathlib.Path(__file__).parent / "golden" / "functional.npz"), gen


def test_as_array_psf_equals_the_reference_bit_for_bit():
    """tests/golden/functional.npz holds what the reference's decorators and as_array_psf produce for the same callables:
    the host route here gives the same samples and spectra, bit for bit."""
    fx, gen = _golden()
    simple = simple_functional_psf(lambda row, col, a=10: 100 * row + col + a)
    assert np.array_equal(simple.as_array_psf([(0, 0), (1, 0)], 5, a=3).values, fx["simple_values"])
    coords = [tuple(int(v) for v in c) for c in fx["coords"]]
    for name, base, field in (("gaussian", rp.elliptical_gaussian, gen.gaussian_field), ("moffat", rp.moffat, gen.moffat_field)):
        arr = varied_functional_psf(base)(field).as_array_psf(coords, int(fx["size"]))
        assert arr.values.dtype == np.float64 and np.array_equal(arr.values, fx[f"{name}_values"])
        assert np.array_equal(arr.fft_evaluations, fx[f"{name}_fft"])


@pytest.mark.gpu
def test_device_rasterisation_against_the_reference_golden():
    fx, gen = _golden()
    coords = [tuple(int(v) for v in c) for c in fx["coords"]]
    for name, base, field in (("gaussian", rp.elliptical_gaussian, gen.gaussian_field), ("moffat", rp.moffat, gen.moffat_field)):
        dev = varied_functional_psf(base)(field).as_array_psf(coords, int(fx["size"]), device=0)
        want, want_fft = fx[f"{name}_values"], fx[f"{name}_fft"]
        assert np.abs(dev.values - want).max() <= 2e-7 * np.abs(want).max()
        assert np.abs(dev.fft_evaluations - want_fft).max() <= 1e-5 * np.abs(want_fft).max()


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["elliptical_gaussian", "moffat"])
def test_device_rasterisation_matches_the_host_formula(model):
    """K6: a varied built-in model rasterised on the GPU = the reference's host loop (one call per patch on the meshgrid),
    to float32 rounding; its spectra = scipy's of the same samples; normalisation to unit sum."""
    from oracle import regpsf_oracle as orc

    n, shape = 64, (300, 420)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]
    if model == "elliptical_gaussian":
        @varied_functional_psf(rp.elliptical_gaussian)
        def field(row, col):
            return {"amplitude": 2.0 + row / 300, "row0": n / 2 + 0.3 * col / 420, "col0": n / 2 - 0.2, "sigma_row": 1.5 + row / 400,
                    "sigma_col": 2.0 + col / 500, "theta": 0.4 + (row + col) / 700, "background": 1e-3}
    else:
        @varied_functional_psf(rp.moffat)
        def field(row, col):
            return {"amplitude": 1.0, "row0": n / 2 + row / 1000, "col0": n / 2 - col / 900, "alpha": 2.0 + col / 400, "beta": 2.5 + row / 600,
                    "background": 0.0}
    host = field.as_array_psf(coords, n)  # upstream route: float64 on the host
    for normalize in (False, True):
        dev = field.as_array_psf(coords, n, device=0, normalize=normalize)
        assert dev._values_cube._loader is not None and dev._fft_cube._loader is not None and len(dev) == len(coords)  # nothing fetched yet
        want = host.values / host.values.sum(axis=(1, 2), keepdims=True) if normalize else host.values
        got = dev.values
        assert got.dtype == np.float32 and got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-7 * np.abs(want).max()
        ref_fft = orc.psf_fft(got)
        assert np.abs(dev.fft_evaluations - ref_fft).max() <= 1e-5 * np.abs(ref_fft).max()
    last = want[-1]  # [i, j] = model(row = j, col = i): a transposed grid would not have passed
    assert np.abs(last - last.T).max() > 1e-3 * last.max()


@pytest.mark.gpu
def test_model_parameters_to_corrected_image_without_leaving_the_gpu():
    """SURVEY 8f-3 end to end: parameter tables -> K6 samples -> K3 spectra -> K2 transfer kernel -> K1 apply.  Only the
    parameters and the image cross PCIe; the result equals the oracle run on the downloaded K."""
    from oracle import regpsf_oracle as orc

    n, shape = 128, (384, 512)
    coords = [tuple(int(v) for v in c) for c in rp.calculate_covering(shape, n)]

    @varied_functional_psf(rp.elliptical_gaussian)
    def source(row, col):
        return {"amplitude": 1.0, "row0": n // 2, "col0": n // 2, "sigma_row": 1.0 + row / 1600, "sigma_col": 1.0 + col / 2100,
                "theta": 0.3, "background": 0.0}

    src = source.as_array_psf(coords, n, device=0, normalize=True)
    tgt = rp.elliptical_gaussian.as_array_psf(coords, n, device=0, normalize=True, row0=n // 2, col0=n // 2, sigma_row=1.3, sigma_col=1.3)
    # (narrow PSFs: in complex64 the reference formula turns bins where both spectra underflow into NaN)
    transform = rp.ArrayPSFTransform.construct(src, tgt, 3.0, 0.1)
    assert transform._transfer_kernel._loader is not None and src._fft_cube._loader is not None  # K and spectra still on the device
    image = orc.starfield(*shape, 12)
    out = transform.apply(image)
    k = transform._transfer_kernel.values
    assert np.isfinite(k).all() and k.dtype == np.complex64
    ref = orc.apply_transfer(image, coords, k)
    assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max()
    # ... and K is what the reference formula gives for the same spectra
    k_ref = orc.construct_transfer(src.fft_evaluations.astype(np.complex128), tgt.fft_evaluations.astype(np.complex128), 3.0, 0.1)
    assert np.abs(k - k_ref).max() <= 1e-5 * np.abs(k_ref).max()
