"""Pin the CPU oracle against fixtures produced by the real reference (tests/golden/make_golden.py)."""

import numpy as np
import pytest

from oracle import regpsf_oracle as orc
from tests.helpers import APPLY_CASES, GOLDEN, load_apply_case, load_c128_case, make_psfs


def test_covering_matches_reference_including_order():
    fx = np.load(GOLDEN / "covering.npz")
    for key in fx.files:
        _, shape, n = key.split("_")
        h, w = (int(v) for v in shape.split("x"))
        got = orc.calculate_covering((h, w), int(n))
        assert got.dtype == fx[key].dtype
        assert np.array_equal(got, fx[key]), key


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_construct_known_answers_and_random_cube_bit_exact(dt):
    fx = np.load(GOLDEN / "construct.npz")
    cdt = np.complex64 if dt == "float32" else np.complex128
    s_fft = orc.psf_fft(fx["rand_values_s"].astype(dt))
    t_fft = orc.psf_fft(fx["rand_values_t"].astype(dt))
    assert np.array_equal(s_fft, fx[f"rand_fft_{dt}"])
    assert s_fft.dtype == cdt
    with np.errstate(all="ignore"):
        for alpha in (0.5, 1.0, 2.0, 3.0):
            for eps in (0.1, 0.01):
                k = orc.construct_transfer(fx["table_s"].astype(cdt), fx["table_t"].astype(cdt), alpha, eps)
                assert np.array_equal(k, fx[f"table_{dt}_a{alpha}_e{eps}"], equal_nan=True)
                k = orc.construct_transfer(s_fft, t_fft, alpha, eps)
                assert k.dtype == cdt
                assert np.array_equal(k, fx[f"rand_{dt}_a{alpha}_e{eps}"], equal_nan=True)


def test_construct_degenerate_bins_follow_the_reference_table():
    """SURVEY.md 8a-4 known answers (alpha, eps=0.1)."""
    fx = np.load(GOLDEN / "construct.npz")
    k3 = fx["table_float64_a3.0_e0.1"].ravel()
    assert np.isnan(k3[0])  # S=T=0
    assert k3[1] == 0  # S=0, T!=0, alpha>=1
    assert k3[2] == 0  # S!=0, T=0
    assert np.isclose(k3[4], 0.9999, atol=1e-6)  # tiny but equal, complex128
    assert np.isnan(fx["table_float32_a3.0_e0.1"].ravel()[4])  # underflows in complex64
    assert np.isclose(k3[5], 0.99990001)  # S == T
    k2 = fx["table_float64_a2.0_e0.1"].ravel()
    assert np.isclose(k2[3], -0.1994359100623313 + 1.3960513704363187j)
    assert np.isnan(fx["table_float64_a0.5_e0.1"].ravel()[1])  # S=0, alpha<1


@pytest.mark.parametrize("case", [c[0] for c in APPLY_CASES])
def test_apply_bit_identical_to_reference(case):
    fx, coords, k = load_apply_case(case)
    out = orc.apply_transfer(fx["image"], coords, k, pad_mode=str(fx["pad_mode"]))
    assert out.dtype == np.float64
    assert np.array_equal(out, fx["expected"])


def test_apply_identity_error_is_the_analytic_value():
    """The upstream pin (tests/test_transform.py:29-49): error = 5 * (1 - 1/(1 + 0.1**4))."""
    fx, coords, k = load_apply_case("n256_identity")
    out = orc.apply_transfer(fx["image"], coords, k)
    assert np.allclose(fx["image"], out, atol=1e-3)
    assert abs(np.abs(out - fx["image"]).max() - 5 * (1 - 1 / (1 + 0.1**4))) < 1e-6


def test_apply_saturation_matches_reference():
    fx = np.load(GOLDEN / "apply_saturation.npz")
    coords = [tuple(int(v) for v in t) for t in fx["coords"]]
    src, _ = make_psfs("identity", coords, 64, 192, 192)
    s_fft = orc.psf_fft(src)
    k = orc.construct_transfer(s_fft, s_fft, 3.0, 0.1)
    for dil, nbw in ((1, 7), (2, 5), (0, 7)):
        with np.errstate(all="ignore"):
            out = orc.apply_transfer(fx["image"], coords, k, saturation_threshold=10,
                                     saturation_dilation=dil, neighborhood_width=nbw)
        assert np.array_equal(out, fx[f"expected_d{dil}_w{nbw}"], equal_nan=True)
        assert out[80, 80] == 100


def test_apply_with_complex128_kernel_bit_identical_to_reference():
    """float64 PSFs -> complex128 K, multiplied in complex128 (transform.py:164): the reference's own precision end to end."""
    fx, coords, k = load_c128_case()
    out = orc.apply_transfer(fx["image"], coords, k)
    assert np.array_equal(out, fx["expected"])


def _config1():
    """BASELINE.json configs[0]: 512 x 512 starfield (seed 1), 32-px patches, constant Gaussian PSF 1.8 -> 1.5, alpha 3, eps 0.1."""
    fx = np.load(GOLDEN / "config1_512_n32.npz")
    h, w, n = (int(v) for v in fx["meta"])
    coords, k = orc.synthetic_transfer(h, w, n, alpha=float(fx["alpha"]), epsilon=float(fx["eps"]), kind="gauss")
    image = orc.starfield(h, w, int(fx["seed"]))
    return fx, coords, k, image


def test_config1_512_n32_full_size_bit_identical_to_reference():
    """The plumbing configuration at its real size (1089 patches): the oracle reproduces the reference's K and output to the bit
    (SHA-256 recorded from the reference by tests/golden/make_golden.py), it is linear in the image, and with source == target
    the transform is the identity within the reference test's own bound
    (tests/test_transform.py:29-49: atol 1e-3 on a frame of 5s)."""
    from tests.helpers import sha

    fx, coords, k, image = _config1()
    assert len(coords) == 1089 and sha(image) == str(fx["image_sha256"]) and sha(k) == str(fx["k_sha256"])
    out = orc.apply_transfer(image, coords, k)
    assert out.dtype == np.float64 and out.shape == image.shape
    assert sha(out) == str(fx["out_sha256"])
    assert np.array_equal(out[::8, ::8], fx["sample"])
    assert np.array_equal(orc.apply_transfer(2 * image, coords, k), 2 * out)  # linear, and exactly so for a power of two
    # identity-style bound, as upstream
    s_fft = orc.psf_fft(np.broadcast_to(orc.gaussian_psf(32, 1.8), (len(coords), 32, 32)))
    ident = orc.construct_transfer(s_fft, s_fft, 3.0, 0.1).astype(np.complex64)
    flat = np.zeros((512, 512), np.float32)
    flat[120:250, 50:100] = 5
    assert np.allclose(orc.apply_transfer(flat, coords, ident), flat, atol=1e-3)
