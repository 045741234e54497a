"""Host-side logic of the drop-in API (no GPU): mirrors what the reference's own tests pin for
IndexedCube / calculate_covering / ArrayPSF validation / error conventions
(tests/test_util.py, tests/test_psf.py:33-89, tests/test_transform.py:76-82,100-109 upstream)."""

import numpy as np
import pytest
import scipy.fft
from hypothesis import given, settings
from hypothesis import strategies as st

import regularizepsf_amd as rp
from tests.helpers import GOLDEN


def gaussian(size, fwhm=3.0):
    x = np.arange(size, dtype=float)
    c = size // 2
    return np.exp(-4 * np.log(2) * ((x[None, :] - c) ** 2 + (x[:, None] - c) ** 2) / fwhm**2)


def covered_four_times(corners, shape, size):
    counts = np.zeros(shape)
    for r, c in corners:
        counts[max(0, r) : min(shape[0], r + size), max(0, c) : min(shape[1], c + size)] += 1
    return bool(np.all(counts == 4))


@pytest.mark.parametrize(("shape", "size"), [((5, 5), 1), ((5, 5), 2), ((15, 15), 3), ((15, 15), 4), ((100, 100), 11)])
def test_covering_is_fourfold(shape, size):
    assert covered_four_times(rp.calculate_covering(shape, size), shape, size)


@given(dim=st.integers(min_value=100, max_value=200), fraction=st.fractions(min_value=0.1, max_value=0.8))
@settings(max_examples=60, deadline=None)
def test_covering_is_fourfold_random(dim, fraction):
    size = int(np.ceil(dim * fraction))
    assert covered_four_times(rp.calculate_covering((dim, dim), size), (dim, dim), size)


def test_covering_order_matches_reference_golden():
    fx = np.load(GOLDEN / "covering.npz")
    for key in fx.files:
        _, shape, n = key.split("_")
        h, w = (int(v) for v in shape.split("x"))
        got = rp.calculate_covering((h, w), int(n))
        assert got.dtype == fx[key].dtype and np.array_equal(got, fx[key]), key


@pytest.mark.parametrize(("layers", "rows", "cols"), [(10, 10, 10), (15, 20, 25), (1, 15, 10), (1, 1, 1), (0, 1, 1), (0, 0, 0)])
def test_indexed_cube_behaviour(layers, rows, cols):
    data = np.zeros((layers, rows, cols))
    for i in range(layers):
        data[i] = i
    coords = [(i, i + 1) for i in range(layers)]
    cube = rp.IndexedCube(coords, data)
    assert cube.sample_shape == (rows, cols) and len(cube) == layers and cube.coordinates == coords
    assert cube.values is data  # no copy
    for coord in coords:
        assert np.all(cube[coord] == coord[0])
        with pytest.raises(rp.InvalidCoordinateError):
            _ = cube[(coord[1], coord[0])]
        with pytest.raises(rp.InvalidCoordinateError):
            cube[(coord[1], coord[0])] = np.zeros((rows, cols))
        with pytest.raises(rp.IncorrectShapeError):
            cube[coord] = np.zeros((rows + 1, cols + 1))
        cube[coord] = np.zeros((rows, cols))
    assert np.all(cube.values == 0)


def test_indexed_cube_errors_and_equality():
    with pytest.raises(TypeError):
        _ = rp.IndexedCube([(0, 0), (0, 1)], np.ones((2, 2, 2))) == np.zeros((2, 2, 2))
    with pytest.raises(rp.IncorrectShapeError):
        rp.IndexedCube([(0, 0), (0, 1), (5, 5)], np.ones((2, 2, 2)))
    with pytest.raises(rp.IncorrectShapeError):
        rp.IndexedCube([(0, 0), (0, 1)], np.ones((2, 2)))
    a = rp.IndexedCube([(0, 0)], np.ones((1, 2, 2)))
    assert a == rp.IndexedCube([(0, 0)], np.ones((1, 2, 2)) + 5e-7)  # atol 1e-6
    assert not (a == rp.IndexedCube([(0, 0)], np.ones((1, 2, 2)) + 1e-3))
    assert not (a == rp.IndexedCube([(1, 0)], np.ones((1, 2, 2))))


def test_arraypsf_validation_and_accessors():
    coords = [(0, 0), (1, 1), (2, 2)]
    g = gaussian(128)
    values = np.stack([g for _ in coords])
    psf = rp.ArrayPSF(rp.IndexedCube(coords, values))
    assert np.all(psf[(0, 0)] == g)
    assert np.all(psf.fft_at((0, 0)) == scipy.fft.fft2(g))  # bit exact, tests/test_psf.py:82-89 upstream
    assert psf.sample_shape == (128, 128) and len(psf) == 3 and psf.coordinates == coords
    assert psf.fft_evaluations.dtype == np.complex128
    assert rp.ArrayPSF(rp.IndexedCube(coords, values.astype(np.float32))).fft_evaluations.dtype == np.complex64
    with pytest.raises(TypeError):
        _ = psf == np.zeros((50, 50))
    with pytest.raises(rp.InvalidCoordinateError):
        rp.ArrayPSF(rp.IndexedCube(coords, values), rp.IndexedCube([(0, 0), (1, 1), (3, 3)], values))
    four = [(0, 0), (1, 1), (2, 2), (3, 3)]
    with pytest.raises(rp.IncorrectShapeError):
        rp.ArrayPSF(rp.IndexedCube(coords, values), rp.IndexedCube(four, np.stack([g for _ in four])))
    with pytest.raises(rp.IncorrectShapeError):
        rp.ArrayPSF(rp.IndexedCube(coords, values), rp.IndexedCube(coords, np.stack([gaussian(64) for _ in coords])))
    assert psf == rp.ArrayPSF(rp.IndexedCube(coords, values.copy()))


def test_construct_rejects_mismatched_coordinates_before_touching_the_gpu():
    src = rp.ArrayPSF(rp.IndexedCube([(0, 0), (1, 1), (2, 2)], np.zeros((3, 128, 128))))
    tgt = rp.ArrayPSF(rp.IndexedCube([(0, 0), (1, 1), (0.5, 0.5)], np.zeros((3, 128, 128))))
    with pytest.raises(rp.InvalidCoordinateError):
        rp.ArrayPSFTransform.construct(src, tgt, 3.0, 0.1)


def test_transform_shell():
    coords = [(0, 0), (16, 16)]
    k = np.ones((2, 32, 32), np.complex64)
    t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
    assert t.psf_shape == (32, 32) and t.coordinates == coords and len(t) == 2
    assert t == rp.ArrayPSFTransform(rp.IndexedCube(coords, k.copy()))
    with pytest.raises(TypeError):
        _ = t == np.zeros((50, 50))
    with pytest.raises(NotImplementedError):
        t.save("kernel.txt")
    with pytest.raises(NotImplementedError):
        rp.ArrayPSFTransform.load("kernel.txt")
    assert rp.ArrayPSFTransform.correct_image is rp.ArrayPSFTransform.apply


def test_apply_argument_errors_come_before_any_device_work():
    k = np.ones((1, 32, 32), np.complex64)
    with pytest.raises(ValueError):
        rp.ArrayPSFTransform(rp.IndexedCube([(0, 0)], k)).apply(np.zeros((2, 8, 8)))
    with pytest.raises(ValueError):  # non-square PSF: broadcast error in the reference
        rp.ArrayPSFTransform(rp.IndexedCube([(0, 0)], np.ones((1, 32, 16), np.complex64))).apply(np.zeros((64, 64)))
    with pytest.raises(NotImplementedError):  # outside 2..4096
        rp.ArrayPSFTransform(rp.IndexedCube([(0, 0)], np.ones((1, 1, 1), np.complex64))).apply(np.zeros((64, 64)))


def test_empty_transform_raises_like_the_reference():
    t = rp.ArrayPSFTransform(rp.IndexedCube([], np.zeros((0, 16, 16), np.complex64)))
    assert len(t) == 0
    with pytest.raises(ValueError):
        t.apply(np.zeros((8, 8)))


def test_deferred_cube_fetches_once_and_tells_its_owner():
    """IndexedCube._deferred (regularizepsf_amd extension: values that stay on the GPU until somebody looks at them): the loader runs once,
    on first access, and a one-shot hook fires right after the values have arrived - before the caller can edit them - which is how a
    transform keeps the stamp of its device-resident kernel in step (looking at K must not cost a re-upload)."""
    calls = []
    data = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)

    def loader():
        calls.append("load")
        return data.copy()

    cube = rp.IndexedCube._deferred([(0, 0), (3, 0)], (2, 3, 3), loader)
    cube._load_hook = lambda c: calls.append(("hook", c._loader is None, c._values_array is not None))
    assert cube.sample_shape == (3, 3) and len(cube) == 2 and cube.coordinates == [(0, 0), (3, 0)] and calls == []
    assert np.array_equal(cube[(3, 0)], data[1])
    assert calls == ["load", ("hook", True, True)]
    cube[(0, 0)] = np.zeros((3, 3), np.float32)
    assert np.array_equal(cube.values[0], np.zeros((3, 3))) and calls == ["load", ("hook", True, True)] and cube._edits == 1
    with pytest.raises(rp.exceptions.IncorrectShapeError):
        rp.IndexedCube._deferred([(0, 0)], (2, 3, 3), loader)
