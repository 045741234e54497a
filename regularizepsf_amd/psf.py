"""Array PSF model (API of regularizepsf/psf.py:192-260,402-416)."""

from __future__ import annotations

import pathlib

import numpy as np
import scipy.fft

from regularizepsf_amd.exceptions import IncorrectShapeError, InvalidCoordinateError
from regularizepsf_amd.util import IndexedCube


class ArrayPSF:
    """A spatially varying PSF sampled as one array per patch, plus the 2-D spectrum of each sample.

    Same constructor and accessors as the reference.  The spectrum is the plain un-shifted
    ``fft2`` over the last two axes (psf.py:216-219); by default it is computed on the host with
    the reference's own backend (``scipy.fft``), which keeps ``fft_at`` bit-identical to the
    reference (its tests/test_psf.py:82-89 pins exactly that).  ``device=<gpu index>`` is an opt-in
    addition that computes float32 spectra with the HIP kernel instead (complex64 result) and **leaves them on the
    GPU**: ``ArrayPSFTransform.construct`` then builds, packs and applies K without a spectrum or K ever crossing
    PCIe; ``fft_evaluations`` / ``fft_at`` download them the first time they are looked at.
    """

    def __init__(self, values_cube: IndexedCube, fft_cube: IndexedCube | None = None,
                 workers: int | None = None, device: int | None = None) -> None:
        self._values_cube = values_cube
        self._workers = workers
        self._fft_dev = None  # (DeviceBuffer, device): the complex64 spectra on the GPU, when they were computed there
        if fft_cube is None:
            from regularizepsf_amd import _native

            shape = values_cube.sample_shape
            if device is None or shape[0] != shape[1] or shape[0] not in _native.SUPPORTED_PATCH_SIZES or len(values_cube) == 0:
                # reference backend; also for sample sizes the GPU spectrum kernel has no plan for
                fft_cube = IndexedCube(values_cube.coordinates, scipy.fft.fft2(values_cube.values, workers=workers))
            else:
                buf = _native.psf_fft_device(values_cube.values, device=device)
                self._fft_dev = (buf, device)
                full = (len(values_cube), *shape)
                fft_cube = IndexedCube._deferred(values_cube.coordinates, full, lambda: buf.download(full, np.complex64))
        self._fft_cube = fft_cube

        if fft_cube.sample_shape != values_cube.sample_shape:
            msg = (f"Values cube and FFT cube have different sample shapes: "
                   f"{values_cube.sample_shape} != {fft_cube.sample_shape}.")
            raise IncorrectShapeError(msg)
        if len(fft_cube) != len(values_cube):
            msg = (f"Values cube and FFT cube have different sample counts: "
                   f"{len(values_cube)} != {len(fft_cube)}.")
            raise IncorrectShapeError(msg)
        if np.any(np.array(values_cube.coordinates) != np.array(fft_cube.coordinates)):
            msg = "Values cube and FFT cube have different coordinates"
            raise InvalidCoordinateError(msg)

    @classmethod
    def _from_device(cls, coordinates: list[tuple[int, int]], size: int, values_buf, fft_buf, device: int) -> "ArrayPSF":
        """Samples and spectra that were produced on the GPU (regularizepsf_amd.functional, built-in device models): both
        cubes are fetched only when somebody looks at them; ``construct`` uses the resident spectra."""
        full = (len(coordinates), size, size)
        self = cls.__new__(cls)
        self._workers = None
        self._values_cube = IndexedCube._deferred(coordinates, full, lambda: values_buf.download(full, np.float32))
        self._fft_cube = IndexedCube._deferred(coordinates, full, lambda: fft_buf.download(full, np.complex64))
        self._fft_dev = (fft_buf, device)
        return self

    @property
    def coordinates(self) -> list[tuple[int, int]]:
        return self._values_cube.coordinates

    @property
    def values(self) -> np.ndarray:
        return self._values_cube.values

    @property
    def fft_evaluations(self) -> np.ndarray:
        return self._fft_cube.values

    @property
    def sample_shape(self) -> tuple[int, int]:
        return self._values_cube.sample_shape

    def __getitem__(self, coord: tuple[int, int]) -> np.ndarray:
        return self._values_cube[coord]

    def fft_at(self, coord: tuple[int, int]) -> np.ndarray:
        return self._fft_cube[coord]

    def __len__(self) -> int:
        return len(self._values_cube)

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, ArrayPSF):
            msg = "Can only compare ArrayPSF to other ArrayPSF."
            raise TypeError(msg)
        return self._values_cube == other._values_cube and self._fft_cube == other._fft_cube

    __hash__ = None

    # ------------------------------------------------------------------ persistence (psf.py:262-332)
    def save(self, path: pathlib.Path) -> None:
        """Save to ``.h5``: datasets ``coordinates``, ``values``, ``fft_evaluations`` as the reference writes them
        (``psf.py:276-280``; like upstream, an existing file is replaced)."""
        path = pathlib.Path(path)
        if path.suffix == ".h5":
            from regularizepsf_amd import _h5min

            _h5min.write_datasets(path, {"coordinates": np.array(self.coordinates, dtype=np.int64).reshape(-1, 2),
                                         "values": self.values, "fft_evaluations": self.fft_evaluations}, overwrite=True)
        elif path.suffix == ".fits":
            msg = "FITS persistence is not implemented in this package (lossy CompImageHDU semantics live in astropy; see DESIGN.md)"
            raise NotImplementedError(msg)
        else:
            msg = f"Unsupported file type {path.suffix}. Change to .h5 or .fits."
            raise NotImplementedError(msg)

    @classmethod
    def load(cls, path: pathlib.Path) -> "ArrayPSF":
        """Load a PSF model saved by this package or by the reference (``psf.py:307-313``); the stored spectra
        are used as they are, not recomputed."""
        path = pathlib.Path(path)
        if path.suffix == ".h5":
            from regularizepsf_amd import _h5min

            data = _h5min.read_datasets(path, ["coordinates", "values", "fft_evaluations"])
            coordinates = [tuple(int(v) for v in c) for c in data["coordinates"]]
            return cls(IndexedCube(coordinates, data["values"]), IndexedCube(coordinates, data["fft_evaluations"]))
        if path.suffix == ".fits":
            msg = "FITS persistence is not implemented in this package (lossy CompImageHDU semantics live in astropy; see DESIGN.md)"
            raise NotImplementedError(msg)
        msg = f"Unsupported file type {path.suffix}. Change to .h5 or .fits."
        raise NotImplementedError(msg)


def __getattr__(name: str):
    """The reference keeps its functional models in the same module (regularizepsf/psf.py:25-189); here they live in
    regularizepsf_amd.functional, which needs ArrayPSF - resolve the upstream import path lazily."""
    if name in ("SimpleFunctionalPSF", "VariedFunctionalPSF", "simple_functional_psf", "varied_functional_psf"):
        from regularizepsf_amd import functional

        return getattr(functional, name)
    msg = f"module {__name__!r} has no attribute {name!r}"
    raise AttributeError(msg)
