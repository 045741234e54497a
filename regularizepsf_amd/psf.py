"""Array PSF model (API of regularizepsf/psf.py:192-260,402-416)."""

from __future__ import annotations

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
    addition that computes float32 spectra with the HIP kernel instead (complex64 result).
    """

    def __init__(self, values_cube: IndexedCube, fft_cube: IndexedCube | None = None,
                 workers: int | None = None, device: int | None = None) -> None:
        self._values_cube = values_cube
        self._workers = workers
        if fft_cube is None:
            if device is None:
                spectra = scipy.fft.fft2(values_cube.values, workers=workers)
            else:
                from regularizepsf_amd import _native

                spectra = _native.psf_fft(values_cube.values, device=device)
            fft_cube = IndexedCube(values_cube.coordinates, spectra)
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
