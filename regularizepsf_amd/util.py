"""Patch geometry and the coordinate-indexed cube (API of regularizepsf/util.py:10-172)."""

from __future__ import annotations

import numpy as np

from regularizepsf_amd.exceptions import IncorrectShapeError, InvalidCoordinateError


def calculate_covering(image_shape: tuple[int, int], size: int) -> np.ndarray:
    """Corners of the half-overlapping patch lattice that covers every pixel exactly four times.

    Same contract and, importantly, same ORDER as regularizepsf/util.py:10-53 (the transfer
    kernel's first axis is indexed by it): four sub-lattices of pitch ``size`` -- origin (0, 0),
    (-h, -h), (-h, 0), (0, -h) with h = ceil(size / 2) -- each enumerated with the first (row)
    coordinate varying fastest.  Returns an (n, 2) array of (row, col) corners.
    """
    half = np.ceil(size / 2).astype(int)
    height, width = image_shape[0], image_shape[1]
    corners = []
    for row_start, col_start in ((0, 0), (-half, -half), (-half, 0), (0, -half)):
        rows = np.arange(row_start, height, size)
        cols = np.arange(col_start, width, size)
        block = np.empty((len(cols) * len(rows), 2), dtype=np.result_type(rows, cols))
        block[:, 0] = np.tile(rows, len(cols))
        block[:, 1] = np.repeat(cols, len(rows))
        corners.append(block)
    return np.concatenate(corners)


class IndexedCube:
    """An (n, rows, cols) stack of samples addressed by the image coordinate of each sample's corner.

    Mirrors regularizepsf/util.py:56-172: no copy of ``values`` is made; lookups by a coordinate
    that is not in the cube raise :class:`InvalidCoordinateError`; assigning a sample of the wrong
    shape raises :class:`IncorrectShapeError`; ``==`` is coordinates-equal, shape-equal and
    ``allclose(rtol=1e-4, atol=1e-6)``, and raises ``TypeError`` against anything else.
    """

    def __init__(self, coordinates: list[tuple[int, int]], values: np.ndarray) -> None:
        if values.ndim != 3:
            msg = "Values must be three dimensional"
            raise IncorrectShapeError(msg)
        if len(coordinates) != values.shape[0]:
            msg = f"{len(coordinates)} coordinates defined but {values.shape[0]} values found."
            raise IncorrectShapeError(msg)
        self._coordinates = coordinates
        self._values_array = values
        self._loader = None
        self._shape = values.shape
        self._where = {tuple(c): layer for layer, c in enumerate(coordinates)}
        self._edits = 0  # bumped by __setitem__, lets device-side copies notice they are stale
        self._frozen = False  # a device copy exists and the array was made read-only for it (ArrayPSFTransform._freeze)

    @classmethod
    def _deferred(cls, coordinates: list[tuple[int, int]], shape: tuple[int, int, int], loader) -> "IndexedCube":
        """A cube whose values live on the GPU until somebody looks at them: ``loader()`` returns the (n, rows, cols)
        array on first access (regularizepsf_amd extension, used by ``ArrayPSF(device=...)`` and ``construct``)."""
        if len(shape) != 3 or len(coordinates) != shape[0]:
            msg = f"{len(coordinates)} coordinates defined but values of shape {shape} announced."
            raise IncorrectShapeError(msg)
        self = cls.__new__(cls)
        self._coordinates = coordinates
        self._values_array = None
        self._loader = loader
        self._load_hook = None  # called once with the cube right after the values have arrived, before anybody can edit them
        self._shape = tuple(shape)
        self._where = {tuple(c): layer for layer, c in enumerate(coordinates)}
        self._edits = 0
        self._frozen = False
        return self

    @property
    def _values(self) -> np.ndarray:
        if self._values_array is None:
            self._values_array = self._loader()
            self._loader = None
            hook, self._load_hook = getattr(self, "_load_hook", None), None
            if hook is not None:
                hook(self)
        return self._values_array

    @property
    def sample_shape(self) -> tuple[int, int]:
        return self._shape[1], self._shape[2]

    @property
    def coordinates(self) -> list[tuple[int, int]]:
        return self._coordinates

    @property
    def values(self) -> np.ndarray:
        return self._values

    def _layer(self, coordinate: tuple[int, int]) -> int:
        try:
            return self._where[coordinate]
        except (KeyError, TypeError):
            msg = f"Coordinate {coordinate} not in TransferKernel."
            raise InvalidCoordinateError(msg) from None

    def __getitem__(self, coordinate: tuple[int, int]) -> np.ndarray:
        return self._values[self._layer(coordinate)]

    def __setitem__(self, coordinate: tuple[int, int], value: np.ndarray) -> None:
        layer = self._layer(coordinate)
        if value.shape != self.sample_shape:
            msg = f"Cannot assign value of shape {value.shape} to transfer kernel of shape {self.sample_shape}."
            raise IncorrectShapeError(msg)
        values = self._values
        if getattr(self, "_frozen", False):  # made read-only when a device copy was taken: this edit is counted, the copy is refreshed at the next apply
            values.flags.writeable = True
            self._frozen = False
        values[layer] = value
        self._edits += 1

    def __len__(self) -> int:
        return len(self._coordinates)

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, IndexedCube):
            msg = "Can only compare IndexedCube instances."
            raise TypeError(msg)
        return (
            self.coordinates == other.coordinates
            and self.sample_shape == other.sample_shape
            and bool(np.allclose(self.values, other.values, rtol=1e-04, atol=1e-06))
        )

    __hash__ = None
