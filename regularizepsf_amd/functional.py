"""Functional PSF models (API of regularizepsf/psf.py:25-189) and their evaluation into ``ArrayPSF`` objects.

Same names, decorators, signature checks and exceptions as the reference: a *simple* model is a callable of
``(row, col, **parameters)``, a *varied* model maps an image position to the parameter values of a simple model.
``as_array_psf(coordinates, size)`` samples a model on the ``size`` x ``size`` grid of every patch.  The reference does
that with one Python call per patch on the host (psf.py:65-70,159-165).  Here that remains the route for arbitrary
callables - optionally followed by the device spectrum kernel, ``device=<gpu index>`` - and two built-in models,
``elliptical_gaussian`` and ``moffat``, can in addition be **rasterised on the GPU** (kernel K6, ``rpsf_psf_model_fft_device``):
only the per-patch parameter table crosses PCIe, samples and spectra stay on the device, and
``ArrayPSFTransform.construct`` -> ``apply`` continue from there.
"""

from __future__ import annotations

import inspect
from functools import partial
from typing import Any, Callable

import numpy as np

from regularizepsf_amd.exceptions import InvalidFunctionError
from regularizepsf_amd.psf import ArrayPSF
from regularizepsf_amd.util import IndexedCube


def _sample_grid(size: int) -> tuple[np.ndarray, np.ndarray]:
    """The (row, col) arrays the reference hands to a model: ``np.meshgrid(arange, arange)`` with its default 'xy'
    indexing, i.e. element [i, j] is evaluated at row = j, col = i (psf.py:67,161)."""
    return np.meshgrid(np.arange(size), np.arange(size))


class SimpleFunctionalPSF:
    """A PSF given as a function of (row, col) and named parameters (psf.py:25-77)."""

    def __init__(self, function: Callable) -> None:
        self._f = function
        self._signature = inspect.signature(function)
        names = list(self._signature.parameters)
        if len(names) < 2:  # noqa: PLR2004
            msg = "row and col must be the first two arguments in your model equation."
            raise InvalidFunctionError(msg)
        if names[0] != "row":
            msg = "row must be the first arguments in your model equation."
            raise InvalidFunctionError(msg)
        if names[1] != "col":
            msg = "col must be the second arguments in your model equation"
            raise InvalidFunctionError(msg)
        self._parameters = set(names[2:])

    def __call__(self, row, col, **kwargs: Any):
        return self._f(row, col, **kwargs)

    @property
    def parameters(self) -> set[str]:
        return self._parameters

    @property
    def f(self) -> Callable:
        return self._f

    def as_array_psf(self, coordinates: list[tuple[int, int]], size: int, device: int | None = None, **kwargs: Any) -> ArrayPSF:
        """The same samples at every coordinate (psf.py:65-70).  ``device``: spectra on that GPU (``ArrayPSF(device=)``)."""
        rr, cc = _sample_grid(size)
        evaluation = self(rr, cc, **kwargs)
        return ArrayPSF(IndexedCube(coordinates, np.stack([evaluation for _ in coordinates])), device=device)


def simple_functional_psf(arg: Any = None) -> SimpleFunctionalPSF:
    """Decorator: ``@simple_functional_psf`` on ``def model(row, col, ...)`` (psf.py:80-85)."""
    if callable(arg):
        return SimpleFunctionalPSF(arg)
    msg = "psf decorator must have no arguments."
    raise TypeError(msg)


class DeviceModelPSF(SimpleFunctionalPSF):
    """A simple model the library can also evaluate on the GPU.  On the host it is an ordinary ``SimpleFunctionalPSF``
    (NumPy formula below); ``pack`` lays its parameters out as kernel K6 reads them (include/rpsf.h)."""

    def __init__(self, function: Callable, kind: str, slots: dict[str, int]) -> None:
        super().__init__(function)
        self.kind = kind
        self._slots = slots
        self._defaults = {k: v.default for k, v in self._signature.parameters.items() if v.default is not inspect.Parameter.empty}

    def pack(self, parameter_sets: list[dict[str, Any]]) -> np.ndarray:
        from regularizepsf_amd import _native

        table = np.zeros((len(parameter_sets), _native.MODEL_PARAMS), np.float64)
        for line, given in zip(table, parameter_sets):
            unknown = set(given) - self._parameters
            if unknown:
                msg = f"{self.kind} has no parameters {sorted(unknown)}"
                raise InvalidFunctionError(msg)
            for name, slot in self._slots.items():
                if name in given:
                    line[slot] = float(given[name])
                elif name in self._defaults:
                    line[slot] = float(self._defaults[name])
                else:
                    msg = f"{self.kind} needs a value for {name!r}"
                    raise InvalidFunctionError(msg)
        return table

    def rasterize(self, coordinates: list[tuple[int, int]], size: int, parameter_sets: list[dict[str, Any]], device: int,
                  normalize: bool = False) -> ArrayPSF:
        from regularizepsf_amd import _native

        values, spectra = _native.psf_model_fft_device(self.kind, size, self.pack(parameter_sets), normalize, device)
        return ArrayPSF._from_device(coordinates, size, values, spectra, device)

    def as_array_psf(self, coordinates: list[tuple[int, int]], size: int, device: int | None = None, normalize: bool = False,
                     **kwargs: Any) -> ArrayPSF:
        from regularizepsf_amd import _native

        if device is None or size not in _native.SUPPORTED_PATCH_SIZES or not coordinates:
            psf = super().as_array_psf(coordinates, size, device=device, **kwargs)
            return _normalized(psf, device) if normalize else psf
        return self.rasterize(coordinates, size, [kwargs] * len(coordinates), device, normalize)


def _normalized(psf: ArrayPSF, device: int | None) -> ArrayPSF:
    values = psf.values
    return ArrayPSF(IndexedCube(psf.coordinates, values / values.sum(axis=(1, 2), keepdims=True)), device=device)


def _elliptical_gaussian(row, col, amplitude=1.0, row0=0.0, col0=0.0, sigma_row=1.0, sigma_col=1.0, theta=0.0, background=0.0):
    dr, dc = row - row0, col - col0
    u = dr * np.cos(theta) + dc * np.sin(theta)
    v = dc * np.cos(theta) - dr * np.sin(theta)
    return background + amplitude * np.exp(-0.5 * ((u * u) / (sigma_row * sigma_row) + (v * v) / (sigma_col * sigma_col)))


def _moffat(row, col, amplitude=1.0, row0=0.0, col0=0.0, alpha=1.0, beta=2.5, background=0.0):
    dr, dc = row - row0, col - col0
    return background + amplitude * np.power(1.0 + (dr * dr + dc * dc) / (alpha * alpha), -beta)


#: ``background + amplitude * exp(-(u^2 / sigma_row^2 + v^2 / sigma_col^2) / 2)``, (u, v) = (row - row0, col - col0) rotated by theta
elliptical_gaussian = DeviceModelPSF(_elliptical_gaussian, "elliptical_gaussian",
                                     {"amplitude": 0, "row0": 1, "col0": 2, "sigma_row": 3, "sigma_col": 4, "theta": 5, "background": 6})
#: ``background + amplitude * (1 + ((row - row0)^2 + (col - col0)^2) / alpha^2) ** -beta``
moffat = DeviceModelPSF(_moffat, "moffat", {"amplitude": 0, "row0": 1, "col0": 2, "alpha": 3, "beta": 4, "background": 6})


class VariedFunctionalPSF:
    """A simple model whose parameters depend on the position in the image (psf.py:88-165)."""

    def __init__(self, vary_function: Callable, base_psf: SimpleFunctionalPSF, validate_at_call: bool = True) -> None:
        self._vary_function = vary_function
        self._base_psf = base_psf
        self.validate_at_call = validate_at_call
        self.parameterization_signature = inspect.signature(vary_function)
        names = list(self.parameterization_signature.parameters)
        if len(names) < 2:  # noqa: PLR2004
            msg = f"Found {len(names)}"
            raise InvalidFunctionError(msg)
        if len(names) > 2:  # noqa: PLR2004
            msg = f"Found function requiring{len(names)}arguments. Expected 2, only `row` and `col`."
            raise InvalidFunctionError(msg)
        if names[0] != "row":
            msg = "row must be the first argument in your parameterization equation."
            raise InvalidFunctionError(msg)
        if names[1] != "col":
            msg = "col must be the second argument in your parameterization equation"
            raise InvalidFunctionError(msg)
        self._origin_parameters = set(vary_function(0, 0).keys())  # the parameter names are fixed by what the origin supplies
        if self._base_psf.parameters != self._origin_parameters:
            msg = (f"The base PSF model has parameters {self._base_psf.parameters} while the varied psf supplies "
                   f"{self._origin_parameters}at the origin. These must match.")
            raise InvalidFunctionError(msg)

    def _variance(self, row, col) -> dict[str, Any]:
        variance = self._vary_function(row, col)
        if self.validate_at_call and set(variance.keys()) != self.parameters:
            msg = (f"At (row, col) the varying parameters were {set(variance.keys())} when the parameters were expected as "
                   f"{self.parameters}.")
            raise InvalidFunctionError(msg)
        return variance

    def __call__(self, row, col):
        return self._base_psf(row, col, **self._variance(row, col))

    @property
    def parameters(self) -> set[str]:
        return self._base_psf.parameters

    def simplify(self, row: int, col: int) -> SimpleFunctionalPSF:
        """The simple model this one is at (row, col) (psf.py:154-157; like upstream, no validation here)."""
        return simple_functional_psf(partial(self._base_psf.f, **self._vary_function(row, col)))

    def as_array_psf(self, coordinates: list[tuple[int, int]], size: int, device: int | None = None, normalize: bool = False,
                     **kwargs: Any) -> ArrayPSF:
        """One sample array per coordinate, the model's parameters taken at that coordinate (psf.py:159-165).

        ``device`` (extension): with a built-in device model as the base the samples are rasterised on that GPU from the
        parameter table alone; with any other base they are evaluated here, as upstream, and only the spectra are computed
        on the device.  ``normalize`` (extension): every patch is scaled to unit sum.
        """
        from regularizepsf_amd import _native

        base = self._base_psf
        if (device is not None and isinstance(base, DeviceModelPSF) and size in _native.SUPPORTED_PATCH_SIZES and coordinates):
            sets = [{**self._vary_function(row, col), **kwargs} for row, col in coordinates]
            return base.rasterize(coordinates, size, sets, device, normalize)
        rr, cc = _sample_grid(size)
        values = [self.simplify(row, col)(rr, cc, **kwargs) for row, col in coordinates]
        psf = ArrayPSF(IndexedCube(coordinates, np.stack(values)), device=device)
        return _normalized(psf, device) if normalize else psf


def _varied_functional_psf(base_psf: SimpleFunctionalPSF) -> Callable:
    if base_psf is None:
        msg = "A base_psf must be provided to the varied_psf decorator."
        raise TypeError(msg)

    def inner(__fn: Callable | None = None, *, check_at_call: bool = True):
        if __fn:
            return VariedFunctionalPSF(__fn, base_psf, validate_at_call=check_at_call)
        return partial(inner, check_at_call=check_at_call)

    return inner


def varied_functional_psf(base_psf: SimpleFunctionalPSF | None = None):
    """Decorator factory: ``@varied_functional_psf(base)`` on ``def parameters(row, col) -> dict`` (psf.py:180-189)."""
    if isinstance(base_psf, SimpleFunctionalPSF):
        return _varied_functional_psf(base_psf)
    if callable(base_psf):
        msg = "varied_psf decorator must be calledwith an argument for the base_psf."
        raise TypeError(msg)
    msg = "varied_psf decorator expects exactlyone argument of type PSF."
    raise TypeError(msg)
