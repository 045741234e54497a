"""MI355X-native patch-wise Fourier PSF correction with the regularizepsf class API.

Drop-in for the ``ArrayPSF`` / ``ArrayPSFTransform`` / ``IndexedCube`` / ``calculate_covering`` path (and the functional
PSF models that feed it) of
punch-mission/regularizepsf (regularizepsf/__init__.py:5-16 re-exports the same names); the compute
runs in hand-written HIP kernels behind the C ABI of include/rpsf.h.
"""

from regularizepsf_amd.exceptions import (
    FunctionParameterMismatchError,
    IncorrectShapeError,
    InvalidCoordinateError,
    InvalidDataError,
    InvalidFunctionError,
    PSFBuilderError,
    RegularizePSFError,
)
from regularizepsf_amd.functional import (
    SimpleFunctionalPSF,
    VariedFunctionalPSF,
    elliptical_gaussian,
    moffat,
    simple_functional_psf,
    varied_functional_psf,
)
from regularizepsf_amd._native import pinned_empty
from regularizepsf_amd.psf import ArrayPSF
from regularizepsf_amd.transform import ArrayPSFTransform
from regularizepsf_amd.util import IndexedCube, calculate_covering

__version__ = "0.1.0"

__all__ = [
    "ArrayPSF",
    "ArrayPSFTransform",
    "FunctionParameterMismatchError",
    "IncorrectShapeError",
    "IndexedCube",
    "InvalidCoordinateError",
    "InvalidDataError",
    "InvalidFunctionError",
    "PSFBuilderError",
    "RegularizePSFError",
    "SimpleFunctionalPSF",
    "VariedFunctionalPSF",
    "calculate_covering",
    "elliptical_gaussian",
    "moffat",
    "pinned_empty",
    "simple_functional_psf",
    "varied_functional_psf",
]
