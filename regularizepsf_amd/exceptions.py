"""Exception hierarchy, same names as regularizepsf/exceptions.py:4-23 so that ``except`` clauses keep working."""


class RegularizePSFError(Exception):
    """Root of every error raised by this package."""


class InvalidCoordinateError(RegularizePSFError):
    """A patch coordinate is unknown to the cube / model, or two models disagree on coordinates."""


class IncorrectShapeError(RegularizePSFError):
    """Array shapes are inconsistent with the cube / model they are used with."""


class InvalidFunctionError(RegularizePSFError):
    """A functional PSF was declared with an unusable signature."""


class FunctionParameterMismatchError(RegularizePSFError):
    """A functional PSF was evaluated with parameters it does not declare."""


class PSFBuilderError(RegularizePSFError):
    """PSF model estimation failed."""


class InvalidDataError(RegularizePSFError):
    """Input data unusable for PSF model estimation."""
