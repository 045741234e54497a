"""Golden vectors for the functional PSF models, produced by the REAL reference (build container only; same import route as
make_golden.py).  The model callables are this package's NumPy formulas; what the fixture pins is everything the reference
does around them: the meshgrid it samples on, partial application of the varied parameters, extra keyword arguments,
stacking order, dtype - and the spectra ArrayPSF computes from the samples.

Usage:  python tests/golden/make_functional_golden.py
"""

from __future__ import annotations

import pathlib
import sys

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from regularizepsf_amd.functional import _elliptical_gaussian, _moffat  # noqa: E402
from tests.golden.make_golden import load_reference  # noqa: E402


def gaussian_field(row, col):
    return {"amplitude": 2.0 + row / 300, "row0": 8 + 0.3 * col / 40, "col0": 7.5, "sigma_row": 1.5 + row / 100, "sigma_col": 2.0 + col / 90,
            "theta": 0.4 + (row + col) / 200, "background": 1e-3}


def moffat_field(row, col):
    return {"amplitude": 1.0, "row0": 8 + row / 100, "col0": 8 - col / 90, "alpha": 2.0 + col / 40, "beta": 2.5 + row / 60, "background": 0.0}


COORDS = [(-8, -8), (0, 8), (24, 16), (40, 56)]
SIZE = 16

if __name__ == "__main__":
    _, psf, _ = load_reference()
    out = {"coords": np.array(COORDS, np.int64), "size": np.array(SIZE)}
    simple = psf.simple_functional_psf(lambda row, col, a=10: 100 * row + col + a)
    out["simple_values"] = simple.as_array_psf([(0, 0), (1, 0)], 5, a=3).values
    for name, base_f, field in (("gaussian", _elliptical_gaussian, gaussian_field), ("moffat", _moffat, moffat_field)):
        varied = psf.varied_functional_psf(psf.simple_functional_psf(base_f))(field)
        arr = varied.as_array_psf(COORDS, SIZE)
        out[f"{name}_values"], out[f"{name}_fft"] = arr.values, arr.fft_evaluations
        assert arr.values.dtype == np.float64 and arr.values.shape == (len(COORDS), SIZE, SIZE)
    np.savez_compressed(HERE / "functional.npz", **out)
    print("functional.npz:", {k: v.shape for k, v in out.items()})
