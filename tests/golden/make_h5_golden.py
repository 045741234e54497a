"""Generate the .h5 golden files by running the REAL reference's save() methods.

Needs an interpreter that has h5py (absent from the build image's main Python; the image's
/opt/conda/bin/python3.9 has h5py 3.3.0 / HDF5 1.10.6) and /root/reference:

    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py

Import route: astropy is stubbed, regularizepsf.visualize is stubbed (it needs
Python >= 3.10 annotations), a bare ``regularizepsf`` package points at the reference tree.
Outputs (data only): h5_transform_c64.h5, h5_transform_c128.h5, h5_psf_f32.h5, h5_psf_f64.h5 and
h5_expected.npz with the arrays that were saved.
"""
import importlib
import pathlib
import sys
import types

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent


def load_reference(root="/root/reference"):
    import matplotlib

    matplotlib.use("Agg")
    for name in ("astropy", "astropy.io", "astropy.io.fits"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["astropy"].io = sys.modules["astropy.io"]
    sys.modules["astropy.io"].fits = sys.modules["astropy.io.fits"]
    vis = types.ModuleType("regularizepsf.visualize")
    vis.visualize_grid = vis.visualize_patch = lambda *a, **k: None
    vis.KERNEL_IMSHOW_ARGS_DEFAULT = vis.PSF_IMSHOW_ARGS_DEFAULT = {}
    pkg = types.ModuleType("regularizepsf")
    pkg.__path__ = [root + "/regularizepsf"]
    sys.modules["regularizepsf"] = pkg
    sys.modules["regularizepsf.visualize"] = vis
    util = importlib.import_module("regularizepsf.util")
    psf = importlib.import_module("regularizepsf.psf")
    transform = importlib.import_module("regularizepsf.transform")
    return util, psf, transform


def main():
    util, psf, transform = load_reference()
    rng = np.random.default_rng(20260101)
    n = 8
    coords = [tuple(int(v) for v in c) for c in util.calculate_covering((16, 24), n)]
    expected = {"coords": np.array(coords, dtype=np.int64)}
    for tag, ctype in (("c64", np.complex64), ("c128", np.complex128)):
        k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(ctype)
        t = transform.ArrayPSFTransform(util.IndexedCube(coords, k))
        path = HERE / f"h5_transform_{tag}.h5"
        t.save(path, overwrite=True)
        expected[f"transform_{tag}"] = k
    for tag, ftype in (("f32", np.float32), ("f64", np.float64)):
        v = rng.random((len(coords), n, n)).astype(ftype)
        p = psf.ArrayPSF(util.IndexedCube(coords, v))
        path = HERE / f"h5_psf_{tag}.h5"
        if path.exists():
            path.unlink()
        p.save(path)
        expected[f"psf_{tag}_values"] = p.values
        expected[f"psf_{tag}_fft"] = p.fft_evaluations
    np.savez_compressed(HERE / "h5_expected.npz", **expected)
    for f in sorted(HERE.glob("h5_*")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
