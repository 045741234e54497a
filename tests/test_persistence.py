""".h5 persistence without h5py (regularizepsf/transform.py:220-282, psf.py:262-332).

The golden files were written by the reference's own ``save`` methods through h5py
(tests/golden/make_h5_golden.py); h5_expected.npz holds the arrays that went in.  The reference's tests for
this behaviour are tests/test_transform.py:11-27,85-97 and tests/test_psf.py:17-31 (round trips)."""

from __future__ import annotations

import pathlib
import shutil
import struct
import subprocess

import numpy as np
import pytest

import regularizepsf_amd as rp
from regularizepsf_amd import _h5min
from tests.helpers import GOLDEN

EXPECTED = np.load(GOLDEN / "h5_expected.npz")
COORDS = [tuple(int(v) for v in c) for c in EXPECTED["coords"]]
CONDA = pathlib.Path("/opt/conda/bin/python3.9")


@pytest.mark.parametrize("tag", ["c64", "c128"])
def test_load_transform_written_by_the_reference(tag):
    t = rp.ArrayPSFTransform.load(GOLDEN / f"h5_transform_{tag}.h5")
    assert t.coordinates == COORDS
    k = t._transfer_kernel.values
    assert k.dtype == EXPECTED[f"transform_{tag}"].dtype and np.array_equal(k, EXPECTED[f"transform_{tag}"])
    assert t.psf_shape == (8, 8) and len(t) == len(COORDS)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_load_psf_written_by_the_reference(tag):
    p = rp.ArrayPSF.load(GOLDEN / f"h5_psf_{tag}.h5")
    assert p.coordinates == COORDS
    assert p.values.dtype == EXPECTED[f"psf_{tag}_values"].dtype
    assert np.array_equal(p.values, EXPECTED[f"psf_{tag}_values"])
    assert np.array_equal(p.fft_evaluations, EXPECTED[f"psf_{tag}_fft"])  # stored spectra, not recomputed


def test_transform_round_trip_and_overwrite_rules(tmp_path):
    """tests/test_transform.py:11-27 (round trip) and :85-97 (no silent overwrite) upstream."""
    k = EXPECTED["transform_c64"]
    t = rp.ArrayPSFTransform(rp.IndexedCube(COORDS, k))
    path = tmp_path / "transform.h5"
    t.save(path)
    loaded = rp.ArrayPSFTransform.load(path)
    assert loaded == t and loaded.coordinates == t.coordinates
    assert np.array_equal(loaded._transfer_kernel.values, k)
    with pytest.raises(FileExistsError):
        t.save(path)
    t.save(path, overwrite=True)
    assert rp.ArrayPSFTransform.load(path) == t
    for bad in ("transform.fits", "transform.txt"):
        with pytest.raises(NotImplementedError):
            t.save(tmp_path / bad)
        with pytest.raises(NotImplementedError):
            rp.ArrayPSFTransform.load(tmp_path / bad)


def test_psf_round_trip(tmp_path):
    """tests/test_psf.py:17-31 upstream."""
    p = rp.ArrayPSF(rp.IndexedCube(COORDS, EXPECTED["psf_f64_values"]))
    path = tmp_path / "psf.h5"
    p.save(path)
    p.save(path)  # like upstream (mode "w"), an existing file is replaced
    loaded = rp.ArrayPSF.load(path)
    assert loaded == p
    assert np.array_equal(loaded.values, p.values) and np.array_equal(loaded.fft_evaluations, p.fft_evaluations)
    with pytest.raises(NotImplementedError):
        p.save(tmp_path / "psf.fits")
    with pytest.raises(NotImplementedError):
        rp.ArrayPSF.load(tmp_path / "psf.json")


def test_writer_layout_matches_the_reference_files(tmp_path):
    """Same structures as the h5py-written golden: superblock v0, symbol-table root, identical dataset headers
    (dataspace, datatype, fill, layout messages) apart from addresses and the modification time."""
    path = tmp_path / "t.h5"
    _h5min.write_datasets(path, {"coordinates": EXPECTED["coords"], "transfer_kernel": EXPECTED["transform_c64"]})
    mine, ref = path.read_bytes(), (GOLDEN / "h5_transform_c64.h5").read_bytes()
    assert mine[:8] == ref[:8] == _h5min.SIGNATURE
    assert mine[8:24] == ref[8:24]  # versions, offset/length sizes, group K values
    assert struct.unpack_from("<Q", mine, 40)[0] == len(mine)  # end-of-file address

    def messages(blob, name):
        r = _h5min._Reader(blob)
        return {kind: body for kind, body in r.messages(r.root()[name]) if kind not in (0x0000, 0x0012, 0x0008)}

    for name in ("coordinates", "transfer_kernel"):
        assert messages(mine, name) == messages(ref, name)


def test_small_types_scalars_and_empty(tmp_path):
    path = tmp_path / "misc.h5"
    data = {"u8": np.arange(7, dtype=np.uint8), "i32": np.array([[1, -2], [3, 4]], np.int32), "empty": np.zeros((0, 3), np.float32),
            "scalar": np.float64(3.5), "c128": np.array([1 + 2j, 3 - 4j]), "h": np.arange(4, dtype=np.int16)}
    _h5min.write_datasets(path, data)
    back = _h5min.read_datasets(path)
    assert sorted(back) == sorted(data)
    for name, value in data.items():
        v = np.asarray(value)
        assert back[name].dtype == v.dtype and back[name].shape == v.shape and np.array_equal(back[name], v)
    with pytest.raises(KeyError):
        _h5min.read_datasets(path, ["missing"])
    with pytest.raises(NotImplementedError):
        _h5min.write_datasets(tmp_path / "s.h5", {"s": np.array(["a", "b"])})
    with pytest.raises(ValueError):
        _h5min.write_datasets(tmp_path / "s.h5", {"a/b": np.zeros(2)})


def test_unsupported_and_damaged_files_are_reported(tmp_path):
    junk = tmp_path / "junk.h5"
    junk.write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(_h5min.H5FormatError):
        _h5min.read_datasets(junk)
    ref = bytearray((GOLDEN / "h5_transform_c64.h5").read_bytes())
    newer = bytes(ref[:8]) + b"\x02" + bytes(ref[9:])  # superblock version 2 (libver="latest")
    (tmp_path / "v2.h5").write_bytes(newer)
    with pytest.raises(NotImplementedError, match="superblock version 2"):
        _h5min.read_datasets(tmp_path / "v2.h5")
    (tmp_path / "cut.h5").write_bytes(bytes(ref[:0x900]))  # data block truncated
    with pytest.raises(_h5min.H5FormatError):
        _h5min.read_datasets(tmp_path / "cut.h5")


@pytest.mark.skipif(not CONDA.exists(), reason="no interpreter with h5py in this environment")
def test_files_we_write_open_in_real_hdf5(tmp_path):
    """libhdf5 (through h5py in the image's conda Python) reads our files and finds the same arrays."""
    probe = subprocess.run([str(CONDA), "-c", "import h5py"], capture_output=True)
    if probe.returncode != 0:
        pytest.skip("h5py not importable")
    t = rp.ArrayPSFTransform(rp.IndexedCube(COORDS, EXPECTED["transform_c128"]))
    p = rp.ArrayPSF(rp.IndexedCube(COORDS, EXPECTED["psf_f32_values"]))
    t.save(tmp_path / "t.h5")
    p.save(tmp_path / "p.h5")
    shutil.copy(GOLDEN / "h5_expected.npz", tmp_path / "expected.npz")
    code = f"""
import h5py, numpy as np
e = np.load(r"{tmp_path / 'expected.npz'}")
with h5py.File(r"{tmp_path / 't.h5'}", "r") as f:
    assert sorted(f.keys()) == ["coordinates", "transfer_kernel"]
    assert np.array_equal(f["coordinates"][:], e["coords"]) and f["coordinates"].dtype == np.int64
    assert np.array_equal(f["transfer_kernel"][:], e["transform_c128"]) and f["transfer_kernel"].dtype == np.complex128
with h5py.File(r"{tmp_path / 'p.h5'}", "r") as f:
    assert np.array_equal(f["values"][:], e["psf_f32_values"]) and f["values"].dtype == np.float32
    assert f["fft_evaluations"].dtype == np.complex64 and f["fft_evaluations"].shape == e["psf_f32_fft"].shape
print("ok")
"""
    out = subprocess.run([str(CONDA), "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr
