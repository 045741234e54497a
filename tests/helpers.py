"""Shared test helpers: seeded synthetic cases used by the golden generator and the parity tests."""

from __future__ import annotations

import hashlib
import pathlib

import numpy as np

from oracle import regpsf_oracle as orc

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"

# (name, H, W, N, alpha, eps, psf kind, pad_mode, image seed)
APPLY_CASES = [
    ("n32_sym", 96, 80, 32, 3.0, 0.1, "coma", "symmetric", 11),
    ("n32_reflect", 96, 80, 32, 1.0, 0.1, "coma", "reflect", 12),
    ("n32_constant", 70, 100, 32, 2.0, 0.01, "gauss", "constant", 13),
    ("n32_edge", 64, 64, 32, 1.0, 0.1, "coma", "edge", 14),
    ("n32_wrap", 64, 96, 32, 1.0, 0.1, "coma", "wrap", 15),
    ("n32_mean", 64, 96, 32, 1.0, 0.1, "coma", "mean", 16),
    ("n16_sym", 40, 56, 16, 1.0, 0.1, "gauss", "symmetric", 17),
    ("n64_sym", 160, 200, 64, 3.0, 0.1, "coma", "symmetric", 21),
    ("n64_identity", 128, 128, 64, 3.0, 0.1, "identity", "symmetric", 22),
    ("n128_sym", 300, 256, 128, 1.0, 0.1, "coma", "symmetric", 31),
    ("n256_sym", 300, 280, 256, 3.0, 0.1, "coma", "symmetric", 41),
    ("n256_identity", 512, 512, 256, 3.0, 0.1, "identity", "symmetric", 42),
]


def make_psfs(kind: str, coords, n: int, h: int, w: int):
    """Source / target PSF cubes (float64) for a case; shared with tests via tests/helpers.py."""
    if kind == "coma":
        src = np.stack([orc.coma_psf(n, r, c, h, w) for r, c in coords])
        tgt = np.broadcast_to(orc.gaussian_psf(n, 1.8), src.shape).copy()
    elif kind == "gauss":
        src = np.broadcast_to(orc.gaussian_psf(n, 1.8), (len(coords), n, n)).copy()
        tgt = np.broadcast_to(orc.gaussian_psf(n, 1.5), src.shape).copy()
    else:  # identity: source == target, as in the reference's tests/test_transform.py:29-49
        src = np.broadcast_to(orc.gaussian_psf(n, 3 / 2.355), (len(coords), n, n)).astype(np.float32)
        tgt = src
    return src, tgt




def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_apply_case(name: str):
    """Return (fixture dict, coords list, K complex64 rebuilt with the oracle and checked against the stored hash)."""
    fx = np.load(GOLDEN / f"apply_{name}.npz")
    h, w, n = (int(v) for v in fx["meta"])
    coords = [tuple(int(v) for v in t) for t in fx["coords"]]
    kind = str(fx["kind"])
    src, tgt = make_psfs(kind, coords, n, h, w)
    s_fft = orc.psf_fft(src)
    t_fft = s_fft if kind == "identity" else orc.psf_fft(tgt)
    with np.errstate(all="ignore"):
        k = orc.construct_transfer(s_fft, t_fft, float(fx["alpha"]), float(fx["eps"])).astype(np.complex64)
    assert sha(k) == str(fx["k_sha256"]), f"oracle construct differs from the reference for case {name}"
    return fx, coords, k


def rel_errors(out: np.ndarray, ref: np.ndarray) -> tuple[float, float]:
    """(max|d| / max|ref|, ||d||2 / ||ref||2): the parity metric of SURVEY.md 8d."""
    d = out.astype(np.float64) - ref.astype(np.float64)
    return float(np.abs(d).max() / np.abs(ref).max()), float(np.linalg.norm(d) / np.linalg.norm(ref))


def load_c128_case():
    """The complex128 fixture (tests/golden/make_golden.py, case v): (fixture, coords, K complex128 rebuilt with the oracle)."""
    fx = np.load(GOLDEN / "apply_c128_n64.npz")
    h, w, n = (int(v) for v in fx["meta"])
    coords = [tuple(int(v) for v in t) for t in fx["coords"]]
    src, tgt = make_psfs("coma", coords, n, h, w)
    k = orc.construct_transfer(orc.psf_fft(src), orc.psf_fft(tgt), float(fx["alpha"]), float(fx["eps"]))
    assert k.dtype == np.complex128 and sha(k) == str(fx["k_sha256"]), "oracle construct differs from the reference (complex128 case)"
    return fx, coords, k
