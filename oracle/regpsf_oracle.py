"""CPU oracle for the patch-wise Fourier PSF-correction path.  TEST INFRASTRUCTURE ONLY.

This module is a plain NumPy/SciPy restatement of the reference algorithm
(punch-mission/regularizepsf 1.2.0).  It exists to *check* the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The product package ``regularizepsf_amd`` never imports it and has
no CPU fallback.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here
against fixtures under ``tests/golden/`` that were produced by importing the real
reference in the build container (``tests/golden/make_golden.py``), bit-for-bit for the
``apply`` restatement (np.array_equal) and for ``calculate_covering``.

Each function cites the reference lines it follows (paths relative to the reference
checkout).  The FFT itself lives in SciPy's pocketfft (un-vendored dependency of the
reference, unpinned in its pyproject.toml; scipy 1.15.3 in the build image); the
oracle calls the same ``scipy.fft`` entry points as the reference call sites
(transform.py:163-164, psf.py:218).
"""

from __future__ import annotations

import math

import numpy as np
import scipy.fft
from scipy.ndimage import binary_dilation


def calculate_covering(image_shape: tuple[int, int], size: int) -> np.ndarray:
    """Stride-size/2 patch lattice, 4x coverage, in the reference's 4-sub-grid order.

    Follows regularizepsf/util.py:10-53 (un-shifted; shifted both; shifted rows;
    shifted cols; each sub-grid flattened from a meshgrid in 'xy' indexing).
    """
    half = np.ceil(size / 2).astype(int)
    sub = [
        (np.arange(0, image_shape[0], size), np.arange(0, image_shape[1], size)),
        (np.arange(-half, image_shape[0], size), np.arange(-half, image_shape[1], size)),
        (np.arange(-half, image_shape[0], size), np.arange(0, image_shape[1], size)),
        (np.arange(0, image_shape[0], size), np.arange(-half, image_shape[1], size)),
    ]
    xs, ys = [], []
    for x, y in sub:
        xg, yg = np.meshgrid(x, y)
        xs.append(xg.flatten())
        ys.append(yg.flatten())
    return np.stack([np.concatenate(xs), np.concatenate(ys)], -1)


def psf_fft(values: np.ndarray, workers: int | None = None) -> np.ndarray:
    """Un-shifted 2-D FFT over the last two axes (regularizepsf/psf.py:216-219)."""
    return scipy.fft.fft2(values, workers=workers)


def construct_transfer(source_fft: np.ndarray, target_fft: np.ndarray, alpha: float, epsilon: float) -> np.ndarray:
    """Regularized transfer kernel (regularizepsf/transform.py:78-82).

    K = conj(S) |S|^(alpha-1) / (|S|^(alpha+1) + (eps |T|)^(alpha+1)) * T, dtype-preserving.
    """
    source_abs = abs(source_fft)
    target_abs = abs(target_fft)
    numerator = source_fft.conjugate() * source_abs ** (alpha - 1)
    denominator = source_abs ** (alpha + 1) + (epsilon * target_abs) ** (alpha + 1)
    return (numerator / denominator) * target_fft


def apodization_window(n0: int, n1: int) -> np.ndarray:
    """Sine window, regularizepsf/transform.py:151-154 (meshgrid 'xy', so [j, i] indexing)."""
    row_arr, col_arr = np.meshgrid(np.arange(n0), np.arange(n1))
    return np.sin((row_arr + 0.5) * (np.pi / n0)) * np.sin((col_arr + 0.5) * (np.pi / n1))


def apply_transfer(
    image: np.ndarray,
    coordinates,
    transfer_kernel: np.ndarray,
    workers: int | None = None,
    pad_mode: str = "symmetric",
    saturation_threshold: float = math.inf,
    saturation_dilation: int = 1,
    neighborhood_width: int = 7,
) -> np.ndarray:
    """Patch-wise correction, restating regularizepsf/transform.py:116-177 step by step."""
    n0, n1 = transfer_kernel.shape[1], transfer_kernel.shape[2]
    image = image.copy().astype(float)  # :117
    padded = np.pad(image, ((2 * n0, 2 * n0), (2 * n1, 2 * n1)), mode=pad_mode)  # :119-123
    raw_padded = padded.copy()  # :126
    mask = padded > saturation_threshold  # :129
    if np.any(mask):  # :132-138
        mask = binary_dilation(mask, iterations=saturation_dilation)
        padded[mask] = np.nan
        for i, j in zip(*np.where(mask)):
            nb = (
                slice(i - neighborhood_width // 2, i + neighborhood_width // 2),
                slice(j - neighborhood_width // 2, j + neighborhood_width // 2),
            )
            padded[i, j] = np.nanmean(padded[nb])

    def sl(coord):  # :141-149
        return (
            slice(coord[0] + n0 * 2, coord[0] + n0 + n0 * 2),
            slice(coord[1] + n1 * 2, coord[1] + n1 + n1 * 2),
        )

    window = np.broadcast_to(apodization_window(n0, n1), (len(coordinates), n0, n1))  # :151-155
    patches = np.stack([padded[sl(c)[0], sl(c)[1]] for c in coordinates])  # :157-162
    patches = scipy.fft.fft2(window * patches, workers=workers)  # :163
    patches = np.real(scipy.fft.ifft2(patches * transfer_kernel, workers=workers))  # :164
    patches = patches * window  # :165
    recon = np.zeros_like(padded)  # :167
    for coord, patch in zip(coordinates, patches, strict=True):  # :168-169
        recon[sl(coord)[0], sl(coord)[1]] += patch
    recon[mask] = raw_padded[mask]  # :172
    return recon[2 * n0 : image.shape[0] + 2 * n0, 2 * n1 : image.shape[1] + 2 * n1]  # :174-177


# --------------------------------------------------------------------------------------
# Synthetic inputs of SURVEY.md section 8d (shared by tests, smoke and bench so that the
# GPU path and the CPU baseline see identical data).  Not part of the reference.
# --------------------------------------------------------------------------------------

def starfield(h: int, w: int, seed: int) -> np.ndarray:
    """100 + N(0,5) background plus round(5e-4*h*w) Gaussian (sigma 1.5) stars of amplitude 10**U(2,5)."""
    rng = np.random.default_rng(seed)
    img = 100.0 + 5.0 * rng.standard_normal((h, w))
    n_stars = int(round(5e-4 * h * w))
    ys = rng.uniform(0, h, n_stars)
    xs = rng.uniform(0, w, n_stars)
    amps = 10.0 ** rng.uniform(2, 5, n_stars)
    half = 8
    gy, gx = np.mgrid[-half : half + 1, -half : half + 1]
    for y, x, a in zip(ys, xs, amps):
        iy, ix = int(round(y)), int(round(x))
        y0, y1 = max(iy - half, 0), min(iy + half + 1, h)
        x0, x1 = max(ix - half, 0), min(ix + half + 1, w)
        if y0 >= y1 or x0 >= x1:
            continue
        sy = gy[y0 - iy + half : y1 - iy + half, x0 - ix + half : x1 - ix + half] + (iy - y)
        sx = gx[y0 - iy + half : y1 - iy + half, x0 - ix + half : x1 - ix + half] + (ix - x)
        img[y0:y1, x0:x1] += a * np.exp(-(sy**2 + sx**2) / (2 * 1.5**2))
    return img.astype(np.float32)


def gaussian_psf(n: int, sigma: float) -> np.ndarray:
    """Gaussian centred at n//2, float64, sum-normalised."""
    x = np.arange(n, dtype=float) - n // 2
    g = np.exp(-(x[:, None] ** 2 + x[None, :] ** 2) / (2 * sigma**2))
    return g / g.sum()


def coma_psf(n: int, r: int, c: int, h: int, w: int) -> np.ndarray:
    """Core sigma 1.5 plus a displaced, elongated tail pointing away from the image centre."""
    cy, cx = r + n / 2 - h / 2, c + n / 2 - w / 2
    rho = min(1.0, math.hypot(cy, cx) / math.hypot(h / 2, w / 2))
    ang = math.atan2(cy, cx) if (cy or cx) else 0.0
    d = 0.5 + 2.5 * rho
    x = np.arange(n, dtype=float) - n // 2
    yy, xx = x[:, None], x[None, :]
    core = np.exp(-(yy**2 + xx**2) / (2 * 1.5**2))
    ty, tx = yy - d * math.sin(ang), xx - d * math.cos(ang)
    along = tx * math.cos(ang) + ty * math.sin(ang)
    across = -tx * math.sin(ang) + ty * math.cos(ang)
    tail = 0.5 * np.exp(-(along**2) / (2 * (1 + d / 2) ** 2) - (across**2) / (2 * 1.5**2))
    p = core + tail
    return p / p.sum()


def synthetic_transfer(h: int, w: int, n: int, alpha: float = 3.0, epsilon: float = 0.1, kind: str = "coma"):
    """(coordinates, K complex64) for the configs of SURVEY.md 8d: coma (or Gaussian 1.8) source -> Gaussian target."""
    coords = [tuple(int(v) for v in t) for t in calculate_covering((h, w), n)]
    if kind == "coma":
        src = np.stack([coma_psf(n, r, c, h, w) for r, c in coords])
        tgt_sigma = 1.8
    else:
        src = np.broadcast_to(gaussian_psf(n, 1.8), (len(coords), n, n))
        tgt_sigma = 1.5
    tgt = gaussian_psf(n, tgt_sigma)
    s_fft = psf_fft(src, workers=-1)
    t_fft = psf_fft(tgt)[None]
    with np.errstate(all="ignore"):
        k = construct_transfer(s_fft, np.broadcast_to(t_fft, s_fft.shape), alpha, epsilon)
    if not np.isfinite(k).all():
        msg = "synthetic transfer kernel is not finite"
        raise ValueError(msg)
    return coords, k.astype(np.complex64)
