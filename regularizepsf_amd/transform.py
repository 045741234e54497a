"""PSF-to-PSF transform applied patch-wise in the Fourier domain (API of regularizepsf/transform.py:25-177,284-289).

``construct`` evaluates the regularized transfer kernel with the K2 HIP kernel and ``apply`` runs the
fused K1 patch kernel; both go through the C ABI of include/rpsf.h.  There is no CPU fallback.
Differences from the reference that a caller can observe are listed in INTEGRATION.md (float32
arithmetic inside the kernels; patch sizes other than 16..256 powers of two run a hipFFT-based fallback).
"""

from __future__ import annotations

import hashlib
import math
import numbers
import pathlib
import threading
import weakref

import numpy as np

from regularizepsf_amd import _native
from regularizepsf_amd.exceptions import InvalidCoordinateError
from regularizepsf_amd.util import IndexedCube


def _kernel_stamp(cube: IndexedCube) -> tuple:
    """What the device copy of a transfer kernel was made from: identity, edit count and a fingerprint of the values.

    The reference multiplies by ``values`` as they are at every ``apply`` (transform.py:164), and ``IndexedCube`` does not
    copy them, so a caller may edit the array in place between two applies.  ``__setitem__`` is counted exactly; edits that
    bypass it (``cube.values[i] = ...``, ``k *= 2``) cannot go unnoticed since round 6: the array is made READ-ONLY while a device copy
    of it exists (``ArrayPSFTransform._freeze``), so such a write raises NumPy's ``ValueError: assignment destination is read-only``
    instead of being silently missed; ``with transform.edit() as k:``, ``cube[coordinate] = patch`` and ``transform.invalidate()``
    are the ways to edit.  The fingerprint below stays as the backstop for memory that is reachable through another, still writeable
    array (a view taken before the upload, a base array): one 64-byte line of every patch (at most 4096 patches, evenly spaced beyond
    that; the line's position inside the patch varies from patch to patch) - 20 ... 60 us, where the evenly strided 64 K-element sample
    of rounds 2-4 cost 0.3 ... 1.4 ms of cache misses per apply, more than a small frame's whole correction.
    """
    if getattr(cube, "_loader", None) is not None:  # still on the GPU only (construct from device-resident spectra): nobody can have edited it
        return ("deferred", id(cube), cube._edits)
    values = cube.values
    if values.ndim == 3 and values.flags.c_contiguous and values.itemsize in (8, 16) and values.size:
        words = values.reshape(-1).view(np.uint64)  # 1 (complex64) or 2 (complex128) words per element
        key = (values.shape, values.itemsize)
        cached = getattr(cube, "_stamp_index", None)
        if cached is None or cached[0] != key:  # (the sample's positions depend on the shape only: built once per cube, 10 us of every apply otherwise)
            wpp = values.shape[1] * values.shape[2] * (values.itemsize // 8)
            patches = np.arange(0, values.shape[0], max(1, values.shape[0] // 4096), dtype=np.int64)
            line = min(8, wpp)
            offset = (patches * 104729) % max(1, wpp - line + 1)
            cached = (key, ((patches * wpp + offset)[:, None] + np.arange(line, dtype=np.int64)[None, :]).reshape(-1))
            cube._stamp_index = cached
        sample = words[cached[1]]
        # (two wrap-around reductions of the sampled words instead of a cryptographic digest of their bytes: any change of one word changes both)
        return (id(values), cube._edits, values.shape, values.dtype.str, int(np.bitwise_xor.reduce(sample)), int(np.add.reduce(sample)))
    elif values.flags.c_contiguous or values.flags.f_contiguous:
        flat = values.reshape(-1, order="A")
        sample = flat[:: max(1, flat.size // 4096)]
    else:  # strided views are rare and small: hash everything
        sample = np.ascontiguousarray(values)
    digest = hashlib.blake2b(sample.tobytes(), digest_size=8).digest()
    return (id(values), cube._edits, values.shape, values.dtype.str, digest)


class ArrayPSFTransform:
    """Transformation from a source PSF to a target PSF that can be applied to images."""

    def __init__(self, transfer_kernel: IndexedCube, device: int = 0) -> None:
        self._transfer_kernel = transfer_kernel
        self._device = device
        self._plan: _native.Plan | None = None
        self._plan_stamp: tuple | None = None
        self._corner_bounds: tuple | None = None
        # one device plan (staging buffers, colour planes, one stream) per transform: calls from several threads take turns
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ accessors (transform.py:37-51)
    @property
    def psf_shape(self) -> tuple[int, int]:
        return self._transfer_kernel.sample_shape

    @property
    def coordinates(self) -> list[tuple[int, int]]:
        return self._transfer_kernel.coordinates

    def __len__(self) -> int:
        return len(self._transfer_kernel)

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, ArrayPSFTransform):
            msg = "Can only compare ArrayPSFTransform to another ArrayPSFTransform."
            raise TypeError(msg)
        return self._transfer_kernel == other._transfer_kernel

    __hash__ = None

    # ------------------------------------------------------------------ construct (transform.py:53-83)
    @classmethod
    def construct(cls, source, target, alpha: float, epsilon: float, device: int = 0) -> "ArrayPSFTransform":
        """Build the transform taking ``source`` to ``target``.

        ``alpha`` controls the hardness of the transition from amplification to attenuation and
        ``epsilon`` the maximum amplification.  Raises :class:`InvalidCoordinateError` when the two
        models are not sampled at the same coordinates (transform.py:74-76).  The kernel keeps the
        dtype of the PSF spectra (complex64 for float32 PSFs, complex128 for float64 ones).
        """
        if np.any(np.array(source.coordinates) != np.array(target.coordinates)):
            msg = "Source PSF coordinates do not match target PSF coordinates."
            raise InvalidCoordinateError(msg)
        n_patch = source.sample_shape[0] if len(source) else 0
        dev_s, dev_t = getattr(source, "_fft_dev", None), getattr(target, "_fft_dev", None)

        def pristine(psf) -> bool:
            # the device copy of the spectra is what the reference would use only while nobody has fetched the cube (a fetched
            # array can be edited in place: transform.py:78-82 reads fft_evaluations as they are at the time of the call)
            cube = getattr(psf, "_fft_cube", None)
            return cube is not None and getattr(cube, "_loader", None) is not None and cube._edits == 0

        if (dev_s is not None and dev_t is not None and pristine(source) and pristine(target)
                and dev_s[1] == dev_t[1] == device and len(source) == len(target) > 0
                and source.sample_shape == target.sample_shape and n_patch in _native.SUPPORTED_PATCH_SIZES
                and all(isinstance(v, numbers.Integral) for c in source.coordinates for v in c)):
            # Both spectra were computed on this GPU (ArrayPSF(device=...)) and are still there: K2 -> pack -> plan without
            # anything crossing PCIe; the IndexedCube downloads K the first time somebody looks at its values.
            # spectra -> packed K in ONE pass (the formula is evaluated where the packer reads K): the full K - 571 MB at 1089 patches of 256
            # pixels - is neither written nor read back; it is built, from the same resident spectra, only if somebody looks at the values.
            count = len(source) * n_patch * n_patch
            plan = _native.Plan(n_patch, source.coordinates, device=device)
            plan.set_transfer_spectra_device(dev_s[0].ptr, dev_t[0].ptr, alpha, epsilon)
            shape = (len(source), n_patch, n_patch)

            def fetch(bufs=(dev_s[0], dev_t[0]), shape=shape, count=count):  # (holds the two spectra alive)
                kbuf = _native.DeviceBuffer(count * 8, device)
                try:
                    _native.build_transfer_device(bufs[0].ptr, bufs[1].ptr, kbuf.ptr, count, False, alpha, epsilon, device)
                    return kbuf.download(shape, np.complex64)
                finally:
                    kbuf.free()

            cube = IndexedCube._deferred(source.coordinates, shape, fetch)
            out = cls(cube, device=device)
            out._plan, out._plan_stamp = plan, _kernel_stamp(cube)
            ref = weakref.ref(out)

            def fetched(c, stamp=out._plan_stamp):
                # looking at K must not cost a re-upload: the values that just arrived ARE the device copy
                t = ref()
                if t is not None and t._plan_stamp == stamp:
                    t._plan_stamp = _kernel_stamp(c)

            cube._load_hook = fetched
            return out
        s_fft, t_fft = source.fft_evaluations, target.fft_evaluations
        resident = (np.result_type(s_fft.dtype, t_fft.dtype) == np.complex64 and s_fft.ndim == 3 and len(source) > 0
                    and s_fft.shape == t_fft.shape and s_fft.shape[1] == s_fft.shape[2]
                    and s_fft.shape[1] in _native.SUPPORTED_PATCH_SIZES
                    and all(isinstance(v, numbers.Integral) for c in source.coordinates for v in c))
        if not resident:
            kernel = _native.build_transfer(s_fft, t_fft, alpha, epsilon, device=device)
            return cls(IndexedCube(source.coordinates, kernel), device=device)
        # complex64 spectra and a patch size the kernels support: K is built on the device, packed for K1 from
        # there (no second trip over PCIe at the first apply) and copied back once for the IndexedCube.
        s_fft = np.ascontiguousarray(s_fft, dtype=np.complex64)
        t_fft = np.ascontiguousarray(t_fft, dtype=np.complex64)
        bufs = [_native.DeviceBuffer(s_fft.nbytes, device) for _ in range(3)]
        try:
            bufs[0].upload(s_fft)
            bufs[1].upload(t_fft)
            _native.build_transfer_device(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, s_fft.size, False, alpha, epsilon, device)
            plan = _native.Plan(s_fft.shape[1], source.coordinates, device=device)
            plan.set_transfer_device(bufs[2].ptr)
            plan.synchronize()
            kernel = bufs[2].download(s_fft.shape, np.complex64)
        finally:
            for b in bufs:
                b.free()
        cube = IndexedCube(source.coordinates, kernel)
        out = cls(cube, device=device)
        out._plan, out._plan_stamp = plan, _kernel_stamp(cube)
        return out

    # ------------------------------------------------------------------ device plan
    def invalidate(self) -> None:
        """Drop the device copy of the transfer kernel; ``values`` becomes writeable again (it was made read-only when the copy was
        taken).  The next ``apply`` uploads whatever the array holds then - the reference reads ``values`` at every apply (transform.py:164)."""
        with self._lock:
            if self._plan is not None:
                self._plan.close()
            self._plan, self._plan_stamp = None, None
            self._thaw()

    def _freeze(self) -> None:
        """A device copy of ``values`` exists: make the array read-only, so that an in-place edit raises instead of leaving the copy stale."""
        cube = self._transfer_kernel
        values = cube._values_array
        if values is not None and values.flags.writeable:
            try:
                values.flags.writeable = False
                cube._frozen = True
            except ValueError:  # (an array that does not own its memory and whose base forbids it: the fingerprint remains)
                pass

    def _thaw(self) -> None:
        cube = self._transfer_kernel
        if getattr(cube, "_frozen", False) and cube._values_array is not None:
            cube._values_array.flags.writeable = True
            cube._frozen = False

    def edit(self):
        """``with transform.edit() as k: k[i, r, c] = x`` - the transfer kernel's array, writeable inside the block; the device copy is
        refreshed at the next ``apply`` (every edit made this way is seen, however small)."""
        import contextlib

        @contextlib.contextmanager
        def editing():
            with self._lock:
                cube = self._transfer_kernel
                values = cube.values
                self._thaw()
                try:
                    yield values
                finally:
                    cube._edits += 1  # the stamp changes: the next apply uploads the array again (and makes it read-only again)

        return editing()

    def _checked_patch_size(self) -> int:
        n0, n1 = self.psf_shape
        if n0 != n1:
            msg = f"operands could not be broadcast together: PSF samples must be square, got {self.psf_shape}"
            raise ValueError(msg)
        if not 2 <= n0 <= 4096:  # 16..256 (powers of two) run the hand-written kernels, the rest the hipFFT fallback
            msg = f"patch size {n0} is outside the supported range 2..4096"
            raise NotImplementedError(msg)
        return n0

    def _device_plan(self) -> _native.Plan:
        cube = self._transfer_kernel
        stamp = _kernel_stamp(cube)
        if self._plan is None or self._plan_stamp != stamp:
            self.invalidate()
            n = self._checked_patch_size()
            for coordinate in cube.coordinates:  # the reference slices with these (transform.py:141-149)
                if not all(isinstance(v, numbers.Integral) for v in coordinate):
                    msg = "slice indices must be integers: patch coordinates must be integral"
                    raise TypeError(msg)
            plan = _native.Plan(n, cube.coordinates, device=self._device)
            plan.set_transfer(cube.values)
            self._plan, self._plan_stamp = plan, stamp
        self._freeze()
        return self._plan

    def _check_corners(self, n: int, height: int, width: int) -> None:
        """Outside the 2N pad the reference's np.stack is ragged -> ValueError (transform.py:141-162)."""
        coords = self.coordinates
        if self._corner_bounds is None or self._corner_bounds[0] != len(coords):  # extremes, once per corner list
            arr = np.asarray(coords).reshape(-1, 2)
            self._corner_bounds = (len(coords), arr[:, 0].min(), arr[:, 0].max(), arr[:, 1].min(), arr[:, 1].max())
        _, rlo, rhi, clo, chi = self._corner_bounds
        if rlo < -2 * n or rhi > height + n or clo < -2 * n or chi > width + n:
            bad = next((r, c) for r, c in coords if r < -2 * n or r > height + n or c < -2 * n or c > width + n)
            msg = f"patch corner {bad} lies outside the padded image"
            raise ValueError(msg)

    # ------------------------------------------------------------------ apply (transform.py:85-177)
    def apply(self, image: np.ndarray, workers: int | None = None, pad_mode: str = "symmetric",
              saturation_threshold: float = math.inf, saturation_dilation: int = 1,
              neighborhood_width: int = 7, *, out: np.ndarray | None = None) -> np.ndarray:
        """Apply the transform to an image and return the corrected image (float64, same shape).

        Parameters follow the reference: ``pad_mode`` is any ``np.pad`` mode; pixels brighter than
        ``saturation_threshold`` are replaced by their neighbourhood mean before correction and
        restored afterwards.  ``workers`` is accepted for compatibility and ignored (the FFTs run
        on the GPU).  The input is never modified.

        ``out`` (keyword only, not in the reference): a C-contiguous float64 or float32 array of the image's shape to write the result
        into and return - a caller's loop that reuses one result array spares every call the first touch of two megabytes of fresh pages
        per 512 x 512 frame, which costs anything between 10 and 200 us on a busy host (`scripts/notebook_factors.py`).
        """
        del workers
        with self._lock:
            result = self._apply_locked(image, pad_mode, saturation_threshold, saturation_dilation, neighborhood_width, out)
        if out is not None and result is not out:  # (the branches that build their result on the host)
            if out.shape != result.shape:
                msg = "out must have the image's shape"
                raise ValueError(msg)
            out[...] = result
            return out
        return result

    def _apply_locked(self, image, pad_mode, saturation_threshold, saturation_dilation, neighborhood_width, out=None) -> np.ndarray:
        image = np.asarray(image)
        if image.ndim != 2:
            msg = f"image must be two dimensional, got shape {image.shape}"
            raise ValueError(msg)
        if len(self) == 0:  # the reference fails in np.stack([]) (transform.py:157-162)
            msg = "need at least one array to stack"
            raise ValueError(msg)
        n = self._checked_patch_size()
        plan = self._device_plan()
        height, width = image.shape
        self._check_corners(n, height, width)

        if saturation_threshold == math.inf and pad_mode in _native.PAD_MODES:  # nothing can exceed +inf
            return plan.apply_host(image, _native.PAD_MODES[pad_mode], out=out)  # float64 out (or the caller's array); conversions inside the library

        if (pad_mode in _native.PAD_MODES and isinstance(saturation_dilation, numbers.Integral) and saturation_dilation >= 1
                and isinstance(neighborhood_width, numbers.Integral) and neighborhood_width >= 0):
            # the saturation branch in one library call: the reference's host steps (pad, mask, dilation, sequential fill, restore) on the
            # plan's own scratch, the correction of the padded frame on the GPU (rpsf_apply_host_saturated)
            return plan.apply_host_saturated(image, _native.PAD_MODES[pad_mode], saturation_threshold, saturation_dilation,
                                             neighborhood_width)

        # Host-side padding: np.pad modes the kernel does not evaluate, and the corner cases of the saturation branch (dilation < 1: scipy
        # iterates to a fixed point; negative widths), which work on the padded image (transform.py:119-138,171-172).
        padded = np.pad(image.astype(float), ((2 * n, 2 * n), (2 * n, 2 * n)), mode=pad_mode)
        raw = padded.copy()
        mask = padded > saturation_threshold
        if np.any(mask):
            from scipy.ndimage import binary_dilation

            mask = binary_dilation(mask, iterations=saturation_dilation)
            padded[mask] = np.nan
            if neighborhood_width >= 0:
                # the reference's per-pixel np.nanmean loop, restated in C (same order, same window rule)
                _native.saturation_fill(padded, mask, neighborhood_width)
            else:  # negative widths: keep NumPy's own slice arithmetic
                half = neighborhood_width // 2
                with np.errstate(all="ignore"):
                    for i, j in zip(*np.where(mask)):
                        padded[i, j] = np.nanmean(padded[i - half : i + half, j - half : j + half])
        corrected = self._apply_prepadded(padded.astype(np.float32), 2 * n)
        corrected = corrected.astype(np.float64)
        corrected[mask] = raw[mask]
        return corrected[2 * n : height + 2 * n, 2 * n : width + 2 * n]

    def _apply_prepadded(self, padded: np.ndarray, shift: int) -> np.ndarray:
        """Run K1 on an image the host already padded by ``shift`` on every side (constant mode, shifted corners)."""
        plan = self._device_plan()
        h, w = padded.shape
        img = _native.DeviceBuffer(padded.nbytes, self._device).upload(padded)
        out = _native.DeviceBuffer(padded.nbytes, self._device)
        try:
            geom = _native.Geometry.whole(h, w, _native.PAD_MODES["constant"], 0.0)
            geom.origin_row = geom.origin_col = shift
            plan.apply_device(img.ptr, out.ptr, geom)
            plan.synchronize()
            return out.download((h, w))
        finally:
            img.free()
            out.free()

    #: pre-1.0 name of :meth:`apply` (the task description refers to it)
    correct_image = apply

    def apply_batch(self, images, workers: int | None = None, pad_mode: str = "symmetric",
                    saturation_threshold: float = math.inf, saturation_dilation: int = 1,
                    neighborhood_width: int = 7, dtype: type = np.float64, out: np.ndarray | None = None) -> np.ndarray:
        """Apply the transform to a stack ``(frames, H, W)`` or a sequence of equally shaped images; returns a stack.

        Equivalent to ``np.stack([self.apply(im, ...) for im in images])`` - the loop a user of the reference writes
        (docs/source/example.ipynb over transform.py:85-177) - and bit-identical to it, but streamed: while one group
        of frames is corrected, the next one crosses PCIe to the device and the previous one comes back and is widened to
        float64, on three streams and a persistent pool of host threads (``rpsf_apply_frames_host``).  Frames that already
        live on the GPU should use ``_native.Plan.apply_batch_device``.  ``dtype`` is the result dtype (the reference
        returns float64; ``np.float32`` skips the widening); ``out`` an existing C-contiguous stack to fill.
        Extension of the reference API - there is no ``apply_batch`` upstream.
        """
        with self._lock:
            return self._apply_batch_locked(images, workers, pad_mode, saturation_threshold, saturation_dilation,
                                            neighborhood_width, dtype, out)

    def _apply_batch_locked(self, images, workers, pad_mode, saturation_threshold, saturation_dilation, neighborhood_width,
                            dtype, out) -> np.ndarray:
        stack = None
        if isinstance(images, np.ndarray):
            if images.ndim != 3:
                msg = f"images must have shape (frames, H, W), got {images.shape}"
                raise ValueError(msg)
            frames = images  # (iterated only on the per-frame routes below)
            stack = images
            shape = images.shape[1:]
        else:
            frames = [np.asarray(im) for im in images]
            shape = frames[0].shape if frames else (0, 0)
            if any(f.ndim != 2 or f.shape != shape for f in frames):
                msg = "images must be a sequence of two dimensional arrays of one shape"
                raise ValueError(msg)
        dtype = np.dtype(out.dtype if out is not None else dtype)
        if (saturation_threshold != math.inf and pad_mode in _native.PAD_MODES and len(frames) > 0 and len(self) > 0
                and dtype in (np.float32, np.float64) and isinstance(saturation_dilation, numbers.Integral) and saturation_dilation >= 1
                and isinstance(neighborhood_width, numbers.Integral) and neighborhood_width >= 0):
            # the saturation branch for a sequence of frames: the host steps of frame i + 1 (pad, mask, dilation, sequential fill) run while the
            # GPU corrects frame i (rpsf_apply_frames_host_saturated)
            n = self._checked_patch_size()
            plan = self._device_plan()
            self._check_corners(n, *shape)
            res = plan.apply_frames_host_saturated(frames, _native.PAD_MODES[pad_mode], saturation_threshold, saturation_dilation,
                                                   neighborhood_width, out_dtype=dtype)
            if out is not None:
                out[...] = res
                return out
            return res
        if (saturation_threshold != math.inf or pad_mode not in _native.PAD_MODES or len(frames) == 0
                or dtype not in (np.float32, np.float64)):
            outs = [self.apply(im, workers, pad_mode, saturation_threshold, saturation_dilation, neighborhood_width)
                    for im in frames]
            res = np.stack(outs).astype(dtype, copy=False) if outs else np.empty((0, *shape), dtype)
            if out is not None:
                out[...] = res
                return out
            return res
        if len(self) == 0:
            msg = "need at least one array to stack"
            raise ValueError(msg)
        n = self._checked_patch_size()
        plan = self._device_plan()
        height, width = shape
        self._check_corners(n, height, width)
        return plan.apply_frames_host(stack if stack is not None else frames, _native.PAD_MODES[pad_mode], out_dtype=dtype, out=out)

    # ------------------------------------------------------------------ persistence (transform.py:220-282)
    def save(self, path: pathlib.Path, overwrite: bool = False) -> None:
        """Save to ``.h5``: datasets ``coordinates`` (n, 2) int64 and ``transfer_kernel`` (n, N, N) complex, exactly
        what the reference writes (``transform.py:236-240``), so either package loads the other's files.
        ``overwrite=False`` raises ``FileExistsError`` for an existing file (h5py mode ``"w-"`` upstream)."""
        path = pathlib.Path(path)
        if path.suffix == ".h5":
            from regularizepsf_amd import _h5min

            _h5min.write_datasets(path, {"coordinates": np.array(self.coordinates, dtype=np.int64).reshape(-1, 2),
                                         "transfer_kernel": self._transfer_kernel.values}, overwrite=overwrite)
        elif path.suffix == ".fits":
            msg = "FITS persistence is not implemented in this package (lossy CompImageHDU semantics live in astropy; see DESIGN.md)"
            raise NotImplementedError(msg)
        else:
            msg = f"Unsupported file type {path.suffix}. Change to .h5 or .fits."
            raise NotImplementedError(msg)

    @classmethod
    def load(cls, path: pathlib.Path, device: int = 0) -> "ArrayPSFTransform":
        """Load a transform saved by this package or by the reference (``transform.py:266-270``)."""
        path = pathlib.Path(path)
        if path.suffix == ".h5":
            from regularizepsf_amd import _h5min

            data = _h5min.read_datasets(path, ["coordinates", "transfer_kernel"])
            coordinates = [tuple(int(v) for v in c) for c in data["coordinates"]]
            return cls(IndexedCube(coordinates, data["transfer_kernel"]), device=device)
        if path.suffix == ".fits":
            msg = "FITS persistence is not implemented in this package (lossy CompImageHDU semantics live in astropy; see DESIGN.md)"
            raise NotImplementedError(msg)
        msg = f"Unsupported file type {path.suffix}. Change to .h5 or .fits."
        raise NotImplementedError(msg)
