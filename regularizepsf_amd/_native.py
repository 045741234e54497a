"""ctypes binding of librpsf_hip.so (the C ABI declared in include/rpsf.h).

There is deliberately no CPU fallback: if the shared library has not been built, or no MI355X
is visible, every compute entry point raises.  Build with ``python -m regularizepsf_amd.build``
(or ``__graft_entry__.build()``).
"""

from __future__ import annotations

import ctypes
import pathlib
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

import numpy as np

import os

# RPSF_LIB points development / ablation builds at another build of the same library
LIB_PATH = pathlib.Path(os.environ.get("RPSF_LIB") or pathlib.Path(__file__).resolve().parent / "librpsf_hip.so")

PAD_MODES = {"constant": 0, "symmetric": 1, "reflect": 2, "edge": 3, "wrap": 4}
SUPPORTED_PATCH_SIZES = (16, 32, 64, 128, 256)

E_BADARG, E_UNSUPPORTED, E_HIP, E_NOMEM, E_RCCL, E_STATE = -1, -2, -3, -4, -5, -6


class NativeError(RuntimeError):
    """A call into librpsf_hip.so failed (HIP / RCCL / device-memory errors)."""

    def __init__(self, code: int, message: str) -> None:
        super().__init__(f"librpsf_hip error {code}: {message}")
        self.code = code


class Geometry(ctypes.Structure):
    """rpsf_geometry of include/rpsf.h."""

    _fields_ = [
        ("height", c_int), ("width", c_int), ("pad_mode", c_int), ("pad_value", c_float),
        ("origin_row", c_int), ("origin_col", c_int),
        ("image_row0", c_int), ("image_rows", c_int), ("ld_image", c_int),
        ("out_row0", c_int), ("out_rows", c_int), ("ld_out", c_int),
    ]

    @classmethod
    def whole(cls, height: int, width: int, pad_mode: int, pad_value: float = 0.0) -> "Geometry":
        return cls(height, width, pad_mode, pad_value, 0, 0, 0, height, width, 0, height, width)


_PROTOTYPES = {
    "rpsf_last_error": (c_char_p, []),
    "rpsf_device_count": (c_int, [POINTER(c_int)]),
    "rpsf_device_info": (c_int, [c_int, POINTER(c_int), c_char_p, c_size_t]),
    "rpsf_plan_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "rpsf_plan_destroy": (None, [c_void_p]),
    "rpsf_plan_set_transfer": (c_int, [c_void_p, c_void_p]),
    "rpsf_plan_set_transfer_device": (c_int, [c_void_p, c_void_p]),
    "rpsf_plan_set_transfer_spectra_device": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_double]),
    "rpsf_plan_transfer_bytes": (c_int, [c_void_p, POINTER(c_size_t)]),
    "rpsf_plan_set_overlap_mode": (c_int, [c_void_p, c_int]),
    "rpsf_plan_set_sweep_regions": (c_int, [c_void_p, c_int]),
    "rpsf_plan_set_option": (c_int, [c_void_p, c_int, c_int]),
    "rpsf_plan_host_bands": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "rpsf_plan_sweep_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(c_int)]),
    "rpsf_plan_set_stagger": (c_int, [c_void_p, c_int]),
    "rpsf_plan_set_image_prefetch": (c_int, [c_void_p, c_int]),
    "rpsf_plan_set_reserved_cus": (c_int, [c_void_p, c_int]),
    "rpsf_plan_set_cu_mask": (c_int, [c_void_p, c_void_p, c_int]),
    "rpsf_stream_create": (c_int, [c_int, c_void_p, c_int, POINTER(c_void_p)]),
    "rpsf_stream_destroy": (c_int, [c_void_p]),
    "rpsf_plan_debug_stamps": (c_int, [c_void_p, c_void_p, c_size_t]),
    "rpsf_apply": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "rpsf_apply_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int]),
    "rpsf_apply_host_saturated": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p, c_int]),
    "rpsf_apply_frames_host_saturated": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p, c_int]),
    "rpsf_apply_device": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(Geometry), c_void_p]),
    "rpsf_apply_device_timed": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(Geometry), c_int, c_void_p, c_void_p]),
    "rpsf_apply_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rpsf_apply_batch_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int]),
    "rpsf_apply_frames_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int]),
    "rpsf_apply_device_loop_ms": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(Geometry), c_int, POINTER(c_double)]),
    "rpsf_host_alloc": (c_int, [c_int, c_size_t, POINTER(c_void_p)]),
    "rpsf_host_free": (c_int, [c_void_p]),
    "rpsf_host_threads": (c_int, [POINTER(c_int)]),
    "rpsf_host_pool_selftest": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "rpsf_device_numa_node": (c_int, [c_int, POINTER(c_int)]),
    "rpsf_pcie_probe": (c_int, [c_int, c_size_t, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "rpsf_apply_batch_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_size_t, POINTER(Geometry),
                                        c_void_p]),
    "rpsf_apply_batch_device_timed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_size_t,
                                              POINTER(Geometry), c_int, c_void_p, c_void_p]),
    "rpsf_plan_stream": (c_void_p, [c_void_p]),
    "rpsf_build_transfer": (c_int, [c_int, c_size_t, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p]),
    "rpsf_build_transfer_device": (c_int, [c_int, c_size_t, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p,
                                           c_void_p]),
    "rpsf_psf_fft": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "rpsf_psf_fft_device": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "rpsf_psf_model_fft_device": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "rpsf_saturation_fill": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int]),
    "rpsf_dev_alloc": (c_int, [c_int, c_size_t, POINTER(c_void_p)]),
    "rpsf_dev_free": (c_int, [c_int, c_void_p]),
    "rpsf_memcpy_h2d": (c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    "rpsf_memcpy_d2h": (c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    "rpsf_device_synchronize": (c_int, [c_int]),
    "rpsf_comm_unique_id": (c_int, [c_void_p]),
    "rpsf_comm_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "rpsf_comm_destroy": (None, [c_void_p]),
    "rpsf_comm_seam_exchange_add": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "rpsf_comm_seam_exchange": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "rpsf_comm_stream": (c_void_p, [c_void_p]),
    "rpsf_comm_ranks": (c_int, [c_void_p, POINTER(ctypes.c_int)]),
    "rpsf_stream_wait": (c_int, [c_int, c_void_p, c_void_p]),
    "rpsf_event_create": (c_int, [c_int, POINTER(c_void_p)]),
    "rpsf_event_record": (c_int, [c_void_p, c_void_p]),
    "rpsf_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "rpsf_event_destroy": (c_int, [c_void_p]),
    "rpsf_add_rows": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rpsf_add_rows_narrow": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "rpsf_comm_barrier": (c_int, [c_void_p, c_void_p]),
    "rpsf_comm_allreduce_max": (c_int, [c_void_p, POINTER(c_double)]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load librpsf_hip.so once; raise ImportError (never fall back) if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            msg = (f"{LIB_PATH} not found: the HIP library has not been built. "
                   "Run `python -m regularizepsf_amd.build` (needs hipcc); there is no CPU fallback.")
            raise ImportError(msg)
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(code: int) -> None:
    if code != 0:
        text = lib().rpsf_last_error()
        raise NativeError(code, text.decode() if text else "unknown error")


def device_count() -> int:
    n = c_int(0)
    check(lib().rpsf_device_count(ctypes.byref(n)))
    return n.value


def device_info(device: int = 0) -> tuple[int, str]:
    cus = c_int(0)
    buf = ctypes.create_string_buffer(256)
    check(lib().rpsf_device_info(device, ctypes.byref(cus), buf, 256))
    return cus.value, buf.value.decode()


def _ptr(a: np.ndarray) -> c_void_p:
    return c_void_p(a.ctypes.data)


class DeviceBuffer:
    """A device allocation owned by Python (frames kept resident by bench / tests / streaming callers)."""

    def __init__(self, nbytes: int, device: int = 0) -> None:
        self.device, self.nbytes = device, int(nbytes)
        p = c_void_p()
        check(lib().rpsf_dev_alloc(device, self.nbytes, ctypes.byref(p)))
        self.ptr = p

    def upload(self, array: np.ndarray, offset_bytes: int = 0) -> "DeviceBuffer":
        a = np.ascontiguousarray(array)
        if offset_bytes + a.nbytes > self.nbytes:
            msg = "upload larger than the device buffer"
            raise ValueError(msg)
        check(lib().rpsf_memcpy_h2d(self.device, c_void_p(self.ptr.value + offset_bytes), _ptr(a), a.nbytes))
        return self

    def download(self, shape, dtype=np.float32, offset_bytes: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype)
        if offset_bytes + out.nbytes > self.nbytes:
            msg = "download larger than the device buffer"
            raise ValueError(msg)
        check(lib().rpsf_memcpy_d2h(self.device, _ptr(out), c_void_p(self.ptr.value + offset_bytes), out.nbytes))
        return out

    def at(self, offset_bytes: int) -> c_void_p:
        return c_void_p(self.ptr.value + offset_bytes)

    def free(self) -> None:
        if self.ptr is not None and self.ptr.value:
            lib().rpsf_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self) -> None:
        try:
            self.free()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class Plan:
    """Device-side form of one ArrayPSFTransform: patch size, corner list, packed transfer kernel."""

    def __init__(self, patch_size: int, coordinates, device: int = 0) -> None:
        coords = np.ascontiguousarray(np.asarray(coordinates, dtype=np.int64).reshape(-1, 2).astype(np.int32))
        self.patch_size, self.n_patches, self.device = int(patch_size), len(coords), device
        self._handle = c_void_p()
        check(lib().rpsf_plan_create(ctypes.byref(self._handle), device, self.patch_size, self.n_patches, _ptr(coords)))

    def set_transfer(self, kernel: np.ndarray) -> None:
        k = np.ascontiguousarray(kernel, dtype=np.complex64)
        if k.shape != (self.n_patches, self.patch_size, self.patch_size):
            msg = f"transfer kernel has shape {k.shape}, expected {(self.n_patches, self.patch_size, self.patch_size)}"
            raise ValueError(msg)
        check(lib().rpsf_plan_set_transfer(self._handle, _ptr(k)))

    def set_transfer_device(self, ptr: c_void_p) -> None:
        check(lib().rpsf_plan_set_transfer_device(self._handle, ptr))

    def set_transfer_spectra_device(self, s_ptr: c_void_p, t_ptr: c_void_p, alpha: float, epsilon: float) -> None:
        """Packed K straight from the two device-resident PSF spectra (construct and pack in one pass; the full K is never materialised)."""
        check(lib().rpsf_plan_set_transfer_spectra_device(self._handle, s_ptr, t_ptr, float(alpha), float(epsilon)))

    def set_overlap_mode(self, mode: str) -> None:
        """'auto' (complete lattices of 16-, 32-, 64-pixel patches: 'sweep'; other lattices - and the fallback's colour classes on a covering -
        'planes'; 'atomic' otherwise), 'atomic', 'planes', 'direct' or 'sweep'."""
        check(lib().rpsf_plan_set_overlap_mode(self._handle, {"auto": 0, "atomic": 1, "planes": 2, "direct": 3, "sweep": 4}[mode]))

    OPTIONS = {"persist": 1, "fuse": 2, "k_cached": 3, "plane_nt": 4, "host_bands": 5, "stream_group": 6, "stream_depth": 7, "debug_orphan": 8}

    def set_option(self, name: str, value: int) -> None:
        """Pin a launch option of the plan (include/rpsf.h, RPSF_OPT_*): what tests and callers may choose instead of environment variables."""
        check(lib().rpsf_plan_set_option(self._handle, self.OPTIONS[name], int(value)))

    def set_sweep_regions(self, target_regions: int) -> None:
        """Sweep kernel (N <= 64): cut the lattice into about this many regions of output pixels (default: one per compute unit)."""
        check(lib().rpsf_plan_set_sweep_regions(self._handle, int(target_regions)))

    def sweep_info(self) -> dict:
        regions, slabs = c_int(0), c_int(0)
        jobs, slots = ctypes.c_long(0), ctypes.c_long(0)
        check(lib().rpsf_plan_sweep_info(self._handle, ctypes.byref(regions), ctypes.byref(jobs), ctypes.byref(slots), ctypes.byref(slabs)))
        return {"regions": regions.value, "jobs": jobs.value, "patch_slots": slots.value, "slabs_per_phase": slabs.value}

    def host_bands(self) -> int:
        """Row bands the last single host frame was cut into (0: it went as a whole)."""
        n = c_int(0)
        check(lib().rpsf_plan_host_bands(self._handle, ctypes.byref(n)))
        return n.value

    def debug_stamps(self) -> np.ndarray:
        out = np.zeros((self.n_patches, 16), np.uint64)
        check(lib().rpsf_plan_debug_stamps(self._handle, _ptr(out), out.size))
        return out

    def set_stagger(self, microseconds: int) -> None:
        check(lib().rpsf_plan_set_stagger(self._handle, int(microseconds)))

    def set_image_prefetch(self, on: bool) -> None:
        """Opt-in for streams of new frames (256-pixel plan): the launch's head summing workgroups touch the image ahead of the gathers."""
        check(lib().rpsf_plan_set_image_prefetch(self._handle, 1 if on else 0))

    def set_reserved_cus(self, cus: int) -> None:
        """Persistent launches leave ``cus`` CUs free for kernels enqueued beside them (the RCCL seam exchange)."""
        check(lib().rpsf_plan_set_reserved_cus(self._handle, int(cus)))

    def set_cu_mask(self, mask: np.ndarray) -> None:
        """Confine the plan's stream to the compute units set in ``mask`` (uint32 words, bit i of word i // 32)."""
        m = np.ascontiguousarray(mask, dtype=np.uint32)
        check(lib().rpsf_plan_set_cu_mask(self._handle, _ptr(m), m.size))

    @property
    def transfer_bytes(self) -> int:
        n = c_size_t(0)
        check(lib().rpsf_plan_transfer_bytes(self._handle, ctypes.byref(n)))
        return n.value

    @property
    def stream(self) -> c_void_p:
        return c_void_p(lib().rpsf_plan_stream(self._handle))

    def apply(self, image: np.ndarray, pad_mode: int, pad_value: float = 0.0) -> np.ndarray:
        img = np.ascontiguousarray(image, dtype=np.float32)
        out = np.empty_like(img)
        check(lib().rpsf_apply(self._handle, _ptr(img), img.shape[0], img.shape[1], pad_mode, pad_value, _ptr(out)))
        return out

    def apply_host(self, image: np.ndarray, pad_mode: int, pad_value: float = 0.0, out_dtype=np.float64,
                   out: np.ndarray | None = None) -> np.ndarray:
        """Host array in (float32/float64 taken as they are, anything else through float32), host array out
        (float64 like the reference, or float32); dtype conversions happen inside the library.  ``out``: an
        existing C-contiguous float32/float64 array of the image's shape to write into."""
        img = np.asarray(image)
        if img.dtype not in (np.float32, np.float64) or img.dtype.byteorder == ">":
            img = img.astype(np.float32)
        img = np.ascontiguousarray(img)
        if out is not None:
            if out.shape != img.shape or not out.flags.c_contiguous or not out.flags.writeable:
                msg = "out must be a writeable C-contiguous array of the image's shape"
                raise ValueError(msg)
            out_dtype = out.dtype
        out_dtype = np.dtype(out_dtype)
        if out_dtype not in (np.float32, np.float64):
            msg = "out_dtype must be float32 or float64"
            raise ValueError(msg)
        if out is None:
            out = np.empty(img.shape, out_dtype)
        check(lib().rpsf_apply_host(self._handle, _ptr(img), int(img.dtype == np.float64), img.shape[0], img.shape[1],
                                    pad_mode, pad_value, _ptr(out), int(out_dtype == np.float64)))
        return out

    def apply_host_saturated(self, image: np.ndarray, pad_mode: int, threshold: float, dilation: int, neighborhood_width: int,
                             out_dtype=np.float64) -> np.ndarray:
        """``ArrayPSFTransform.apply`` with a finite saturation threshold in one library call (mask, dilation, sequential fill and
        restore on the host as the reference does them, the correction of the padded frame on the GPU)."""
        img = np.asarray(image)
        if img.dtype not in (np.float32, np.float64) or img.dtype.byteorder == ">":
            img = img.astype(np.float64)  # (the reference's astype(float): integer frames compare against the threshold as float64)
        img = np.ascontiguousarray(img)
        out = np.empty(img.shape, out_dtype)
        check(lib().rpsf_apply_host_saturated(self._handle, _ptr(img), int(img.dtype == np.float64), img.shape[0], img.shape[1], pad_mode,
                                              float(threshold), int(dilation), int(neighborhood_width), _ptr(out),
                                              int(np.dtype(out_dtype) == np.float64)))
        return out

    def apply_frames_host_saturated(self, images, pad_mode: int, threshold: float, dilation: int, neighborhood_width: int,
                                    out_dtype=np.float64) -> np.ndarray:
        """A sequence of equally shaped frames through the saturation branch: the host steps of frame i + 1 overlap the GPU's work on frame i."""
        frames = [np.asarray(im) for im in images]
        if not frames or any(f.ndim != 2 or f.shape != frames[0].shape for f in frames):
            msg = "frames must be two dimensional and of one shape"
            raise ValueError(msg)
        is_f32 = all(f.dtype == np.float32 for f in frames)
        want = np.float32 if is_f32 else np.float64
        frames = [np.ascontiguousarray(f if f.dtype == want and f.dtype.byteorder != ">" else f.astype(want)) for f in frames]
        n, shape = len(frames), frames[0].shape
        out = np.empty((n, *shape), out_dtype)
        in_ptrs = (c_void_p * n)(*[f.ctypes.data for f in frames])
        out_ptrs = (c_void_p * n)(*[out[i].ctypes.data for i in range(n)])
        check(lib().rpsf_apply_frames_host_saturated(self._handle, in_ptrs, int(not is_f32), n, shape[0], shape[1], pad_mode, float(threshold),
                                                     int(dilation), int(neighborhood_width), out_ptrs, int(np.dtype(out_dtype) == np.float64)))
        return out

    def apply_device(self, image_ptr: c_void_p, out_ptr: c_void_p, geometry: Geometry, stream: c_void_p | None = None) -> None:
        check(lib().rpsf_apply_device(self._handle, image_ptr, out_ptr, ctypes.byref(geometry), stream))

    def apply_device_timed(self, image_ptr: c_void_p, out_ptr: c_void_p, geometry: Geometry, iters: int):
        total = np.zeros(iters, np.float32)
        kern = np.zeros(iters, np.float32)
        check(lib().rpsf_apply_device_timed(self._handle, image_ptr, out_ptr, ctypes.byref(geometry), iters,
                                            _ptr(total), _ptr(kern)))
        return total, kern

    def apply_device_loop_ms(self, image_ptr: c_void_p, out_ptr: c_void_p, geometry: Geometry, iters: int) -> float:
        """Average device time of an apply over ``iters`` back-to-back applies (one event pair around the loop)."""
        ms = c_double(0.0)
        check(lib().rpsf_apply_device_loop_ms(self._handle, image_ptr, out_ptr, ctypes.byref(geometry), iters, ctypes.byref(ms)))
        return ms.value

    def apply_batch(self, images: np.ndarray, pad_mode: int, pad_value: float = 0.0) -> np.ndarray:
        """(frames, H, W) host stack in, float32 stack out; the frames share the installed transfer kernel (streamed)."""
        imgs = np.ascontiguousarray(images, dtype=np.float32)
        if imgs.ndim != 3:
            msg = "images must have shape (frames, H, W)"
            raise ValueError(msg)
        out = np.empty_like(imgs)
        if imgs.shape[0]:
            check(lib().rpsf_apply_batch(self._handle, _ptr(imgs), imgs.shape[0], imgs.shape[1], imgs.shape[2], pad_mode,
                                         pad_value, _ptr(out)))
        return out

    def apply_frames_host(self, images, pad_mode: int, pad_value: float = 0.0, out_dtype=np.float64,
                          out: np.ndarray | None = None) -> np.ndarray:
        """The streamed host path (rpsf_apply_frames_host): ``images`` is a (frames, H, W) array or a sequence of 2-D arrays
        of one shape; float32 / float64 frames are taken as they are (no stacking, no astype on the host side of the call),
        anything else goes through float32.  Returns (or fills ``out``, C-contiguous float32 / float64) a (frames, H, W) stack.
        H2D of the next frames, the shared-K launches and D2H + widening of the previous ones overlap on three streams."""
        if (isinstance(images, np.ndarray) and images.ndim == 3 and images.shape[0] and images.flags.c_contiguous
                and images.dtype in (np.float32, np.float64) and images.dtype.byteorder != ">"):
            # a contiguous stack: one call on the base pointers (no per-frame Python work - for 512^2 frames that was half of the time)
            out_dtype = np.dtype(out.dtype if out is not None else out_dtype)
            if out_dtype not in (np.float32, np.float64):
                msg = "out_dtype must be float32 or float64"
                raise ValueError(msg)
            if out is None:
                out = np.empty(images.shape, out_dtype)
            elif out.shape != images.shape or not out.flags.c_contiguous or not out.flags.writeable:
                msg = "out must be a writeable C-contiguous array of shape (frames, H, W)"
                raise ValueError(msg)
            check(lib().rpsf_apply_batch_host(self._handle, _ptr(images), int(images.dtype == np.float64), images.shape[0], images.shape[1],
                                              images.shape[2], pad_mode, pad_value, _ptr(out), int(out_dtype == np.float64)))
            return out
        frames = [np.asarray(im) for im in images]
        if not frames:
            msg = "need at least one frame"
            raise ValueError(msg)
        shape = frames[0].shape
        if len(shape) != 2 or any(f.shape != shape for f in frames):
            msg = "frames must be two dimensional and of one shape"
            raise ValueError(msg)
        is_f64 = all(f.dtype == np.float64 for f in frames)
        want = np.float64 if is_f64 else np.float32
        frames = [np.ascontiguousarray(f if f.dtype == want and f.dtype.byteorder != ">" else f.astype(want)) for f in frames]
        if out is not None:
            if out.shape != (len(frames), *shape) or not out.flags.c_contiguous or not out.flags.writeable:
                msg = "out must be a writeable C-contiguous array of shape (frames, H, W)"
                raise ValueError(msg)
            out_dtype = out.dtype
        out_dtype = np.dtype(out_dtype)
        if out_dtype not in (np.float32, np.float64):
            msg = "out_dtype must be float32 or float64"
            raise ValueError(msg)
        if out is None:
            out = np.empty((len(frames), *shape), out_dtype)
        n = len(frames)
        in_ptrs = (c_void_p * n)(*[f.ctypes.data for f in frames])
        out_ptrs = (c_void_p * n)(*[out[i].ctypes.data for i in range(n)])
        check(lib().rpsf_apply_frames_host(self._handle, in_ptrs, int(is_f64), n, shape[0], shape[1], pad_mode, pad_value,
                                           out_ptrs, int(out_dtype == np.float64)))
        return out

    def apply_batch_device(self, images_ptr: c_void_p, outs_ptr: c_void_p, n_frames: int, image_stride: int,
                           out_stride: int, geometry: Geometry, stream: c_void_p | None = None) -> None:
        check(lib().rpsf_apply_batch_device(self._handle, images_ptr, outs_ptr, n_frames, image_stride, out_stride,
                                            ctypes.byref(geometry), stream))

    def apply_batch_device_timed(self, images_ptr: c_void_p, outs_ptr: c_void_p, n_frames: int, image_stride: int,
                                 out_stride: int, geometry: Geometry, iters: int):
        total = np.zeros(iters, np.float32)
        kern = np.zeros(iters, np.float32)
        check(lib().rpsf_apply_batch_device_timed(self._handle, images_ptr, outs_ptr, n_frames, image_stride, out_stride,
                                                  ctypes.byref(geometry), iters, _ptr(total), _ptr(kern)))
        return total, kern

    def synchronize(self) -> None:
        check(lib().rpsf_device_synchronize(self.device))

    def close(self) -> None:
        if self._handle is not None and self._handle.value:
            lib().rpsf_plan_destroy(self._handle)
            self._handle = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def build_transfer(source_fft: np.ndarray, target_fft: np.ndarray, alpha: float, epsilon: float, device: int = 0) -> np.ndarray:
    """K2 on the GPU; dtype-preserving like the NumPy expression it replaces (transform.py:78-82)."""
    dtype = np.result_type(source_fft.dtype, target_fft.dtype)
    dtype = np.complex128 if dtype == np.complex128 else np.complex64
    s = np.ascontiguousarray(source_fft, dtype=dtype)
    t = np.ascontiguousarray(np.broadcast_to(target_fft, s.shape), dtype=dtype)
    k = np.empty_like(s)
    check(lib().rpsf_build_transfer(device, s.size, _ptr(s), _ptr(t), int(dtype == np.complex128), float(alpha),
                                    float(epsilon), _ptr(k)))
    return k


def build_transfer_device(s_ptr: c_void_p, t_ptr: c_void_p, k_ptr: c_void_p, count: int, is_f64: bool, alpha: float,
                          epsilon: float, device: int = 0) -> None:
    """K2 on device-resident spectra (count complex values each); returns once K is complete."""
    check(lib().rpsf_build_transfer_device(device, count, s_ptr, t_ptr, int(is_f64), float(alpha), float(epsilon), k_ptr,
                                           None))
    check(lib().rpsf_device_synchronize(device))


def psf_fft(values: np.ndarray, device: int = 0) -> np.ndarray:
    """K3 on the GPU: (n, N, N) real -> complex64 un-shifted 2-D spectra (psf.py:216-219)."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    if v.ndim != 3 or v.shape[1] != v.shape[2]:
        msg = "values must have shape (n, N, N)"
        raise ValueError(msg)
    out = np.empty(v.shape, np.complex64)
    if v.shape[0]:
        check(lib().rpsf_psf_fft(device, v.shape[1], v.shape[0], _ptr(v), _ptr(out)))
    return out


def psf_fft_device(values: np.ndarray, device: int = 0) -> DeviceBuffer:
    """K3 with the spectra left on the device: returns the DeviceBuffer holding (n, N, N) complex64."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    if v.ndim != 3 or v.shape[1] != v.shape[2]:
        msg = "values must have shape (n, N, N)"
        raise ValueError(msg)
    out = DeviceBuffer(max(1, v.size * 8), device)
    if v.shape[0]:
        check(lib().rpsf_psf_fft_device(device, v.shape[1], v.shape[0], _ptr(v), out.ptr))
    return out


MODEL_PARAMS = 8  # RPSF_MODEL_PARAMS
MODELS = {"elliptical_gaussian": 0, "moffat": 1}  # RPSF_MODEL_*


def psf_model_fft_device(model: str, patch_size: int, params: np.ndarray, normalize: bool = False, device: int = 0,
                         keep_values: bool = True) -> tuple[DeviceBuffer | None, DeviceBuffer]:
    """K6 + K3: rasterise a built-in parametric PSF model for every row of ``params`` (n, 8) on the device and transform the
    samples there.  Returns (float32 samples or None, complex64 spectra), both device-resident (n, N, N)."""
    q = np.ascontiguousarray(params, dtype=np.float64)
    if q.ndim != 2 or q.shape[1] != MODEL_PARAMS:
        msg = f"params must have shape (n, {MODEL_PARAMS})"
        raise ValueError(msg)
    count = q.shape[0]
    per = patch_size * patch_size
    values = DeviceBuffer(max(1, count * per * 4), device) if keep_values else None
    spectra = DeviceBuffer(max(1, count * per * 8), device)
    if count:
        check(lib().rpsf_psf_model_fft_device(device, MODELS[model], patch_size, count, _ptr(q), int(bool(normalize)),
                                              values.ptr if values is not None else None, spectra.ptr))
    return values, spectra


def pinned_empty(shape, dtype=np.float32, device: int = 0) -> np.ndarray:
    """An uninitialised array in page-locked host memory (freed with the array).  float32 frames kept in such arrays - detector
    buffers, result rings - go through ``apply`` / ``apply_batch`` without the staging copy the pageable path needs: the copy
    engines read and write them in place (for every side that needs no dtype conversion)."""
    import weakref

    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    ptr = c_void_p()
    check(lib().rpsf_host_alloc(device, max(1, nbytes), ctypes.byref(ptr)))
    buf = (ctypes.c_char * max(1, nbytes)).from_address(ptr.value)
    weakref.finalize(buf, lib().rpsf_host_free, c_void_p(ptr.value))  # runs when the last view of the memory is gone
    return np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)


def host_threads() -> int:
    """Width of the library's persistent host worker pool (dtype conversions and staging copies of the host-array entry points)."""
    n = c_int(0)
    check(lib().rpsf_host_threads(ctypes.byref(n)))
    return n.value


def device_numa_node(device: int = 0) -> int:
    """NUMA node the device hangs off (-1: unknown)."""
    n = c_int(-1)
    check(lib().rpsf_device_numa_node(device, ctypes.byref(n)))
    return n.value


def bind_to_device_node(device: int = 0) -> int:
    """Restrict the calling thread (and the threads it starts later) to the CPUs of the device's NUMA node, so that the arrays it
    allocates from now on are first touched - and therefore placed - next to the GPU.  What `numactl --cpunodebind` does for a
    one-process-per-GPU launcher; returns the node, or -1 if nothing was changed."""
    node = device_numa_node(device)
    if node < 0:
        return -1
    try:
        text = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
        cpus = set()
        for part in text.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return -1
        os.sched_setaffinity(0, cpus)
    except OSError:
        return -1
    return node


def pcie_probe(nbytes: int, iters: int = 5, device: int = 0) -> dict:
    """Milliseconds PCIe takes for ``nbytes`` from / to pinned host memory: each direction alone and both at once."""
    a, b, c = c_double(0), c_double(0), c_double(0)
    check(lib().rpsf_pcie_probe(device, int(nbytes), int(iters), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return {"h2d_ms": a.value, "d2h_ms": b.value, "duplex_ms": c.value}


def saturation_fill(padded: np.ndarray, mask: np.ndarray, neighborhood_width: int) -> None:
    """In-place sequential neighbourhood-mean fill of ``padded[mask]`` (float64, C-contiguous), transform.py:135-138."""
    if padded.dtype != np.float64 or not padded.flags.c_contiguous:
        msg = "padded must be a C-contiguous float64 array"
        raise ValueError(msg)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    check(lib().rpsf_saturation_fill(_ptr(padded), padded.shape[0], padded.shape[1], _ptr(m), int(neighborhood_width)))


def add_rows(accum_ptr: c_void_p, src_ptr: c_void_p, count: int, device: int = 0, stream: c_void_p | None = None,
             max_workgroups: int = 0) -> None:
    """accum[0:count] += src[0:count] on the device (kernel K4: the add of the seam exchange); ``max_workgroups`` > 0: on at
    most that many workgroups (for running it beside a persistent patch launch)."""
    if max_workgroups > 0:
        check(lib().rpsf_add_rows_narrow(device, accum_ptr, src_ptr, count, max_workgroups, stream))
    else:
        check(lib().rpsf_add_rows(device, accum_ptr, src_ptr, count, stream))


def stream_wait(waiter: c_void_p, signaller: c_void_p, device: int = 0) -> None:
    """Work enqueued on ``waiter`` from now on starts after everything enqueued on ``signaller`` so far."""
    check(lib().rpsf_stream_wait(device, waiter, signaller))


def cu_masks(compute_units: int, carve: int, layout: str | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Two complementary CU masks over ``compute_units`` bits: (everything but the carved bits, the carved bits).  ``layout``:
    "spread" - the carved CUs evenly spaced over the bit range (one per 32-bit word for 8 of 256: one per XCD if the numbering
    is XCD-major); "tail" - the last ``carve`` bits (one per XCD if it is round-robin)."""
    layout = layout or os.environ.get("RPSF_CU_MASK_LAYOUT", "tail")
    words = (compute_units + 31) // 32
    big, small = np.zeros(words, np.uint32), np.zeros(words, np.uint32)
    if layout == "tail":
        carved = set(range(compute_units - carve, compute_units))
    else:
        carved = {(k + 1) * compute_units // carve - 1 for k in range(carve)}
    for i in range(compute_units):
        (small if i in carved else big)[i // 32] |= np.uint32(1 << (i % 32))
    return big, small


class Stream:
    """A HIP stream of the caller's own, optionally confined to the compute units of a mask."""

    def __init__(self, device: int = 0, mask: np.ndarray | None = None) -> None:
        self._handle = c_void_p()
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint32)
        check(lib().rpsf_stream_create(device, None if m is None else _ptr(m), 0 if m is None else m.size, ctypes.byref(self._handle)))

    @property
    def ptr(self) -> c_void_p:
        return self._handle

    def close(self) -> None:
        if self._handle is not None and self._handle.value:
            lib().rpsf_stream_destroy(self._handle)
            self._handle = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Event:
    """A HIP event (no timing): ``record(stream)`` now, ``make_wait(stream)`` later."""

    def __init__(self, device: int = 0) -> None:
        self._handle = c_void_p()
        check(lib().rpsf_event_create(device, ctypes.byref(self._handle)))
        self.recorded = False

    def record(self, stream: c_void_p) -> None:
        check(lib().rpsf_event_record(self._handle, stream))
        self.recorded = True

    def make_wait(self, stream: c_void_p) -> None:
        if self.recorded:
            check(lib().rpsf_stream_wait_event(stream, self._handle))

    def close(self) -> None:
        if self._handle is not None and self._handle.value:
            lib().rpsf_event_destroy(self._handle)
            self._handle = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Comm:
    """RCCL neighbour exchange for the row-band split (one process per GPU)."""

    def __init__(self, device: int, rank: int, world: int, unique_id: bytes) -> None:
        self.rank, self.world, self.device = rank, world, device
        self._handle = c_void_p()
        buf = ctypes.create_string_buffer(unique_id, 128)
        check(lib().rpsf_comm_create(ctypes.byref(self._handle), device, rank, world, buf))

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        check(lib().rpsf_comm_unique_id(buf))
        return buf.raw

    def seam_exchange_add(self, send_ptr, send_count: int, recv_ptr, recv_count: int, accum_ptr, stream=None) -> None:
        check(lib().rpsf_comm_seam_exchange_add(self._handle, send_ptr, send_count, recv_ptr, recv_count, accum_ptr, stream))

    def seam_exchange(self, send_ptr, send_count: int, recv_ptr, recv_count: int, stream=None) -> None:
        """Send to rank + 1 / receive from rank - 1 without the add (asynchronous on ``stream``, default: the communicator's)."""
        check(lib().rpsf_comm_seam_exchange(self._handle, send_ptr, send_count, recv_ptr, recv_count, stream))

    @property
    def stream(self) -> c_void_p:
        return c_void_p(lib().rpsf_comm_stream(self._handle))

    def barrier(self, stream=None) -> None:
        check(lib().rpsf_comm_barrier(self._handle, stream))

    def ranks(self) -> int:
        """The rank count RCCL itself reports for this communicator (ncclCommCount)."""
        n = ctypes.c_int(0)
        check(lib().rpsf_comm_ranks(self._handle, ctypes.byref(n)))
        return int(n.value)

    def allreduce_max(self, value: float) -> float:
        v = c_double(value)
        check(lib().rpsf_comm_allreduce_max(self._handle, ctypes.byref(v)))
        return v.value

    def close(self) -> None:
        if self._handle is not None and self._handle.value:
            lib().rpsf_comm_destroy(self._handle)
            self._handle = None
