"""3-D real FFT for the cross-correlation of ``dynatrack._phase_cross_corr``, axis by axis.

``torch.fft.rfftn`` / ``irfftn`` of the tracker's grid (180 x 2048 x 2304) take 13.6 / 16.9 ms on an
MI355X, of which the butterfly kernels are 4.4 ms: the rest is rocFFT's own transposes between the
axes, its real <-> complex pre / post passes, and a device-to-device clone of the input that PyTorch
makes before every rocFFT call on ROCm plus a separate normalisation kernel on the way back
(``tools/probes/fft_split.py``, ``fft_calls.py``; ``profiles/r02_secondary_kernel_stats.csv``).  Here
the same library does only what it is fast at -- one contiguous, batched, in-place 1-D transform per
axis, called through hipFFT's C API (the copy PyTorch ships and has already loaded: no second FFT
library enters the process) -- and the layout changes between the axes are this package's transpose
kernel (``lsr_transpose_last2_c64``).  The spectrum stays in the transposed layout ``[XC][Y][Z]``
(``XC = X // 2 + 1``): the cross power is element-wise and does not care, and the way back undoes it.
No normalisation either way: the consumer takes an argmax.

The long y axis stays a library call (rocFFT).  The short z leg (``lsr_cross_correlate_z_c64``) and the
x leg each way (``lsr_rfft_rows_t_c64``, ``lsr_irfft_rows_peak``) are this package's own LDS-resident
transforms, fused with the steps around them -- see :func:`correlation_peak`; :func:`rfft3` /
:func:`irfft3` / :func:`correlate_with_spectrum` are the routes for grids those kernels do not take.
"""

from __future__ import annotations

import ctypes
import os

from . import _lib

__all__ = ["available", "rfft3", "irfft3", "correlate_with_spectrum", "rows_supported", "spectrum_of", "correlation_peak",
           "AxisFftError"]

_HIPFFT_R2C, _HIPFFT_C2R, _HIPFFT_C2C = 0x2A, 0x2C, 0x29
_FORWARD, _BACKWARD = -1, 1


class AxisFftError(RuntimeError):
    """hipFFT refused a plan or an execution (the caller falls back to ``torch.fft``)."""


_hipfft = None
_hipfft_tried = False


def _lib_hipfft():
    global _hipfft, _hipfft_tried
    if not _hipfft_tried:
        _hipfft_tried = True
        if os.environ.get("LSR_AXIS_FFT", "1") in ("0", "off", "false"):
            return None
        try:
            import torch

            path = os.path.join(os.path.dirname(torch.__file__), "lib", "libhipfft.so")
            lib = ctypes.CDLL(path)
            lib.hipfftCreate.restype = ctypes.c_int
            lib.hipfftCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
            lib.hipfftSetAutoAllocation.restype = ctypes.c_int
            lib.hipfftSetAutoAllocation.argtypes = [ctypes.c_void_p, ctypes.c_int]
            lib.hipfftMakePlanMany.restype = ctypes.c_int
            lib.hipfftMakePlanMany.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
            lib.hipfftSetWorkArea.restype = ctypes.c_int
            lib.hipfftSetWorkArea.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            lib.hipfftSetStream.restype = ctypes.c_int
            lib.hipfftSetStream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            lib.hipfftExecC2C.restype = ctypes.c_int
            lib.hipfftExecC2C.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            lib.hipfftExecR2C.restype = ctypes.c_int
            lib.hipfftExecR2C.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            lib.hipfftExecC2R.restype = ctypes.c_int
            lib.hipfftExecC2R.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            lib.hipfftDestroy.restype = ctypes.c_int
            lib.hipfftDestroy.argtypes = [ctypes.c_void_p]
            _hipfft = lib
        except (OSError, AttributeError, ImportError):
            _hipfft = None
    return _hipfft


def available() -> bool:
    """hipFFT's C API is loadable from the PyTorch installation (``LSR_AXIS_FFT=0`` switches this path off)."""
    return _lib_hipfft() is not None


# (device index, kind, n, batch) -> (plan handle, work-area bytes); a handful of plans per grid shape,
# dropped oldest first.  Plans do not own device memory: hipFFT's automatic work-area allocation is off
# (the real-to-complex plan of the tracker's grid wants 3.4 GB) and every execution borrows the work
# area from PyTorch's caching allocator for the duration of the call.
_plans: dict = {}
_MAX_PLANS = 24


def _plan(device, kind: int, n: int, batch: int):
    lib = _lib_hipfft()
    key = (device.index, kind, int(n), int(batch))
    plan = _plans.get(key)
    if plan is None:
        if batch >= 2 ** 31 or n >= 2 ** 31:
            raise AxisFftError(f"transform of length {n} x batch {batch} exceeds hipFFT's int arguments")
        while len(_plans) >= _MAX_PLANS:
            lib.hipfftDestroy(_plans.pop(next(iter(_plans)))[0])
        handle = ctypes.c_void_p()
        rc = lib.hipfftCreate(ctypes.byref(handle))
        if rc == 0:
            rc = lib.hipfftSetAutoAllocation(handle, 0)
        work = ctypes.c_size_t(0)
        if rc == 0:
            dims = (ctypes.c_int * 1)(int(n))
            # NULL embeds: contiguous sequences, distance n (C2C), n -> n // 2 + 1 (R2C), n // 2 + 1 -> n (C2R)
            rc = lib.hipfftMakePlanMany(handle, 1, dims, None, 1, 0, None, 1, 0, kind, int(batch), ctypes.byref(work))
        if rc != 0:
            if handle:
                lib.hipfftDestroy(handle)
            raise AxisFftError(f"hipFFT plan (n={n}, batch={batch}, type={kind:#x}) failed with status {rc}")
        plan = _plans[key] = (handle, int(work.value))
    else:
        _plans[key] = _plans.pop(key)        # most recently used last
    return plan


def _exec(device, kind: int, n: int, batch: int, src_ptr: int, dst_ptr: int, direction: int = _FORWARD) -> None:
    import torch

    lib = _lib_hipfft()
    plan, work_bytes = _plan(device, kind, n, batch)
    rc = lib.hipfftSetStream(plan, ctypes.c_void_p(_lib.stream_ptr(device)))
    work = None
    if rc == 0 and work_bytes:
        # freed back to the caching allocator right after the launches are queued: stream order keeps it safe
        work = torch.empty((work_bytes,), dtype=torch.uint8, device=device)
        rc = lib.hipfftSetWorkArea(plan, ctypes.c_void_p(work.data_ptr()))
    if rc == 0:
        if kind == _HIPFFT_C2C:
            rc = lib.hipfftExecC2C(plan, ctypes.c_void_p(src_ptr), ctypes.c_void_p(dst_ptr), direction)
        elif kind == _HIPFFT_R2C:
            rc = lib.hipfftExecR2C(plan, ctypes.c_void_p(src_ptr), ctypes.c_void_p(dst_ptr))
        else:
            rc = lib.hipfftExecC2R(plan, ctypes.c_void_p(src_ptr), ctypes.c_void_p(dst_ptr))
    del work
    if rc != 0:
        raise AxisFftError(f"hipFFT execution (n={n}, batch={batch}, type={kind:#x}) failed with status {rc}")


def _transpose(src, dst, a: int, b: int, c: int, device) -> None:
    _lib.call("lsr_transpose_last2_c64", src.data_ptr(), dst.data_ptr(), a, b, c, _lib.stream_ptr(device))


def rfft3(volume):
    """Unnormalised forward transform of a contiguous float32 ``(Z, Y, X)`` device volume.

    Returns a complex64 tensor of shape ``(X // 2 + 1, Y, Z)``: ``torch.fft.rfftn(volume)`` with its
    axes reversed (``result[kx, ky, kz] == rfftn(volume)[kz, ky, kx]`` up to rounding).
    """
    import torch

    if not available():
        raise AxisFftError("hipFFT's C API is not loadable")
    if volume.dim() != 3 or volume.dtype != torch.float32 or not volume.is_contiguous() or volume.device.type != "cuda":
        raise ValueError("rfft3 takes a contiguous float32 (Z, Y, X) tensor on a HIP device")
    z, y, x = (int(v) for v in volume.shape)
    xc = x // 2 + 1
    dev = volume.device
    with torch.cuda.device(dev):
        a = torch.empty((z, y, xc), dtype=torch.complex64, device=dev)
        _exec(dev, _HIPFFT_R2C, x, z * y, volume.data_ptr(), a.data_ptr())
        b = torch.empty((z, xc, y), dtype=torch.complex64, device=dev)
        _transpose(a, b, z, y, xc, dev)                                   # [Z][Y][XC] -> [Z][XC][Y]
        _exec(dev, _HIPFFT_C2C, y, z * xc, b.data_ptr(), b.data_ptr(), _FORWARD)
        a = a.view(-1).view(xc, y, z)
        _transpose(b, a, 1, z, xc * y, dev)                               # [Z][XC Y] -> [XC Y][Z]
        del b
        _exec(dev, _HIPFFT_C2C, z, xc * y, a.data_ptr(), a.data_ptr(), _FORWARD)
    return a


def irfft3(spectrum, shape_zyx):
    """Unnormalised inverse of :func:`rfft3` (``N`` times ``irfftn``): a float32 ``(Z, Y, X)`` volume.
    ``spectrum`` (``(X // 2 + 1, Y, Z)`` complex64) is used as scratch and holds garbage afterwards."""
    import torch

    if not available():
        raise AxisFftError("hipFFT's C API is not loadable")
    z, y, x = (int(v) for v in shape_zyx)
    xc = x // 2 + 1
    if (tuple(spectrum.shape) != (xc, y, z) or spectrum.dtype != torch.complex64 or not spectrum.is_contiguous()
            or spectrum.device.type != "cuda"):
        raise ValueError(f"irfft3 takes a contiguous complex64 {(xc, y, z)} tensor on a HIP device")
    dev = spectrum.device
    with torch.cuda.device(dev):
        _exec(dev, _HIPFFT_C2C, z, xc * y, spectrum.data_ptr(), spectrum.data_ptr(), _BACKWARD)
        b = torch.empty((z, xc, y), dtype=torch.complex64, device=dev)
        _transpose(spectrum, b, 1, xc * y, z, dev)                        # [XC Y][Z] -> [Z][XC Y]
        _exec(dev, _HIPFFT_C2C, y, z * xc, b.data_ptr(), b.data_ptr(), _BACKWARD)
        a = spectrum.view(-1).view(z, y, xc)
        _transpose(b, a, z, xc, y, dev)                                   # [Z][XC][Y] -> [Z][Y][XC]
        del b
        out = torch.empty((z, y, x), dtype=torch.float32, device=dev)
        _exec(dev, _HIPFFT_C2R, x, z * y, a.data_ptr(), out.data_ptr())
    return out


_twiddles: dict = {}


def _twiddle_table(n: int, device):
    """``exp(-2 pi i k / n)``, k < n, complex64 on ``device`` (worked out in float64)."""
    import numpy as np
    import torch

    key = (int(n), device.index)
    t = _twiddles.get(key)
    if t is None:
        if len(_twiddles) >= 16:
            _twiddles.pop(next(iter(_twiddles)))
        k = np.arange(n, dtype=np.float64)
        t = _twiddles[key] = torch.as_tensor(np.exp(-2j * np.pi * k / n).astype(np.complex64), device=device)
    return t


def correlate_with_spectrum(ref_spectrum, volume):
    """``irfft3(ref_spectrum * conj(rfft3(volume)))`` -- the cross-correlation of the volume behind
    ``ref_spectrum`` (an :func:`rfft3` result, left intact) with ``volume``, unnormalised, float32
    ``(Z, Y, X)`` -- without the moving spectrum ever taking the ``[XC][Y][Z]`` layout: after the x
    and y transforms one kernel (``lsr_cross_correlate_z_c64``) does the forward z transform, the
    product and the inverse z transform on LDS-resident columns.  Needs a z length that kernel takes
    (5-smooth, <= 256: ``lsr_cross_correlate_z_supported``); callers fall back to
    ``rfft3`` / cross power / ``irfft3`` otherwise.
    """
    import torch

    if not available():
        raise AxisFftError("hipFFT's C API is not loadable")
    if volume.dim() != 3 or volume.dtype != torch.float32 or not volume.is_contiguous() or volume.device.type != "cuda":
        raise ValueError("correlate_with_spectrum takes a contiguous float32 (Z, Y, X) tensor on a HIP device")
    z, y, x = (int(v) for v in volume.shape)
    xc = x // 2 + 1
    if (tuple(ref_spectrum.shape) != (xc, y, z) or ref_spectrum.dtype != torch.complex64 or not ref_spectrum.is_contiguous()
            or ref_spectrum.device != volume.device):
        raise ValueError(f"the reference spectrum must be a contiguous complex64 {(xc, y, z)} tensor on {volume.device}")
    if not _lib.call_value("lsr_cross_correlate_z_supported", z):
        raise AxisFftError(f"z length {z} is not handled by lsr_cross_correlate_z_c64")
    dev = volume.device
    with torch.cuda.device(dev):
        a = torch.empty((z, y, xc), dtype=torch.complex64, device=dev)
        _exec(dev, _HIPFFT_R2C, x, z * y, volume.data_ptr(), a.data_ptr())
        b = torch.empty((z, xc, y), dtype=torch.complex64, device=dev)
        _transpose(a, b, z, y, xc, dev)                                   # [Z][Y][XC] -> [Z][XC][Y]
        _exec(dev, _HIPFFT_C2C, y, z * xc, b.data_ptr(), b.data_ptr(), _FORWARD)
        _lib.call("lsr_cross_correlate_z_c64", ref_spectrum.data_ptr(), b.data_ptr(), _twiddle_table(z, dev).data_ptr(),
                  z, y, xc, _lib.stream_ptr(dev))
        _exec(dev, _HIPFFT_C2C, y, z * xc, b.data_ptr(), b.data_ptr(), _BACKWARD)
        _transpose(b, a, z, xc, y, dev)                                   # [Z][XC][Y] -> [Z][Y][XC]
        del b
        out = torch.empty((z, y, x), dtype=torch.float32, device=dev)
        _exec(dev, _HIPFFT_C2R, x, z * y, a.data_ptr(), out.data_ptr())
    return out


# ---- the x leg in this package's own kernels (csrc/rfft_rows.hip) ------------------------------------------

_row_tables: dict = {}


def _row_twiddles(x: int, device):
    """``(exp(-2 pi i k / (x/2)), k < x/4)`` and ``(exp(-2 pi i k / x), k <= x/2)``, complex64 on ``device``."""
    import numpy as np
    import torch

    key = (int(x), device.index)
    t = _row_tables.get(key)
    if t is None:
        if len(_row_tables) >= 16:
            _row_tables.pop(next(iter(_row_tables)))
        m = x // 2
        half = np.exp(-2j * np.pi * np.arange(m // 2, dtype=np.float64) / m).astype(np.complex64)
        full = np.exp(-2j * np.pi * np.arange(m + 1, dtype=np.float64) / x).astype(np.complex64)
        t = _row_tables[key] = (torch.as_tensor(half, device=device), torch.as_tensor(full, device=device))
    return t


def rows_supported(shape_zyx) -> bool:
    """The row kernels take this FFT grid (x a multiple of 4, x / 2 5-smooth and <= 2048) and the z-leg
    kernel its z length: the whole correlation then runs without rocFFT along x and z."""
    z, _, x = (int(v) for v in shape_zyx)
    return (available() and bool(_lib.call_value("lsr_rfft_rows_supported", x))
            and bool(_lib.call_value("lsr_cross_correlate_z_supported", z)))


def _rows_forward(source, shape_zyx):
    """``[Z][XC][Y]`` complex64: ``source`` matched to the grid (reflect pad / centre crop, never written),
    transformed along x and along y."""
    import torch

    z, y, x = (int(v) for v in shape_zyx)
    xc = x // 2 + 1
    dev = source.device
    half, full = _row_twiddles(x, dev)
    b = torch.empty((z, xc, y), dtype=torch.complex64, device=dev)
    zi, yi, xi = (int(v) for v in source.shape)
    _lib.call("lsr_rfft_rows_t_c64", source.data_ptr(), zi, yi, xi, b.data_ptr(), z, y, x, half.data_ptr(), full.data_ptr(),
              _lib.stream_ptr(dev))
    _exec(dev, _HIPFFT_C2C, y, z * xc, b.data_ptr(), b.data_ptr(), _FORWARD)
    return b


def _check_source(source):
    import torch

    if source.dim() != 3 or source.dtype != torch.float32 or not source.is_contiguous() or source.device.type != "cuda":
        raise ValueError("a contiguous float32 (Z, Y, X) tensor on a HIP device is required")


def spectrum_of(source, shape_zyx):
    """:func:`rfft3` of ``source`` reflect-padded / centre-cropped to ``shape_zyx`` (``_match_shape``), with the
    padding, the x transform and the first transpose in one kernel: ``(X // 2 + 1, Y, Z)`` complex64."""
    import torch

    _check_source(source)
    if not rows_supported(shape_zyx):
        raise AxisFftError(f"FFT grid {tuple(shape_zyx)} is not handled by the row kernels")
    z, y, x = (int(v) for v in shape_zyx)
    xc = x // 2 + 1
    dev = source.device
    with torch.cuda.device(dev):
        b = _rows_forward(source, shape_zyx)
        a = torch.empty((xc, y, z), dtype=torch.complex64, device=dev)
        _transpose(b, a, 1, z, xc * y, dev)                               # [Z][XC Y] -> [XC Y][Z]
        del b
        _exec(dev, _HIPFFT_C2C, z, xc * y, a.data_ptr(), a.data_ptr(), _FORWARD)
    return a


def correlation_peak(ref_spectrum, source, shape_zyx):
    """Flat index (a one-element int64 device tensor) of ``argmax(fftshift(|corr|))``, ``corr`` = the
    cross-correlation on the FFT grid ``shape_zyx`` of the volume behind ``ref_spectrum`` (a
    :func:`spectrum_of` / :func:`rfft3` result, left intact) with ``source`` matched to the grid.  Six
    launches: rows forward (pad + x transform + transpose), y transform (rocFFT), z leg (transform, cross
    power, inverse), y inverse (rocFFT), rows inverse + peak search; neither the padded volume, nor the
    moving spectrum in z-contiguous layout, nor the correlation volume is ever written."""
    import torch

    _check_source(source)
    if not rows_supported(shape_zyx):
        raise AxisFftError(f"FFT grid {tuple(shape_zyx)} is not handled by the row kernels")
    z, y, x = (int(v) for v in shape_zyx)
    xc = x // 2 + 1
    if (tuple(ref_spectrum.shape) != (xc, y, z) or ref_spectrum.dtype != torch.complex64 or not ref_spectrum.is_contiguous()
            or ref_spectrum.device != source.device):
        raise ValueError(f"the reference spectrum must be a contiguous complex64 {(xc, y, z)} tensor on {source.device}")
    dev = source.device
    half, full = _row_twiddles(x, dev)
    with torch.cuda.device(dev):
        b = _rows_forward(source, shape_zyx)
        _lib.call("lsr_cross_correlate_z_c64", ref_spectrum.data_ptr(), b.data_ptr(), _twiddle_table(z, dev).data_ptr(),
                  z, y, xc, _lib.stream_ptr(dev))
        _exec(dev, _HIPFFT_C2C, y, z * xc, b.data_ptr(), b.data_ptr(), _BACKWARD)
        scratch = torch.empty((z * (-(-y // 8)) * 16,), dtype=torch.uint8, device=dev)
        peak = torch.empty((1,), dtype=torch.int64, device=dev)
        _lib.call("lsr_irfft_rows_peak", b.data_ptr(), z, y, x, half.data_ptr(), full.data_ptr(), peak.data_ptr(),
                  scratch.data_ptr(), _lib.stream_ptr(dev))
    return peak
