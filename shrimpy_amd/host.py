"""The hot path on HOST tensors: what runs when no HIP device is visible.

The reference resolves its device as ``cuda`` if available, else ``cpu`` (``shrimpy/preprocessing.py:78-82``)
and its CI has no GPU (``shrimpy/tests/conftest.py:11-17``), so a drop-in has to accept CPU tensors too --
BASELINE config 1 ("deskew-only via the CPU path, plumbing, no GPU") is exactly that.  The public functions
(``deskew.fast_deskew_zyx``, ``register.apply_affine_transform_zyx``, ``deconvolve.richardson_lucy`` /
``correlate3d``) hand CPU tensors to the functions below, which call the native host twins of
``csrc/host_twins.hip`` (``lsr_*_cpu``: the device entry points' signatures, host pointers, the same
arithmetic in the same order -- results equal the kernels' bit for bit).  Nothing here imports, calls or
reads anything under ``oracle/``; the library must be built (``_lib.load`` raises otherwise).

Threads: ``lsr_set_host_threads`` is set from ``torch.get_num_threads()`` at each call -- the knob the
reference's CPU path is governed by -- and the twins use plain ``std::thread`` (no second OpenMP runtime).
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

__all__ = ["is_host", "deskew_with_matrix", "average_n_slices", "apply_affine", "correlate3d", "richardson_lucy"]

_MODES = {"constant": _lib.MODE_CONSTANT, "grid-constant": _lib.MODE_GRID_CONSTANT}


def is_host(t) -> bool:
    import torch

    return isinstance(t, torch.Tensor) and t.device.type == "cpu"


_warned = [False]


def _threads() -> None:
    """Worker count of the twins = torch's CPU thread count; and, once per process, a warning when a HIP device
    is visible: the kernels are the product's path there, and a CPU tensor reaching this module on such a box
    should be a decision, not an accident (nothing falls back silently)."""
    import torch

    if not _warned[0] and torch.cuda.is_available():
        import warnings

        _warned[0] = True
        warnings.warn("shrimpy_amd: a CPU tensor is being processed by the native host twins although a HIP device is "
                      "visible; move the tensor to the device to run the gfx950 kernels", RuntimeWarning, stacklevel=3)
    _lib.call("lsr_set_host_threads", max(1, min(1024, int(torch.get_num_threads()))))


def _f32(t, name: str):
    import torch

    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if t.device.type != "cpu":
        raise ValueError(f"{name} is on {t.device}; the host path takes CPU tensors")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def average_n_slices(data, avg: int):
    import torch

    data = _f32(data, "data")
    if avg == 1:
        return data
    zd, y, x = (int(v) for v in data.shape)
    out = torch.empty((-(-zd // avg), y, x), dtype=torch.float32)
    _threads()
    _lib.call("lsr_average_slices_f32_cpu", data.data_ptr(), zd, y, x, out.data_ptr(), int(out.shape[0]), avg, None)
    return out


def deskew_with_matrix(raw, m, pre_average_shape, avg: int, out=None, flat_field=None, border: str = "constant", cval=None):
    """``deskew.deskew_with_matrix`` for a CPU tensor (float32 or uint16 counts), dense ``out`` only.
    ``cval``: ``None`` (zero) or a one-element float32 CPU tensor (``deskew.fill_value``)."""
    import torch

    if flat_field is not None:     # (the fusion is a kernel matter; the values are those of correcting first)
        raw = flat_field.apply(raw)
    u16 = raw.dtype == torch.uint16
    if not u16:
        raw = _f32(raw, "raw_data")
    elif not raw.is_contiguous():
        raise ValueError("raw_data must be contiguous")
    zd, yo, xo = (int(v) for v in pre_average_shape)
    zo = -(-zd // avg)
    if out is None:
        out = torch.empty((zo, yo, xo), dtype=torch.float32)
    else:
        if hasattr(out, "logical_ptr"):
            raise ValueError("a padded device volume cannot receive a host deskew")
        _f32(out, "out")
        if tuple(out.shape) != (zo, yo, xo):
            raise ValueError(f"out must be {(zo, yo, xo)} on cpu")
    z, y, x = (int(v) for v in raw.shape)
    _threads()
    try:
        if border != "constant":
            raise _lib.LsrUnsupported("lsr_deskew_f32_cpu", _lib.E_UNSUPPORTED, "grid-constant border")
        if cval is not None:
            _lib.call("lsr_deskew_cval_cpu", raw.data_ptr(), 1 if u16 else 0, z, y, x, out.data_ptr(), zo, yo, xo, xo, yo * xo,
                      zd, _lib.matrix12(m), avg, _lib.MODE_CONSTANT, None, None, cval.data_ptr(), None)
        else:
            _lib.call("lsr_deskew_u16_cpu" if u16 else "lsr_deskew_f32_cpu", raw.data_ptr(), z, y, x, out.data_ptr(),
                      zo, yo, xo, xo, yo * xo, zd, _lib.matrix12(m), avg, None)
    except _lib.LsrUnsupported:
        # a general matrix, or the blending border rule: trilinear resample, then average
        src = raw.to(torch.float32) if u16 else raw
        pre = out if avg == 1 else torch.empty((zd, yo, xo), dtype=torch.float32)
        _lib.call("lsr_affine_f32_cpu", src.data_ptr(), z, y, x, pre.data_ptr(), zd, yo, xo, _lib.matrix12(m),
                  ctypes.c_float(0.0 if cval is None else float(cval.item())), _MODES[border], None)
        if avg > 1:
            _lib.call("lsr_average_slices_f32_cpu", pre.data_ptr(), zd, yo, xo, out.data_ptr(), zo, avg, None)
    _lib.mark_written(out)
    return out


def apply_affine(moving, m, shape, mode: str, cval: float, out=None):
    """``register.apply_affine_transform_zyx`` for a CPU tensor: scipy's fp64 arithmetic (``exact``)."""
    import torch

    moving = _f32(moving, "moving")
    if out is None:
        out = torch.empty(shape, dtype=torch.float32)
    else:
        if hasattr(out, "logical_ptr"):
            raise ValueError("a padded device volume cannot receive a host resample")
        _f32(out, "out")
        if tuple(out.shape) != tuple(shape):
            raise ValueError(f"out must be {tuple(shape)} on cpu")
        if out.data_ptr() == moving.data_ptr():
            raise ValueError("out must not alias moving")
    zi, yi, xi = (int(v) for v in moving.shape)
    _threads()
    _lib.call("lsr_affine_f32_cpu", moving.data_ptr(), zi, yi, xi, out.data_ptr(), int(shape[0]), int(shape[1]),
              int(shape[2]), _lib.matrix12(m), ctypes.c_float(float(cval)), _MODES[mode], None)
    _lib.mark_written(out)
    return out


def _taps(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def correlate3d(volume, weights=None, weight_factors=None):
    """``scipy.ndimage.correlate(volume, weights, mode="constant", cval=0)`` on a CPU tensor."""
    import torch

    from .deconvolve import prepare_psf

    vol = _f32(volume, "volume")
    if vol.dim() != 3:
        raise ValueError("volume must be (Z, Y, X)")
    out = torch.empty_like(vol)
    z, y, x = (int(v) for v in vol.shape)
    _threads()
    if weight_factors is not None:
        wz, wy, wx = (_taps(np.asarray(k).ravel()) for k in weight_factors)
        _lib.call("lsr_correlate_sep_f32_cpu", vol.data_ptr(), out.data_ptr(), None, z, y, x, wz.ctypes.data, len(wz),
                  wy.ctypes.data, len(wy), wx.ctypes.data, len(wx), _lib.EPI_NONE, ctypes.c_float(0.0), None, None, None,
                  None)
    else:
        w = prepare_psf(weights)
        _lib.call("lsr_correlate_dense_f32_cpu", vol.data_ptr(), out.data_ptr(), None, z, y, x, w.ctypes.data, w.shape[0],
                  w.shape[1], w.shape[2], _lib.EPI_NONE, ctypes.c_float(0.0), None, None)
    return out


def richardson_lucy(y, psf=None, iterations: int = 20, eps: float = 1e-6, x0=None, *, separable: str = "auto",
                    separable_rtol: float = 1e-6, psf_factors=None, tol: float | None = None,
                    return_stats: bool = False):
    """``deconvolve.richardson_lucy`` for a CPU tensor: ``x <- x * H^T(y / (H x + eps)) / H^T 1`` with the two
    correlations and their epilogues as the host twins of the device launches (rank-1 PSFs run the
    separable form, others the dense one).  ``tol`` / ``return_stats``: the iteration scalars of
    ``deconvolve.RLStats``, summed by the twins' UPDATE pass; the loop stops after the first iteration whose relative
    change is below ``tol``."""
    import torch

    from .deconvolve import MAX_TAPS, RLStats, _axis_norm, _prefix_table, factor_psf, prepare_psf

    if tol is not None and not (tol >= 0 and np.isfinite(tol)):
        raise ValueError("tol must be a finite number >= 0")
    want = bool(return_stats) or tol is not None

    y = _f32(y, "y")
    if separable not in ("auto", "force", "never"):
        raise ValueError("separable must be 'auto', 'force' or 'never'")
    iterations = int(iterations)
    if iterations < 0:
        raise ValueError("iterations must be >= 0")
    if not eps > 0:
        raise ValueError("eps must be > 0")
    factors = None
    if psf_factors is not None:
        factors = tuple(_taps(np.asarray(k).ravel()) for k in psf_factors)
        from .deconvolve import MAX_Z_TAPS

        if (len(factors) != 3 or any(len(k) % 2 == 0 for k in factors) or len(factors[0]) > MAX_Z_TAPS
                or max(len(factors[1]), len(factors[2])) > MAX_TAPS):
            raise ValueError("psf_factors must be three odd-length 1-D kernels (<= 31 taps along z, <= 15 in plane)")
    else:
        # (up to the extents the device takes through the Fourier domain: the loops below take any count)
        from .deconvolve_fft import MAX_FFT_TAPS

        w = prepare_psf(psf, MAX_FFT_TAPS, MAX_FFT_TAPS)
        if separable != "never":
            factors = factor_psf(w, separable_rtol)
            if factors is None and separable == "force":
                raise ValueError("psf is not rank-1 within separable_rtol")
    init = y if x0 is None else _f32(x0, "x0")
    if tuple(init.shape) != tuple(y.shape):
        raise ValueError(f"x0 must be {tuple(y.shape)}")
    x = init.clone()
    if iterations == 0:
        return (x, RLStats.from_array(np.zeros((0, 3)), 0)) if return_stats else x
    stats = np.zeros((iterations, 3), dtype=np.float64) if want else None

    def met(i):
        return stats[i, 2] == 0 or stats[i, 1] <= tol * stats[i, 2]
    z, yy, xx = (int(v) for v in y.shape)
    ratio, nxt = torch.empty_like(y), torch.empty_like(y)
    e = ctypes.c_float(eps)
    _threads()
    if factors is not None:
        k = [_taps(f) for f in factors]
        kf = [_taps(f[::-1]) for f in factors]
        norm = [_axis_norm(f, n) for f, n in zip(factors, (z, yy, xx))]
        sizes = [len(f) for f in k]

        def corr(src, dst, aux, taps, epi, srow=None):
            _lib.call("lsr_correlate_sep_stats_f32_cpu", src.data_ptr(), dst.data_ptr(), aux.data_ptr(), z, yy, xx,
                      taps[0].ctypes.data, sizes[0], taps[1].ctypes.data, sizes[1], taps[2].ctypes.data, sizes[2], epi, e,
                      norm[0].ctypes.data, norm[1].ctypes.data, norm[2].ctypes.data,
                      None if srow is None else srow.ctypes.data, None)
    else:    # the dense loop is one native call per chunk (the twin of lsr_rl_dense_f32: x updated in place)
        k, kf = _taps(w), _taps(w[::-1, ::-1, ::-1])
        table = np.ascontiguousarray(_prefix_table(w).ravel(), dtype=np.float64)
        # ``tol`` as the device plans read it (deconvolve.RichardsonLucyPlan._run_to_tolerance): the scalars of iteration i
        # are looked at after iteration i + 1 has run, so the estimate returned is the one iteration PAST the first that
        # met tol -- the same iterate, and the same RLStats.iterations, whichever device the tensor lives on
        done, stopped = 0, False
        step = iterations if tol is None else 1
        while done < iterations:
            _lib.call("lsr_rl_dense_stats_f32_cpu", y.data_ptr(), x.data_ptr(), ratio.data_ptr(), z, yy, xx, k.ctypes.data,
                      kf.ctypes.data, w.shape[0], w.shape[1], w.shape[2], table.ctypes.data, step, e,
                      None if stats is None else stats[done:].ctypes.data, None)
            done += step
            if tol is not None and done >= 2 and met(done - 2):
                stopped = True
                break
        if tol is not None and not stopped:
            stopped = bool(met(done - 1))
        return (x, RLStats.from_array(stats, done, stopped)) if return_stats else x
    done, stopped = 0, False
    for it in range(iterations):
        corr(x, ratio, y, kf, _lib.EPI_RATIO)       # ratio = y / (H x + eps): H = correlation with the flipped taps
        corr(ratio, nxt, x, k, _lib.EPI_UPDATE, None if stats is None else stats[it])     # x <- x * H^T ratio / H^T 1
        x, nxt = nxt, x
        done = it + 1
        if tol is not None and it >= 1 and met(it - 1):     # (one iteration past the first that met tol: see above)
            stopped = True
            break
    if tol is not None and not stopped:
        stopped = bool(met(done - 1))
    return (x, RLStats.from_array(stats, done, stopped)) if return_stats else x
