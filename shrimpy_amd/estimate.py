"""Estimate the affine that registers one volume onto another (label-free <-> light-sheet) on MI355X.

The "estimate" half of the registration step; ``register.apply_affine_transform_zyx`` is the "apply"
half.  The reference ships neither ("algorithms for deconvolution and registration ... are being
developed", ``docs/data_structure.md:58-62``); SURVEY.md section 8 f-4 lists the estimation as the row that
closes the register loop end to end.  What is estimated is exactly what ``RegisterSettings`` stores and
the apply kernels consume: ``affine_transform_zyx``, a 4x4 in ZYX voxel units mapping a TARGET index to
the MOVING coordinate (the ``scipy.ndimage.affine_transform`` convention).

Method: Gauss-Newton on the sum of squared differences with a linear intensity map,

    minimise  sum_x ( gain * M(A x + t) + offset - T(x) )^2 ,

coarse to fine (Gaussian blur + strided sampling of the target grid).  Each iteration is ONE launch of
``lsr_affine_normal_equations_f32`` (``csrc/estimate_affine.hip``: trilinear taps, analytic gradient,
the 14 x 14 normal equations accumulated in fp64 registers) and a 14 x 14 solve on the host; an
optional phase cross-correlation (the DynaTrack kernels) supplies the starting translation.
The estimate runs on a HIP device only (its host side is a 14 x 14 solve; no host twin of the normal-equations kernel).
"""

from __future__ import annotations

import ctypes
import logging

from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .geometry import as_matrix_3x4

logger = logging.getLogger(__name__)

__all__ = ["RegistrationEstimate", "normal_equations", "estimate_affine_zyx", "default_levels"]

N_PARAMS = 14          # 12 affine + gain + offset
MODELS = ("translation", "affine")


@dataclass
class RegistrationEstimate:
    """Result of :func:`estimate_affine_zyx`."""

    affine_transform_zyx: np.ndarray      # 4x4, target index -> moving coordinate
    gain: float
    offset: float
    rms: float                            # residual RMS on the finest level, in target intensity units
    n_samples: int
    iterations: int
    converged: bool
    history: list = field(default_factory=list)   # (stride, iteration, rms, step in voxels) per accepted step

    def to_settings_dict(self, **extra) -> dict:
        """The ``RegisterSettings`` YAML mapping for this estimate."""
        return {"affine_transform_zyx": [[float(v) for v in row] for row in self.affine_transform_zyx], **extra}


def _f64p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _unpack(row: np.ndarray):
    """(H 14x14 symmetric, b 14, sse, n) from one 121-entry row of sums."""
    h = np.zeros((N_PARAMS, N_PARAMS))
    h[np.triu_indices(N_PARAMS)] = row[:105]
    h = h + np.triu(h, 1).T
    return h, row[105:119].copy(), float(row[119]), int(round(row[120]))


def _stride3(stride) -> tuple[int, int, int]:
    if np.isscalar(stride):
        return (int(stride),) * 3
    st = tuple(int(v) for v in stride)
    if len(st) != 3:
        raise ValueError(f"stride must be an int or three ints, got {stride!r}")
    return st


def normal_equations(moving, target, matrix, gain: float = 1.0, offset: float = 0.0, stride=1,
                     centre=None, scale: float | None = None):
    """One launch of the normal-equations kernel: ``(H, b, sse, n)`` of the Gauss-Newton step at
    ``matrix`` (3x4 / 4x4, target index -> moving coordinate) over the target grid sampled every
    ``stride`` voxels (an int, or one per axis).  Parameter order: the 3x4 matrix row by row IN
    CENTRED, SCALED target coordinates ``((x - centre) / scale, 1)``, then gain, offset."""
    import torch

    mov = _lib.require_device_f32(moving, "moving")
    tgt = _lib.require_device_f32(target, "target")
    if mov.dim() != 3 or tgt.dim() != 3 or mov.device != tgt.device:
        raise ValueError("moving and target must be (Z, Y, X) tensors on the same device")
    m = as_matrix_3x4(matrix)
    shape = tuple(int(v) for v in tgt.shape)
    c = np.ascontiguousarray(centre if centre is not None else [(n - 1) / 2 for n in shape], dtype=np.float64)
    s = float(scale if scale is not None else max(shape) / 2)
    st = (ctypes.c_int * 3)(*_stride3(stride))
    n_out = _lib.call_value("lsr_affine_normal_size")
    n_blocks = _lib.call_value("lsr_affine_normal_blocks")
    partial = torch.empty((n_blocks, n_out), dtype=torch.float64, device=tgt.device)
    with torch.cuda.device(tgt.device):
        _lib.call("lsr_affine_normal_equations_f32", mov.data_ptr(), *(int(v) for v in mov.shape), tgt.data_ptr(),
                  *shape, _lib.matrix12(m), ctypes.c_double(float(gain)), ctypes.c_double(float(offset)), st,
                  _f64p(c), ctypes.c_double(s), partial.data_ptr(), _lib.stream_ptr(tgt.device))
    # the 256 workgroup rows are added in row order on the host: the same sums on every run
    return _unpack(partial.cpu().numpy().sum(axis=0))


def _to_normalised(m: np.ndarray, c: np.ndarray, s: float) -> np.ndarray:
    """3x4 in voxel units -> 3x4 acting on ((x - c) / s, 1)."""
    q = np.empty((3, 4))
    q[:, :3] = m[:, :3] * s
    q[:, 3] = m[:, :3] @ c + m[:, 3]
    return q


def _from_normalised(q: np.ndarray, c: np.ndarray, s: float) -> np.ndarray:
    m = np.empty((3, 4))
    m[:, :3] = q[:, :3] / s
    m[:, 3] = q[:, 3] - m[:, :3] @ c
    return m


def _corner_motion(a: np.ndarray, b: np.ndarray, shape) -> float:
    """Largest displacement (voxels) of the target volume's corners between two 3x4 maps."""
    corners = np.array([[z, y, x, 1.0] for z in (0, shape[0] - 1) for y in (0, shape[1] - 1) for x in (0, shape[2] - 1)])
    return float(np.abs(corners @ (a - b).T).max())


def default_levels(shape, max_stride: int = 32, min_samples: int = 8):
    """Coarse-to-fine schedule ``[(strides zyx, sigmas zyx), ...]`` for a target of ``shape``: strides
    halve from the coarsest level -- per axis the largest power of two (<= ``max_stride``) that still
    leaves ``min_samples`` samples along that axis -- down to 1; sigma = stride / 2 (0 at stride 1).  A
    Gauss-Newton step only sees displacements of a few sigma: the coarsest level has to span the
    largest misalignment at the volume's corners (a 2 % scale error is 20 voxels at 1024)."""
    top = []
    for n in shape:
        s = 1
        while s * 2 <= max_stride and n // (s * 2) >= min_samples:
            s *= 2
        top.append(s)
    levels = []
    k = max(top)
    while k >= 1:
        strides = tuple(min(k, t) for t in top)
        levels.append((strides, tuple(st / 2 if st > 1 else 0.0 for st in strides)))
        k //= 2
    return levels


def _blur_axes(vol, sigmas):
    """Separable Gaussian blur with a sigma per axis (reflect padding; the kernel of DESIGN.md 4.7)."""
    import torch

    if not any(sg > 0 for sg in sigmas):
        return vol
    z, y, x = (int(v) for v in vol.shape)
    src = vol
    with torch.cuda.device(vol.device):
        stream = _lib.stream_ptr(vol.device)
        for axis, (n, sg) in enumerate(zip((z, y, x), sigmas)):
            if sg <= 0:
                continue
            r = min(int(4 * sg + 0.5), n - 1)
            if r < 1:
                continue
            xs = torch.arange(-r, r + 1, device=vol.device, dtype=torch.float32)
            k1d = torch.exp(-0.5 * (xs / float(sg)) ** 2)
            k1d = (k1d / k1d.sum()).contiguous()
            dst = torch.empty_like(vol)
            _lib.call("lsr_blur_reflect_f32", src.data_ptr(), dst.data_ptr(), z, y, x, axis, k1d.data_ptr(), r,
                      ctypes.c_float(0.0), ctypes.c_float(0.0), stream)
            src = dst
    return src


def estimate_affine_zyx(moving, target, *, initial=None, model: str = "affine", intensity: bool = True,
                        levels=None, max_iterations: int = 40, tol: float = 2e-3,
                        init_translation: str | None = "pcc") -> RegistrationEstimate:
    """Estimate ``affine_transform_zyx`` (target index -> moving coordinate) between two device volumes.

    Parameters
    ----------
    moving, target : (Z, Y, X) float32 device tensors (the result resamples ``moving`` onto ``target``'s grid).
    initial : 4x4 / 3x4 starting map; default identity (plus ``init_translation``).
    model : ``"affine"`` (12 parameters) or ``"translation"`` (3).
    intensity : also fit ``gain`` / ``offset`` of the linear intensity map (two modalities or exposures).
    levels : ``(stride, sigma)`` per resolution level, coarse to fine, each an int / float or one per
        axis: both volumes are blurred with a Gaussian of ``sigma`` voxels and the target grid is sampled
        every ``stride`` voxels.  Default: :func:`default_levels` of the target shape.
    tol : stop a level when a step moves no corner of the target volume by more than ``tol`` voxels.
    init_translation : ``"pcc"`` = whole-voxel shift from the phase cross-correlation, ``None`` = none.
    """
    from .dynatrack import _phase_cross_corr

    if model not in MODELS:
        raise ValueError(f"model must be one of {MODELS}, got {model!r}")
    if init_translation not in (None, "pcc"):
        raise ValueError("init_translation must be 'pcc' or None")
    mov = _lib.require_device_f32(moving, "moving")
    tgt = _lib.require_device_f32(target, "target")
    if mov.dim() != 3 or tgt.dim() != 3:
        raise ValueError("moving and target must be (Z, Y, X)")
    shape = tuple(int(v) for v in tgt.shape)
    if levels is None:
        levels = default_levels(shape)
    levels = [(_stride3(st), tuple(float(v) for v in (np.full(3, sg) if np.isscalar(sg) else sg))) for st, sg in levels]
    c = np.array([(n - 1) / 2 for n in shape])
    s = max(shape) / 2
    m = as_matrix_3x4(initial if initial is not None else np.eye(4)).copy()
    gain, offset = 1.0, 0.0
    if intensity:   # moments give the starting intensity map
        sm, st_ = float(mov.std()), float(tgt.std())
        if sm > 0 and st_ > 0:
            gain = st_ / sm
            offset = float(tgt.mean()) - gain * float(mov.mean())

    def free_mask(kind):
        free = np.zeros(N_PARAMS, dtype=bool)
        free[[3, 7, 11]] = True
        if kind == "affine":
            free[:12] = True
        if intensity:
            free[12:] = True
        return free

    if init_translation == "pcc" and initial is None and tuple(mov.shape) == shape:
        shift = np.array(_phase_cross_corr(tgt, mov), dtype=np.float64)
        if np.any(shift):
            # the correlation's sign convention is settled by the data: keep whichever direction
            # (or neither) has the smaller residual
            probe = tuple(max(2, v) for v in levels[0][0])
            best = None
            for sign in (0.0, 1.0, -1.0):
                trial = m.copy()
                trial[:, 3] += sign * shift
                _, _, sse, n = normal_equations(mov, tgt, trial, gain, offset, probe, c, s)
                if n > 0.25 * tgt.numel() / np.prod(probe) and (best is None or sse / n < best[0]):
                    best = (sse / n, trial)
            if best is not None:
                m = best[1]

    history, total_iters, converged = [], 0, False
    rms, n_used = float("nan"), 0
    for li, (strides, sigmas) in enumerate(levels):
        level = (_blur_axes(mov, sigmas), _blur_axes(tgt, sigmas), strides)
        h, b, sse, n = normal_equations(level[0], level[1], m, gain, offset, strides, c, s)
        if n < 64:
            raise _lib.LsrError("estimate_affine_zyx", -1, f"only {n} target samples fall inside the moving volume "
                                "at the starting transform: give a better `initial`")
        # the coarsest level settles the translation before it frees the linear part
        stages = (["translation", model] if (li == 0 and model == "affine" and len(levels) > 1) else [model])
        for kind in stages:
            free = free_mask(kind)
            lam, converged = 1e-3, False
            for it in range(int(max_iterations)):
                hf, bf = h[np.ix_(free, free)], b[free]
                step = None
                for _ in range(8):   # Levenberg-Marquardt: raise the damping until the residual goes down
                    try:
                        delta = np.linalg.solve(hf + lam * np.diag(np.diag(hf)) + 1e-12 * np.eye(hf.shape[0]), -bf)
                    except np.linalg.LinAlgError:
                        lam *= 10
                        continue
                    full = np.zeros(N_PARAMS)
                    full[free] = delta
                    q = _to_normalised(m, c, s) + full[:12].reshape(3, 4)
                    m_new = _from_normalised(q, c, s)
                    g_new, o_new = gain + full[12], offset + full[13]
                    h2, b2, sse2, n2 = normal_equations(level[0], level[1], m_new, g_new, o_new, strides, c, s)
                    if n2 >= 0.5 * n and sse2 / max(n2, 1) <= sse / n * (1 + 1e-12):
                        step = (m_new, g_new, o_new, h2, b2, sse2, n2)
                        lam = max(lam / 3, 1e-9)
                        break
                    lam *= 10
                total_iters += 1
                if step is None:
                    converged = True    # no downhill step left at this level
                    break
                moved = _corner_motion(step[0], m, shape)
                m, gain, offset, h, b, sse, n = step
                history.append((strides, it, float(np.sqrt(sse / n)), moved))
                if moved < tol * max(strides):
                    converged = True
                    break
        rms, n_used = float(np.sqrt(sse / n)), n
        logger.info("estimate_affine: strides %s sigmas %s -> rms %.4g on %d samples", strides, sigmas, rms, n)
    out = np.eye(4)
    out[:3] = m
    return RegistrationEstimate(out, float(gain), float(offset), rms, n_used, total_iters, converged, history)
