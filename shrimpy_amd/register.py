"""Affine registration apply (label-free <-> fluorescence) on MI355X.

The reference has no symbol for this step ("algorithms for deconvolution and registration ...
are being developed", ``docs/data_structure.md:58-62``); the north-star defines it as the
``scipy.ndimage.affine_transform(moving, M, output_shape=target.shape, order=1,
mode="constant", cval=0)`` path.  ``affine_transform_zyx`` is a 4x4 homogeneous matrix in ZYX voxel
units mapping TARGET (output) index -> SOURCE (moving) coordinate.

Runs the HIP kernel ``lsr_affine_f32`` (``csrc/affine.hip``); results are bit-identical to scipy
for finite inputs.  No CPU fallback.
"""

from __future__ import annotations

import ctypes

from . import _lib
from .geometry import as_matrix_3x4

__all__ = ["apply_affine_transform_zyx", "affine_transform", "PitchedVolume"]

_MODES = {"constant": _lib.MODE_CONSTANT, "grid-constant": _lib.MODE_GRID_CONSTANT}


class PitchedVolume:
    """A (Z, Y, X) float32 volume whose rows are padded with zeros to a multiple of 4 floats.

    The LDS-staged affine kernels move 16-byte chunks and need 16-byte aligned rows; a deskewed
    volume is ``ceil(Z / r - Y cos(theta))`` wide -- a multiple of 4 one time in four.  ``.view`` is the
    logical window of ``.full`` (``(Z, Y, pitch)``); the deskew writes into it in place
    (``deskew_with_matrix(out=...)`` takes any object with ``logical_ptr`` / ``pitch`` / ``plane``),
    ``apply_affine_transform_zyx`` reads it through ``lsr_affine_pitched_f32``.
    """

    def __init__(self, shape_zyx, device):
        import torch

        z, y, x = (int(v) for v in shape_zyx)
        self.pitch = (x + 3) & ~3
        self.plane = y * self.pitch
        self.full = torch.zeros((z, y, self.pitch), dtype=torch.float32, device=device)
        self.view = self.full[:, :, :x]

    @property
    def shape(self):
        return tuple(self.view.shape)

    def logical_ptr(self) -> int:
        return self.view.data_ptr()

    @classmethod
    def copy_of(cls, volume) -> "PitchedVolume":
        out = cls(volume.shape, volume.device)
        out.view.copy_(volume)
        return out



def apply_affine_transform_zyx(moving, affine_transform_zyx, output_shape_zyx=None, *,
                               mode: str = "constant", cval: float = 0.0, out=None,
                               exact: bool = True):
    """Resample ``moving`` (Z, Y, X float32 device tensor) onto the target grid.

    Parameters
    ----------
    affine_transform_zyx : 4x4 or 3x4 array-like, target index -> moving coordinate.
    output_shape_zyx : target grid shape; defaults to ``moving.shape``.
    mode : ``"constant"`` (scipy default: any coordinate outside ``[0, n-1]`` gives ``cval``) or
        ``"grid-constant"`` (blend towards ``cval`` across the border).
    moving : a dense tensor, or a :class:`PitchedVolume` (rows padded to 16-byte multiples).
    out : a dense tensor, or a padded working volume with ``logical_ptr`` / ``pitch`` / ``plane`` / ``view``
        (``deconvolve.PaddedVolume``): the result then lands where the RL kernels read it.
    exact : ``True`` (default) interpolates in fp64 in scipy's operation order -- bit-identical to
        ``scipy.ndimage.affine_transform``; ``False`` interpolates in float32 (border decisions
        unchanged, ~1e-6 relative difference), which is about 2x faster (HBM-bound).
    """
    import torch

    if mode not in _MODES:
        raise ValueError(f"mode must be one of {sorted(_MODES)}, got {mode!r}")
    pitched = moving if isinstance(moving, PitchedVolume) else None
    if pitched is not None:
        moving = pitched.view
    if not isinstance(moving, torch.Tensor):
        raise TypeError(f"moving must be a torch.Tensor, got {type(moving).__name__}")
    if moving.dim() != 3:
        raise ValueError(f"moving must be (Z, Y, X), got shape {tuple(moving.shape)}")
    if pitched is None and moving.device.type == "cpu":
        # no HIP device in play: the native host twin (always scipy's fp64 arithmetic, whatever `exact` says)
        from . import host

        src = moving.to(torch.float32).contiguous()
        shape = tuple(int(v) for v in (output_shape_zyx if output_shape_zyx is not None else src.shape))
        if len(shape) != 3 or min(shape) <= 0:
            raise ValueError(f"output_shape_zyx must be three positive ints, got {shape}")
        return host.apply_affine(src, as_matrix_3x4(affine_transform_zyx), shape, mode, cval, out=out)
    if pitched is None:
        if moving.dtype != torch.float32:
            moving = moving.to(torch.float32)
        # (a dense volume whose width is not a multiple of 4 runs the gather kernel: making the padded
        # copy here costs more than the LDS-staged kernels save -- 3.5 against 3.2 ms at 171 x 2048 x 2270;
        # a producer that can write padded rows, like the deskew, hands over a PitchedVolume instead)
        moving = _lib.require_device_f32(moving.contiguous(), "moving")
    m = as_matrix_3x4(affine_transform_zyx)
    shape = tuple(int(v) for v in (output_shape_zyx if output_shape_zyx is not None else moving.shape))
    if len(shape) != 3 or min(shape) <= 0:
        raise ValueError(f"output_shape_zyx must be three positive ints, got {shape}")
    result = out
    if out is None:
        out = result = torch.empty(shape, dtype=torch.float32, device=moving.device)
        out_ptr, out_pitch, out_plane = out.data_ptr(), shape[2], shape[1] * shape[2]
    elif hasattr(out, "logical_ptr"):   # a padded working volume (e.g. the RL plan's input): written in place
        if tuple(out.view.shape) != shape or out.full.device != moving.device:
            raise ValueError(f"out must hold a {shape} window on {moving.device}")
        if out.full.data_ptr() == (pitched.full if pitched is not None else moving).data_ptr():
            raise ValueError("out must not alias moving")
        out_ptr, out_pitch, out_plane = out.logical_ptr(), out.pitch, out.plane
    else:
        _lib.require_device_f32(out, "out")
        if tuple(out.shape) != shape or out.device != moving.device:
            raise ValueError(f"out must be {shape} on {moving.device}")
        if out.data_ptr() == moving.data_ptr():
            raise ValueError("out must not alias moving")
        out_ptr, out_pitch, out_plane = out.data_ptr(), shape[2], shape[1] * shape[2]
    zi, yi, xi = (int(v) for v in moving.shape)
    in_ptr, in_pitch, in_plane = ((pitched.logical_ptr(), pitched.pitch, pitched.plane) if pitched is not None
                                  else (moving.data_ptr(), xi, yi * xi))
    flags = _MODES[mode] | (0 if exact else _lib.MODE_F32_INTERP)
    with torch.cuda.device(moving.device):
        _lib.call(
            "lsr_affine_pitched_f32", in_ptr, zi, yi, xi, in_pitch, in_plane, out_ptr, shape[0], shape[1], shape[2],
            out_pitch, out_plane, _lib.matrix12(m), ctypes.c_float(float(cval)), flags, _lib.stream_ptr(moving.device),
        )
    _lib.mark_written(out.full if hasattr(out, "logical_ptr") else out)
    return result


def affine_transform(input, matrix, offset=0.0, output_shape=None, order=1, mode="constant",
                     cval=0.0):
    """``scipy.ndimage.affine_transform``-shaped convenience (order 1 only, 3-D, device tensors)."""
    import numpy as np

    if order != 1:
        raise ValueError("only order=1 (trilinear) is implemented")
    mat = np.asarray(matrix, dtype=np.float64)
    if mat.shape == (3,):
        mat = np.diag(mat)
    if mat.shape == (3, 3):
        off = np.broadcast_to(np.asarray(offset, dtype=np.float64), (3,))
        mat = np.concatenate([mat, off[:, None]], axis=1)
    return apply_affine_transform_zyx(input, mat, output_shape, mode=mode, cval=cval)
