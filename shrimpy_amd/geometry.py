"""Deskew geometry: the output->input affine map and the output-shape rule (host logic).

These are the conventions of ``biahub.deskew`` at the revision the reference pins
(``pyproject.toml:91``), as the reference's call sites use them:

* ``get_deskewed_data_shape(raw_data_shape=, ls_angle_deg=, px_to_scan_ratio=, keep_overhang=,
  average_n_slices=, pixel_size_um=) -> (shape, voxel_size)`` -- ``shrimpy/preprocessing.py:226-231``,
  ``scripts/measure_psf.py:230-234``;
* raw axes ``(Z=scan, Y=tilt, X=coverslip)`` -- ``scripts/measure_psf.py:91,101``;
* raw X maps 1:1, reversed, onto output axis -2 -- ``scripts/measure_psf.py:221,249``;
* one output plane gathers one tilt row across the scan stack -- ``shrimpy/viewer/ring_buffer.py:98-105``.

The matrix is an explicit input of the HIP kernel (``lsr_deskew_f32``), so a correction of the
convention never touches device code.
"""

from __future__ import annotations

import math

from typing import NamedTuple

import numpy as np


class DeskewGeometry(NamedTuple):
    matrix_3x4: np.ndarray  # float64, output (Z', Y', X') index -> raw (z, y, x) coordinate
    pre_average_shape: tuple[int, int, int]  # (Y, X, Xp)
    output_shape: tuple[int, int, int]  # (ceil(Y / avg), X, Xp)
    voxel_size: tuple[float, float, float]


def _trig(ls_angle_deg: float) -> tuple[float, float]:
    # libm (math), not numpy: numpy may dispatch to SIMD kernels that differ by an ulp between
    # machines, and an ulp of cos(theta) moves samples that sit exactly on the volume border.
    theta = ls_angle_deg * math.pi / 180
    return math.sin(theta), math.cos(theta)


def scan_extent(n_scan: int, n_tilt: int, ls_angle_deg: float, px_to_scan_ratio: float,
                keep_overhang: bool) -> int:
    """Length Xp of the deskewed scan axis (``ceil``, may be <= 0 for a too-short scan)."""
    _, ct = _trig(ls_angle_deg)
    if keep_overhang:
        return int(math.ceil((n_scan / px_to_scan_ratio) + (n_tilt * ct)))
    return int(math.ceil((n_scan / px_to_scan_ratio) - (n_tilt * ct)))


def deskew_matrix(raw_shape, ls_angle_deg: float, px_to_scan_ratio: float,
                  keep_overhang: bool) -> np.ndarray:
    """3x4 output->input matrix of the deskew shear (scipy.ndimage convention)."""
    _, n_tilt, n_x = (int(s) for s in raw_shape)
    _, ct = _trig(ls_angle_deg)
    z_shift = 0 if keep_overhang else int(math.floor(n_tilt * ct * px_to_scan_ratio))
    return np.array(
        [
            [-px_to_scan_ratio * ct, 0.0, px_to_scan_ratio, z_shift],
            [-1.0, 0.0, 0.0, n_tilt - 1],
            [0.0, -1.0, 0.0, n_x - 1],
        ],
        dtype=np.float64,
    )


def deskew_geometry(raw_shape, ls_angle_deg: float, px_to_scan_ratio: float, keep_overhang: bool,
                    average_n_slices: int = 1, pixel_size_um: float = 1.0) -> DeskewGeometry:
    if len(raw_shape) != 3:
        raise ValueError(f"raw_data_shape must be (Z, Y, X), got {tuple(raw_shape)}")
    n_scan, n_tilt, n_x = (int(s) for s in raw_shape)
    if min(n_scan, n_tilt, n_x) <= 0:
        raise ValueError(f"raw_data_shape must be positive, got {tuple(raw_shape)}")
    if not (px_to_scan_ratio > 0):
        raise ValueError(f"px_to_scan_ratio must be positive, got {px_to_scan_ratio}")
    average_n_slices = int(average_n_slices)
    if average_n_slices < 1:
        raise ValueError(f"average_n_slices must be >= 1, got {average_n_slices}")
    st, _ = _trig(ls_angle_deg)
    xp = scan_extent(n_scan, n_tilt, ls_angle_deg, px_to_scan_ratio, keep_overhang)
    pre = (n_tilt, n_x, xp)
    out = (int(math.ceil(n_tilt / average_n_slices)), n_x, xp)
    voxel = (average_n_slices * st * pixel_size_um, pixel_size_um, pixel_size_um)
    matrix = deskew_matrix(raw_shape, ls_angle_deg, px_to_scan_ratio, keep_overhang)
    return DeskewGeometry(matrix, pre, out, voxel)


def as_matrix_3x4(matrix) -> np.ndarray:
    """Accept a 3x4 or a homogeneous 4x4 ZYX matrix; return float64 3x4."""
    m = np.asarray(matrix, dtype=np.float64)
    if m.shape == (4, 4):
        if not np.allclose(m[3], [0.0, 0.0, 0.0, 1.0]):
            raise ValueError("last row of a 4x4 affine must be [0, 0, 0, 1]")
        m = m[:3]
    if m.shape != (3, 4):
        raise ValueError(f"affine must be 3x4 or 4x4, got {m.shape}")
    if not np.all(np.isfinite(m)):
        raise ValueError("affine contains non-finite entries")
    return np.ascontiguousarray(m)


# ---------------------------------------------------------------------------------------------
# Orientation of the deskewed volume (a Python-level post-step, never baked into the kernel).
#
# Which way round ``fast_deskew_zyx`` hands the volume back is a convention of the absent biahub
# revision (the comment at ``shrimpy/preprocessing.py:224`` still mentions
# an ``np.rot90`` the current ``_deskew`` body no longer contains).  The kernel always produces the
# canonical (Z' = reversed tilt, Y' = reversed raw X, X' = scan) volume; an ``orientation`` spec is
# applied to that result, so a corrected convention is a settings change.
#
# Spec: ``"identity"`` or a ``+``-joined sequence, applied left to right, of
#   flip_z | flip_y | flip_x      reverse one output axis
#   transpose_yx                  swap Y' and X'
#   rot90 | rot180 | rot270       ``numpy.rot90(v, k, axes=(1, 2))`` with k = 1, 2, 3
# ---------------------------------------------------------------------------------------------

_ORIENT_OPS = ("identity", "flip_z", "flip_y", "flip_x", "transpose_yx", "rot90", "rot180", "rot270")


def parse_orientation(spec) -> tuple[str, ...]:
    """Validate an orientation spec; returns its operations (``identity`` dropped)."""
    if spec is None:
        return ()
    if not isinstance(spec, str):
        raise TypeError(f"orientation must be a string, got {type(spec).__name__}")
    ops = tuple(op.strip() for op in spec.split("+"))
    for op in ops:
        if op not in _ORIENT_OPS:
            raise ValueError(f"orientation op {op!r} is not one of {_ORIENT_OPS}")
    return tuple(op for op in ops if op != "identity")


def orient_axes(spec) -> tuple[tuple[int, int, int], tuple[bool, bool, bool]]:
    """``(perm, reversed)``: output axis ``i`` is canonical axis ``perm[i]``, walked backwards when
    ``reversed[i]``.  (``out = canonical.permute(perm).flip(dims where reversed)``.)"""
    perm, rev = [0, 1, 2], [False, False, False]

    def swap_yx():
        perm[1], perm[2] = perm[2], perm[1]
        rev[1], rev[2] = rev[2], rev[1]

    for op in parse_orientation(spec):
        if op.startswith("flip_"):
            a = "zyx".index(op[-1])
            rev[a] = not rev[a]
        elif op == "transpose_yx":
            swap_yx()
        else:  # numpy.rot90(v, 1, axes=(1, 2)) == swapaxes(flip(v, 2), 1, 2)
            for _ in range({"rot90": 1, "rot180": 2, "rot270": 3}[op]):
                rev[2] = not rev[2]
                swap_yx()
    return tuple(perm), tuple(rev)


def orient_shape(shape_zyx, spec) -> tuple[int, int, int]:
    perm, _ = orient_axes(spec)
    return tuple(int(shape_zyx[a]) for a in perm)


def orient_voxel(voxel_zyx, spec) -> tuple[float, float, float]:
    perm, _ = orient_axes(spec)
    return tuple(float(voxel_zyx[a]) for a in perm)


def raw_x_chunk_layout(spec) -> tuple[int, bool]:
    """Where chunks cut along raw X end up: ``(output axis, reversed)``.

    The reference deskews raw-X chunks independently and joins them reversed on output axis -2
    (``scripts/measure_psf.py:221, 249``) -- that is ``(1, True)``, the canonical orientation.  Raw X
    is canonical Y' walked backwards, so under an orientation the join axis is wherever Y' went and
    the order flips once more if that axis is reversed.
    """
    perm, rev = orient_axes(spec)
    axis = perm.index(1)
    return axis, not rev[axis]
