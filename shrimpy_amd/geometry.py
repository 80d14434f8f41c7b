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
