"""Drop-in for ``biahub.deskew`` on MI355X: same names, argument meaning and error behaviour.

``import shrimpy_amd.deskew as biahub_deskew`` serves every call the reference makes:

* ``fast_deskew_zyx(raw_data=volume, **kwargs)``            ``shrimpy/preprocessing.py:408-413``
* ``get_deskewed_data_shape(raw_data_shape=..., **kwargs)`` ``shrimpy/preprocessing.py:226-231``,
  ``scripts/measure_psf.py:230-234`` (with ``pixel_size_um=``)
* ``deskew_data(chunk, device=..., **settings)`` (older API)  ``scripts/measure_psf.py:238-246``

The reference filters kwargs by ``inspect.signature`` (``shrimpy/preprocessing.py:53-56``), so the
parameter NAMES below are the API.  The arithmetic runs in the HIP kernel ``lsr_deskew_f32``
(``csrc/deskew.hip``): fused slice averaging, results bit-identical to
``scipy.ndimage.affine_transform(order=1, mode="constant", cval=0)`` + edge-padded float32 mean.
A CPU tensor -- what the reference hands over on a box without a GPU (``shrimpy/preprocessing.py:78-82``) --
runs the native host twin of the same entry point (``shrimpy_amd.host`` / ``csrc/host_twins.hip``: same
arithmetic, same bits); nothing routes through the test oracle, and a missing library raises.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _lib, host
from .geometry import as_matrix_3x4, deskew_geometry, orient_axes, orient_shape, orient_voxel

__all__ = [
    "fast_deskew_zyx",
    "get_deskewed_data_shape",
    "deskew_data",
    "deskew_with_matrix",
    "average_n_slices",
    "orient_volume",
]

BORDERS = ("constant", "grid-constant")


def get_deskewed_data_shape(
    raw_data_shape,
    ls_angle_deg: float,
    px_to_scan_ratio: float,
    keep_overhang: bool,
    average_n_slices: int = 1,
    pixel_size_um: float = 1,
    orientation: str = "identity",
):
    """Shape of the deskewed volume and its voxel size.

    Returns ``((ceil(Y/avg), X, Xp), (avg*sin(theta)*px, px, px))`` for a raw ``(Z, Y, X)`` stack,
    with the axes permuted as ``orientation`` says (``geometry.parse_orientation``).
    """
    geo = deskew_geometry(
        raw_data_shape, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices, pixel_size_um
    )
    return orient_shape(geo.output_shape, orientation), orient_voxel(geo.voxel_size, orientation)


def orient_volume(volume, orientation: str = "identity"):
    """Apply an orientation spec to a canonical deskewed tensor: an axis permutation (a view) and
    flips (one copy, only when the spec is not the identity).  Python-level on purpose -- the
    orientation convention of the absent biahub revision is [RECALLED], and
    changing it must not touch the kernel."""
    import torch

    perm, rev = orient_axes(orientation)
    if perm == (0, 1, 2) and not any(rev):
        return volume
    out = volume.permute(*perm)
    dims = [i for i, r in enumerate(rev) if r]
    return (torch.flip(out, dims) if dims else out).contiguous()


def fill_value(raw, cval):
    """The ``cval`` argument as what the kernels take: ``None`` for zero, else a one-element float32 tensor on ``raw``'s
    device -- the number itself, or for ``"min"`` / ``None``-as-minimum the stack's minimum, reduced on the device
    (``lsr_minmax_f32`` / ``lsr_minmax_u16``) without a host round trip."""
    import torch

    if cval is None:
        cval = "min"
    if isinstance(cval, str):
        if cval != "min":
            raise ValueError(f'cval must be a number or "min", got {cval!r}')
        if host.is_host(raw):
            src = raw.to(torch.int32) if raw.dtype == torch.uint16 else raw
            return src.min().to(torch.float32).reshape(1)
        out2 = torch.empty(2, dtype=torch.float32, device=raw.device)
        scratch = torch.empty(_lib.call_value("lsr_reduce_scratch_bytes"), dtype=torch.uint8, device=raw.device)
        with torch.cuda.device(raw.device):
            _lib.call("lsr_minmax_u16" if raw.dtype == torch.uint16 else "lsr_minmax_f32", raw.data_ptr(), raw.numel(),
                      out2.data_ptr(), scratch.data_ptr(), _lib.stream_ptr(raw.device))
        return out2[:1]
    value = float(cval)
    if value != value or value in (float("inf"), float("-inf")):
        raise ValueError(f"cval must be finite, got {cval!r}")
    if value == 0.0:
        return None
    return torch.full((1,), value, dtype=torch.float32, device=raw.device)


def deskew_with_matrix(raw_data, matrix_3x4, pre_average_shape, average_n_slices: int = 1, out=None,
                       flat_field=None, border: str = "constant", cval=0.0):
    """Deskew with an explicit output->input matrix over the pre-average grid.

    ``flat_field`` = a :class:`shrimpy_amd.flatfield.FlatFieldPattern` of ``raw_data``: the
    bright-field correction is then applied to every raw sample inside the deskew kernel
    (bit-identical to correcting first, without writing the corrected volume).

    Shear-structured matrices (only ``z_in`` fractional) run the fused transpose kernel; any other
    3x4 map runs the general trilinear kernel followed by the slice-averaging kernel.
    ``out`` may be a dense tensor or a :class:`shrimpy_amd.deconvolve.PaddedVolume` (shear
    matrices only): the deskewed volume then lands, line-aligned, where the RL kernels read it.

    ``border``: ``"constant"`` (default) is scipy's ``mode="constant"`` -- a sample with any
    coordinate outside ``[0, n-1]`` is zero, no blending; ``"grid-constant"`` blends towards zero
    across the border the way ``scipy mode="grid-constant"`` / torch ``grid_sample(zeros)`` do
    (SURVEY.md section 7).  Both rules run the fused transpose kernel (``lsr_deskew_border``) and may
    write into a padded RL volume; matrices that are not a deskew shear fall back to the general
    trilinear kernel under either rule.

    ``cval``: the value outside the stack, as a float32 -- a number (default 0) or ``"min"`` / ``None`` for the stack's minimum
    (scipy's ``cval``; [RECALLED] biahub's ``deskew_data(cval=None)`` fills with the minimum: the third unpinned
    convention beside ``orientation`` and ``border``).  With flat-field fusion the minimum is
    that of the uncorrected stack.
    """
    import torch

    if border not in BORDERS:
        raise ValueError(f"border must be one of {BORDERS}, got {border!r}")
    if host.is_host(raw_data):     # no HIP device in play: the native host twin (same arithmetic, same bits)
        if raw_data.dim() != 3:
            raise ValueError(f"raw_data must be (Z, Y, X), got shape {tuple(raw_data.shape)}")
        avg = int(average_n_slices)
        if avg < 1:
            raise ValueError(f"average_n_slices must be >= 1, got {avg}")
        if min(int(v) for v in pre_average_shape) <= 0:
            raise ValueError(
                f"deskewed shape {tuple(int(v) for v in pre_average_shape)} is empty: the scan is too short for this "
                "tilt (use keep_overhang=True or a longer scan)")
        return host.deskew_with_matrix(raw_data, as_matrix_3x4(matrix_3x4), pre_average_shape, avg, out=out,
                                       flat_field=flat_field, border=border, cval=fill_value(raw_data, cval))

    # uint16 camera counts are deskewed as they are (converted to float32 inside the kernel, exact):
    # half the HBM read and, upstream, half the PCIe upload of a float32 stack
    u16 = isinstance(raw_data, torch.Tensor) and raw_data.dtype == torch.uint16
    if u16:
        if raw_data.device.type != "cuda" or not raw_data.is_contiguous():
            raise _lib.LsrError("require_device", -1, "raw_data must be a contiguous tensor on a HIP device "
                                "(MI355X); CPU tensors take the host twin earlier in this function.")
        raw = raw_data
    else:
        raw = _lib.require_device_f32(raw_data, "raw_data")
    if raw.dim() != 3:
        raise ValueError(f"raw_data must be (Z, Y, X), got shape {tuple(raw.shape)}")
    m = as_matrix_3x4(matrix_3x4)
    zd, yo, xo = (int(v) for v in pre_average_shape)
    avg = int(average_n_slices)
    if avg < 1:
        raise ValueError(f"average_n_slices must be >= 1, got {avg}")
    if min(zd, yo, xo) <= 0:
        raise ValueError(
            f"deskewed shape {(zd, yo, xo)} is empty: the scan is too short for this tilt "
            "(use keep_overhang=True or a longer scan)"
        )
    zo = -(-zd // avg)
    result = out
    if out is None:
        out = result = torch.empty((zo, yo, xo), dtype=torch.float32, device=raw.device)
        out_ptr, out_pitch, out_plane = out.data_ptr(), xo, yo * xo
    elif hasattr(out, "logical_ptr"):  # a PaddedVolume: write the RL input in place
        if tuple(out.view.shape) != (zo, yo, xo) or out.full.device != raw.device:
            raise ValueError(f"out must hold a {(zo, yo, xo)} window on {raw.device}")
        out_ptr, out_pitch, out_plane = out.logical_ptr(), out.pitch, out.plane
    else:
        _lib.require_device_f32(out, "out")
        if tuple(out.shape) != (zo, yo, xo) or out.device != raw.device:
            raise ValueError(f"out must be {(zo, yo, xo)} on {raw.device}")
        out_ptr, out_pitch, out_plane = out.data_ptr(), xo, yo * xo
    z, y, x = (int(v) for v in raw.shape)
    fill = fill_value(raw, cval)
    with torch.cuda.device(raw.device):
        stream = _lib.stream_ptr(raw.device)
        if flat_field is not None and tuple(flat_field.pattern.shape) != (y, x):
            raise ValueError(f"flat_field pattern must be {(y, x)}, got {tuple(flat_field.pattern.shape)}")
        try:
            if fill is not None:
                _lib.call(
                    "lsr_deskew_cval", raw.data_ptr(), 1 if u16 else 0, z, y, x, out_ptr, zo, yo, xo, out_pitch,
                    out_plane, zd, _lib.matrix12(m), avg,
                    _lib.MODE_CONSTANT if border == "constant" else _lib.MODE_GRID_CONSTANT,
                    flat_field.pattern.data_ptr() if flat_field is not None else None,
                    flat_field.mean.data_ptr() if flat_field is not None else None, fill.data_ptr(), stream,
                )
            elif border != "constant":
                _lib.call(
                    "lsr_deskew_border", raw.data_ptr(), 1 if u16 else 0, z, y, x, out_ptr, zo, yo, xo, out_pitch,
                    out_plane, zd, _lib.matrix12(m), avg, _lib.MODE_GRID_CONSTANT,
                    flat_field.pattern.data_ptr() if flat_field is not None else None,
                    flat_field.mean.data_ptr() if flat_field is not None else None, stream,
                )
            elif flat_field is not None:
                _lib.call(
                    "lsr_deskew_flat_u16" if u16 else "lsr_deskew_flat_f32", raw.data_ptr(), z, y, x, out_ptr, zo, yo, xo, out_pitch,
                    out_plane, zd, _lib.matrix12(m), avg, flat_field.pattern.data_ptr(),
                    flat_field.mean.data_ptr(), stream,
                )
            else:
                _lib.call(
                    "lsr_deskew_u16" if u16 else "lsr_deskew_f32", raw.data_ptr(), z, y, x, out_ptr, zo, yo, xo,
                    out_pitch, out_plane, zd, _lib.matrix12(m), avg, stream,
                )
        except _lib.LsrUnsupported:
            if hasattr(out, "logical_ptr"):
                raise
            if flat_field is not None:  # general matrix: correct first, then resample
                raw = flat_field.apply(raw)
            elif u16:
                raw = raw.to(torch.float32)
            # general matrix: trilinear gather, then average
            pre = out if avg == 1 else torch.empty((zd, yo, xo), dtype=torch.float32, device=raw.device)
            _lib.call(
                "lsr_affine_f32", raw.data_ptr(), z, y, x, pre.data_ptr(), zd, yo, xo,
                _lib.matrix12(m), ctypes.c_float(0.0 if fill is None else float(fill.item())),
                _lib.MODE_CONSTANT if border == "constant" else _lib.MODE_GRID_CONSTANT, stream,
            )
            if avg > 1:
                _average_into(pre, out, avg, stream)
    _lib.mark_written(out.full if hasattr(out, "logical_ptr") else out)
    return result


def _average_into(pre, out, avg: int, stream: int) -> None:
    zd, y, x = (int(v) for v in pre.shape)
    _lib.call("lsr_average_slices_f32", pre.data_ptr(), zd, y, x, out.data_ptr(), int(out.shape[0]), avg, stream)


def average_n_slices(data, average_window_width: int = 1):
    """Mean over groups of slices along axis 0, remainder edge-padded (device tensor in/out)."""
    import torch

    avg = int(average_window_width)
    if avg < 1:
        raise ValueError(f"average_window_width must be >= 1, got {avg}")
    if host.is_host(data):
        return host.average_n_slices(data, avg)
    data = _lib.require_device_f32(data, "data")
    if avg == 1:
        return data
    zd = int(data.shape[0])
    out = torch.empty((-(-zd // avg),) + tuple(data.shape[1:]), dtype=torch.float32, device=data.device)
    with torch.cuda.device(data.device):
        _average_into(data, out, avg, _lib.stream_ptr(data.device))
    return out


def fast_deskew_zyx(
    raw_data,
    ls_angle_deg: float,
    px_to_scan_ratio: float,
    keep_overhang: bool,
    average_n_slices: int = 1,
    orientation: str = "identity",
    border: str = "constant",
    cval=0.0,
):
    """Deskew a raw ``(Z_scan, Y_tilt, X)`` float32 device tensor; returns a tensor on the same device.

    Output axes ``(Z', Y', X')``: ``Z'`` = reversed tilt rows averaged in groups of
    ``average_n_slices``, ``Y'`` = reversed raw X, ``X'`` = scan direction (the interpolated axis).
    ``orientation`` re-orients that canonical result afterwards (``orient_volume``); ``border``
    selects the border rule and ``cval`` the value outside the stack -- a number or ``"min"``
    (``deskew_with_matrix``).  The reference filters the kwargs it passes by
    this signature (``shrimpy/preprocessing.py:44-56``): settings fields that are not parameters
    here never arrive, fields that are (these two, when ``DeskewSettings`` carries them) do.
    """
    import torch

    if not isinstance(raw_data, torch.Tensor):
        raise TypeError(f"raw_data must be a torch.Tensor, got {type(raw_data).__name__}")
    if raw_data.dim() != 3:
        raise ValueError(f"raw_data must be (Z, Y, X), got shape {tuple(raw_data.shape)}")
    if raw_data.dtype not in (torch.float32, torch.uint16):   # uint16 counts go in as they are
        raw_data = raw_data.to(torch.float32)
    raw_data = raw_data.contiguous()
    geo = deskew_geometry(
        tuple(raw_data.shape), ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices
    )
    out = deskew_with_matrix(raw_data, geo.matrix_3x4, geo.pre_average_shape, average_n_slices,
                             border=border, cval=cval)
    return orient_volume(out, orientation)


def deskew_data(
    raw_data,
    ls_angle_deg: float,
    px_to_scan_ratio: float,
    keep_overhang: bool,
    average_n_slices: int = 1,
    device="cuda",
    orientation: str = "identity",
    border: str = "constant",
    cval=0.0,
):
    """Older biahub entry point: numpy in, numpy out, compute on ``device`` (``"cpu"`` runs the host twin)."""
    import torch

    dev = torch.device(device)
    vol = torch.as_tensor(np.ascontiguousarray(raw_data, dtype=np.float32), device=dev)
    out = fast_deskew_zyx(vol, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices,
                          orientation=orientation, border=border, cval=cval)
    return out.cpu().numpy()
