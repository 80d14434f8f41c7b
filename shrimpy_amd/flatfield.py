"""Bright-field flat-field correction on MI355X.

Mirrors ``_LabelfreePreprocessor._flat_field_BF`` (reference ``shrimpy/preprocessing.py:385-404``):
``static_pattern = volume.quantile(0.5, dim=0)``; ``volume / static_pattern * static_pattern.mean()``.
The median runs as a streaming radix select (``csrc/flatfield.hip``: at most four reads of the
volume); the division is either its own launch or fused into the deskew kernel's staging pass
(``deskew_with_matrix(..., flat_field=pattern)``).  Bright field only -- see the reference's note.
CPU tensors run the native host twins (``lsr_flatfield_*_cpu``, same median rule, f64 mean).
"""

from __future__ import annotations

from dataclasses import dataclass

from . import _lib

__all__ = ["FlatFieldPattern", "flat_field_pattern", "flat_field_bf", "flat_field_correction"]


@dataclass
class FlatFieldPattern:
    """Per-pixel median over Z (``pattern``, (Y, X) device tensor) and its mean (1-element tensor)."""

    pattern: object
    mean: object

    def apply(self, volume, out=None):
        """``volume / pattern * mean`` (new tensor, or into ``out`` -- which may be ``volume``)."""
        import torch

        u16 = isinstance(volume, torch.Tensor) and volume.dtype == torch.uint16
        host = isinstance(volume, torch.Tensor) and volume.device.type == "cpu"
        if host:
            vol = volume.contiguous() if u16 else volume.to(torch.float32).contiguous()
        else:
            vol = _require_raw(volume, "volume") if u16 else _lib.require_device_f32(volume, "volume")
        if vol.dim() != 3 or tuple(vol.shape[1:]) != tuple(self.pattern.shape):
            raise ValueError(f"volume must be (Z, {self.pattern.shape[0]}, {self.pattern.shape[1]}), "
                             f"got {tuple(vol.shape)}")
        if out is None:
            out = torch.empty(tuple(vol.shape), dtype=torch.float32, device=vol.device)
        else:
            if not host:
                _lib.require_device_f32(out, "out")
            if (tuple(out.shape) != tuple(vol.shape) or out.device != vol.device or out.dtype != torch.float32
                    or not out.is_contiguous()):
                raise ValueError("out must match volume")
        z, y, x = (int(v) for v in vol.shape)
        if host:     # no HIP device in play: the native host twin (csrc/host_twins.hip)
            _lib.call("lsr_flatfield_apply_u16_cpu" if u16 else "lsr_flatfield_apply_f32_cpu", vol.data_ptr(),
                      self.pattern.data_ptr(), self.mean.data_ptr(), out.data_ptr(), z, y, x, None)
            _lib.mark_written(out)
            return out
        with torch.cuda.device(vol.device):
            _lib.call("lsr_flatfield_apply_u16" if u16 else "lsr_flatfield_apply_f32", vol.data_ptr(), self.pattern.data_ptr(),
                      self.mean.data_ptr(), out.data_ptr(), z, y, x, _lib.stream_ptr(vol.device))
        _lib.mark_written(out)
        return out


def _require_raw(t, name):
    import torch

    if t.device.type != "cuda":
        raise _lib.LsrError("require_device", -1, f"{name} is on {t.device}; this path runs only on a HIP "
                            "device (MI355X); CPU tensors take the host twins (shrimpy_amd.host) from the public functions.")
    return t.contiguous()


def flat_field_pattern(volume) -> FlatFieldPattern:
    """Median over Z of every (y, x) pixel of a (Z, Y, X) device tensor -- float32, or the camera's
    uint16 counts as they are -- and its mean."""
    import torch

    if not isinstance(volume, torch.Tensor):
        raise TypeError(f"volume must be a torch.Tensor, got {type(volume).__name__}")
    if volume.dim() != 3:
        raise ValueError(f"volume must be (Z, Y, X), got shape {tuple(volume.shape)}")
    u16 = volume.dtype == torch.uint16
    if volume.device.type == "cpu":     # no HIP device in play: the native host twin
        vol = volume.contiguous() if u16 else volume.to(torch.float32).contiguous()
        z, y, x = (int(v) for v in vol.shape)
        pattern, mean = torch.empty((y, x), dtype=torch.float32), torch.empty((1,), dtype=torch.float32)
        from . import host

        host._threads()
        _lib.call("lsr_flatfield_pattern_u16_cpu" if u16 else "lsr_flatfield_pattern_f32_cpu", vol.data_ptr(), z, y, x,
                  pattern.data_ptr(), mean.data_ptr(), None, None)
        return FlatFieldPattern(pattern, mean)
    if u16:
        vol = _require_raw(volume, "volume")
    else:
        if volume.dtype != torch.float32:
            volume = volume.to(torch.float32)
        vol = _lib.require_device_f32(volume.contiguous(), "volume")
    z, y, x = (int(v) for v in vol.shape)
    pattern = torch.empty((y, x), dtype=torch.float32, device=vol.device)
    mean = torch.empty((1,), dtype=torch.float32, device=vol.device)
    scratch = torch.empty((_lib.call_value("lsr_flatfield_scratch_bytes"),), dtype=torch.uint8,
                          device=vol.device)
    with torch.cuda.device(vol.device):
        _lib.call("lsr_flatfield_pattern_u16" if u16 else "lsr_flatfield_pattern_f32", vol.data_ptr(), z, y, x, pattern.data_ptr(),
                  mean.data_ptr(), scratch.data_ptr(), _lib.stream_ptr(vol.device))
    return FlatFieldPattern(pattern, mean)


def flat_field_bf(volume):
    """The reference's ``_flat_field_BF`` on the device: new corrected (Z, Y, X) tensor."""
    import torch

    if isinstance(volume, torch.Tensor) and volume.dtype not in (torch.float32, torch.uint16):
        volume = volume.to(torch.float32)
    return flat_field_pattern(volume).apply(volume.contiguous())


def flat_field_correction(data, axis: int = 0):
    """``biahub.flat_field_correction.flat_field_correction(data, axis)`` -- the numpy-level entry the reference's
    own test compares its ``_flat_field_BF`` with (``shrimpy/tests/test_preprocessing.py:145-162``): divide a stack
    by its per-pixel median along ``axis`` and keep the pattern's mean.  A numpy array comes back as a numpy array
    (computed where a tensor of it lives: the host twins without a GPU tensor), a tensor as a tensor."""
    import numpy as np
    import torch

    is_tensor = isinstance(data, torch.Tensor)
    t = data if is_tensor else torch.as_tensor(np.ascontiguousarray(data))
    if t.dim() != 3:
        raise ValueError(f"expected a 3-D stack, got shape {tuple(t.shape)}")
    moved = t.movedim(int(axis), 0).contiguous()
    out = flat_field_bf(moved).movedim(0, int(axis))
    return out if is_tensor else out.contiguous().cpu().numpy()
