"""Chunk codecs that run on the MI355X: blosc-zstd frames written (and read) in HBM.

The acquisition's format is Zarr v3 shards of blosc frames with the zstd compressor and byte shuffle
(``shrimpy/mantis/mantis_engine.py:474-481``; asserted in ``shrimpy/tests/test_mantis_integration.py:177-190``),
and the CLI writes its results the same way.  Round 4 measured what that costs on the host: zstd level 1 over the
1.75 GB float32 result of one config-4 unit is ~6 core-seconds -- 0.38 s per unit on a rank's 16 cores against
30 ms of kernels.  Here the frames are produced where the result already is:

* :class:`DeviceBloscEncoder` -- ``lsr_blosc_encode_device`` (``csrc/blosc_encode.hip``): a volume in HBM ->
  one c-blosc 1.x frame per Zarr chunk, compacted in one device buffer, plus the (offset, size) table.  The
  download then moves compressed bytes and the writer threads only ``pwrite`` them.
* the host twin (``lsr_blosc_encode_device_cpu``: the same bytes from host memory) serves CPU tensors and the
  GPU-less tests, where its frames are decoded by the system libzstd.

The frames are a subset of zstd (Huffman-coded literals, RLE and raw blocks, no sequences) that every zstd
decoder reads -- numcodecs / c-blosc behind iohub included.
"""

from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib

__all__ = ["DeviceBloscEncoder", "encode_frames_host", "plan_frames"]


def plan_frames(src_bytes: int, typesize: int, frame_bytes: int, blocksize: int = 0) -> tuple[int, int, int]:
    """``(n_frames, scratch_bytes, out_capacity)`` of one encode call."""
    nf, sb, cap = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _lib.call("lsr_blosc_encode_device_plan", int(src_bytes), int(typesize), int(frame_bytes), int(blocksize),
              ctypes.byref(nf), ctypes.byref(sb), ctypes.byref(cap))
    return int(nf.value), int(sb.value), int(cap.value)


def encode_frames_host(array: np.ndarray, frame_bytes: int, blocksize: int = 0) -> list[bytes]:
    """The host twin: ``array`` (C-contiguous, 1/2/4-byte elements) cut into chunks of ``frame_bytes`` (the last one
    zero-padded), each as one blosc-zstd frame -- byte for byte what :class:`DeviceBloscEncoder` writes."""
    raw = np.ascontiguousarray(array).reshape(-1).view(np.uint8)
    typesize = int(np.asarray(array).dtype.itemsize)
    n_frames, _, cap = plan_frames(raw.size, typesize, frame_bytes, blocksize)
    out = np.empty(cap, dtype=np.uint8)
    frames = np.zeros(2 * n_frames, dtype=np.int64)
    _lib.call("lsr_blosc_encode_device_cpu", raw.ctypes.data, raw.size, typesize, int(frame_bytes), int(blocksize),
              None, 0, out.ctypes.data, out.size, frames.ctypes.data, None)
    return [out[frames[2 * f]:frames[2 * f] + frames[2 * f + 1]].tobytes() for f in range(n_frames)]


class DeviceBloscEncoder:
    """Frames of every chunk of one volume shape, written by the GPU.

    Buffers (block streams, the compacted frames, the frame table) are allocated once for the shape and reused:
    ``encode`` launches three kernels on the current stream and returns device tensors without synchronising.
    """

    def __init__(self, src_bytes: int, typesize: int, frame_bytes: int, device, blocksize: int = 0):
        import torch

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LsrError("DeviceBloscEncoder", -1, f"device {self.device}: the device encoder needs a HIP device "
                                "(CPU arrays go through encode_frames_host)")
        self.src_bytes, self.typesize, self.frame_bytes, self.blocksize = int(src_bytes), int(typesize), int(frame_bytes), int(blocksize)
        self.n_frames, scratch, cap = plan_frames(src_bytes, typesize, frame_bytes, blocksize)
        self.capacity = cap
        self._scratch = torch.empty(scratch, dtype=torch.uint8, device=self.device)
        self.out = torch.empty(cap, dtype=torch.uint8, device=self.device)
        self.frames = torch.zeros((self.n_frames, 2), dtype=torch.int64, device=self.device)

    def encode(self, volume):
        """``volume``: contiguous device tensor of ``src_bytes`` bytes.  Returns ``(out, frames)``: the uint8 buffer
        holding the frames and the ``(n_frames, 2)`` int64 table of (offset, size), both on the device; valid after
        the current stream reaches this point, until the next ``encode``."""
        if volume.device != self.device or not volume.is_contiguous():
            raise ValueError("volume must be a contiguous tensor on the encoder's device")
        if volume.numel() * volume.element_size() != self.src_bytes:
            raise ValueError(f"volume has {volume.numel() * volume.element_size()} bytes, the encoder was planned for {self.src_bytes}")
        if volume.element_size() != self.typesize:
            raise ValueError(f"element size {volume.element_size()} differs from the planned typesize {self.typesize}")
        _lib.call("lsr_blosc_encode_device", volume.data_ptr(), self.src_bytes, self.typesize, self.frame_bytes,
                  self.blocksize, self._scratch.data_ptr(), self._scratch.numel(), self.out.data_ptr(), self.out.numel(),
                  self.frames.data_ptr(), _lib.stream_ptr(self.device))
        return self.out, self.frames

    def encode_to_host(self, volume) -> list[bytes]:
        """Convenience (tests, small volumes): encode, synchronise, return the frames as ``bytes``."""
        out, frames = self.encode(volume)
        table = frames.cpu().numpy()
        end = int(table[-1, 0] + table[-1, 1])
        host = out[:end].cpu().numpy()
        return [host[o:o + n].tobytes() for o, n in table]
